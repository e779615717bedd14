"""Experiment: time the neighbourhood kernels per degree class on a skewed (RMAT) graph.

    python tools/exp_skew.py [--scale 20] [--edges 10000000] [--slices 256] [--weighted]

Prints, per class (register path 1..32, LDS path 33..2048, global path > 2048): rows, edges, ms, and the achieved
gather rate (4 B * edges * S / time).  The forward of config 3 never leaves the register path; this is the tool for
the graphs that do.
"""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fsw_gnn_amd import FSW_embedding, _lib, synth   # noqa: E402
from fsw_gnn_amd.graph import build_csr   # noqa: E402


def timed_ms(fn, reps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=int, default=20)
    ap.add_argument("--edges", type=int, default=10_000_000)
    ap.add_argument("--slices", type=int, default=256)
    ap.add_argument("--feat", type=int, default=128)
    ap.add_argument("--weighted", action="store_true")
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    n, S, d = 1 << args.scale, args.slices, args.feat
    ei = torch.from_numpy(synth.rmat_graph(args.scale, args.edges, 7)).to(dev)
    w = torch.from_numpy(synth.edge_weights(args.edges, 9)).to(dev) if args.weighted else None
    x = torch.from_numpy(synth.features(n, d, 3)).to(dev)
    emb = FSW_embedding(d, S + 1, device=dev, encode_total_mass=True)
    L = _lib.lib()
    stream = torch.cuda.current_stream(dev).cuda_stream
    graph = build_csr(ei[1], ei[0], w, n, n)
    st = graph.stats()
    deg = graph.in_degrees().cpu().numpy()
    print("build %.3f ms; max degree %d" % (timed_ms(lambda: build_csr(ei[1], ei[0], w, n, n), 3), st[_lib.STAT_MAX_DEGREE]))
    prepared = emb.prepare(x, graph)
    Xp, ldp, table = prepared["Xp"], prepared["ldp"], prepared["table"]
    out = torch.empty((n, S + 1), dtype=torch.float32, device=dev)
    sb = int(L.fsw_embed_scratch_bytes(int(st[_lib.STAT_MAX_DEGREE])))
    scratch = torch.empty(max(sb, 16), dtype=torch.uint8, device=dev)
    classes = {"reg": (deg >= 1) & (deg <= 32), "lds": (deg > 32) & (deg <= 2048), "global": deg > 2048}
    for name, mask in classes.items():
        a = emb.make_args(graph, st, Xp, ldp, emb.freqs.detach(), S, table, out.data_ptr(), out.stride(0), None, 1.0, 1, scratch=scratch)
        a.num_zero_rows = 0
        if name != "reg": a.num_reg_rows = 0
        if name != "lds": a.num_lds_rows = 0
        if name != "global": a.num_global_rows = 0
        rows, edges = int(mask.sum()), int(deg[mask].sum())
        if rows == 0:
            print("%-6s no rows" % name)
            continue
        ms = timed_ms(lambda: _lib.check(L.fsw_embed_f32(ctypes.byref(a), stream), "embed"), args.reps)
        print("%-6s rows %8d edges %9d  %.3f ms  gather %.0f GB/s" % (name, rows, edges, ms, 4.0 * edges * S / ms / 1e6), flush=True)


if __name__ == "__main__":
    main()
