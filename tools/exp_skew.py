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
    ap.add_argument("--only", default="", help="substring of the class name to time")
    ap.add_argument("--fine", action="store_true", help="one line per degree BIN of the 33..256 class (padded sizes FSW_MID_SIZES)")
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
    degt = graph.in_degrees()
    # one sub-graph per degree class: only the edges whose recipient's in-degree falls in the class
    classes = [("reg 1..32", 0, 32), ("mid 33..256", 32, 256), ("ws 257..512", 256, 512), ("ws 513..1024", 512, 1024),
               ("ws 1025..2048", 1024, 2048), ("hub 2049..4096", 2048, 4096), ("hub 4097..8192", 4096, 8192),
               ("hub 8193..16384", 8192, 16384), ("hub 16385..32768", 16384, 32768), ("global > 32768", 32768, 1 << 30)]
    if args.fine:
        edges_ = (32,) + tuple(_lib.MID_SIZES)
        classes = [("mid %d..%d" % (a_ + 1, b_), a_, b_) for a_, b_ in zip(edges_[:-1], edges_[1:])] + classes[2:]
    for name, lo, hi in classes:
        if args.only and args.only not in name:
            continue
        keep = (degt[ei[1]] > lo) & (degt[ei[1]] <= hi)
        edges = int(keep.sum())
        if edges == 0:
            print("%-18s no rows" % name)
            continue
        sub = ei[:, keep].contiguous()
        g = build_csr(sub[1], sub[0], w[keep].contiguous() if w is not None else None, n, n)
        stg = g.stats()
        scratch = None
        if stg[_lib.STAT_NUM_GLOBAL] > 0:
            scratch = torch.empty(int(L.fsw_embed_scratch_bytes(int(stg[_lib.STAT_MAX_DEGREE]))), dtype=torch.uint8, device=dev)
        a = emb.make_args(g, stg, Xp, ldp, emb.freqs.detach(), S, table, out.data_ptr(), out.stride(0), None, 1.0, 1, scratch=scratch)
        a.num_zero_rows = 0
        rows = int(((deg > lo) & (deg <= hi)).sum())
        ms = timed_ms(lambda: _lib.check(L.fsw_embed_f32(ctypes.byref(a), stream), "embed"), args.reps)
        print("%-18s rows %8d edges %9d  %8.3f ms  %6.1f Gkeys/s  gather %5.0f GB/s" % (name, rows, edges, ms, edges * S / ms / 1e6,
                                                                                   4.0 * edges * S / ms / 1e6), flush=True)


if __name__ == "__main__":
    main()
