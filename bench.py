#!/usr/bin/env python3
"""Headline benchmark: edges*slices/sec of FSW_conv.forward on the BASELINE config-3 workload.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): ER-style directed multigraph with
1,000,000 nodes / 10,000,000 edges, 128 features, FSW_conv(128 -> 128, embed_dim = 257) = 256 slices + degree
column, 'spread' frequencies, one Linear + LeakyReLU layer -- synthetic data, random-init weights.
A step is ONE full FSW_conv.forward(x, edge_index): CSR build from the int64 edge_index (rebuilt every step like
the reference, fsw_conv.py:352 -- nothing is cached), fp32-MFMA projection, fused neighbourhood sort / cumulative
sum / Fourier readout, concat with the vertex features and the Linear layer.  Inputs are resident in HBM before
the timed region.  With N > 1 the slice axis is sharded over the ranks (256 / N slices each, total work fixed ->
strong scaling) and reassembled by one RCCL all-gather.

The JSON line also carries
  roofline     the dominant kernel (k_conv_fused_unit; k_embed_reg_unit with --no-fuse) timed alone, HIP events on the launch stream:
               achieved = its algorithmic bytes per launch / mean duration, against the 8 TB/s HBM peak;
  cpu_baseline the C oracle (a port of the reference algorithm, oracle/fsw_oracle.c) on this box's host cores,
               rank 0 at N = 1 only, on a bounded sample of the same workload.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_NODES, N_EDGES, D_FEAT, EMBED_DIM, OUT_CH = 1_000_000, 10_000_000, 128, 257, 128
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--nodes", type=int, default=N_NODES)
    ap.add_argument("--edges", type=int, default=N_EDGES)
    ap.add_argument("--kernel-reps", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-slices", type=int, default=256)
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--no-fuse", action="store_true", help="force the unfused kernels (embedding written to HBM, torch Linear)")
    ap.add_argument("--shard", choices=("nodes", "slices"), default="nodes",
                    help="N > 1: 'nodes' = every rank runs the fused layer on its block of recipient rows, all-gather of the "
                         "128-wide output; 'slices' = BASELINE north_star's slice-axis shard, all-gather of the 257-wide embedding")
    return ap.parse_args()


def make_inputs(n, num_edges, dev):
    g = torch.Generator(device="cpu")
    g.manual_seed(1234)
    x = torch.randn((n, D_FEAT), generator=g, dtype=torch.float32)
    ei = torch.randint(0, n, (2, num_edges), generator=g, dtype=torch.int64)
    return x.to(dev), ei.to(dev)


def timed_ms(fn, reps, dev):
    """Mean milliseconds of fn() over reps launches, HIP events on the current (= launch) stream."""
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    torch.cuda.synchronize(dev)
    start.record()
    for _ in range(reps):
        fn()
    stop.record()
    torch.cuda.synchronize(dev)
    return start.elapsed_time(stop) / reps


def dominant_kernel_roofline(conv, x, ei, n, e_coalesced, reps, dev):
    """Times the dominant kernel of the step alone (HIP events on the launch stream) plus the other stages.

    Dominant kernel = k_conv_fused_unit (neighbourhood gather/sort/readout fused with the first Linear layer) when
    FSW_conv takes its fused path, else k_embed_reg_unit.  Both gather Xp[src, k] for every (edge, slice).
    """
    from fsw_gnn_amd import _lib
    L = _lib.lib()
    emb = conv.fsw_embed
    S = emb.nSlices
    stream = torch.cuda.current_stream(dev).cuda_stream
    graph = conv.build_graph(ei, n)
    fused = conv._fusable()
    y = torch.empty((n, conv.mlp[0].out_features), dtype=torch.float32, device=dev)
    yin = torch.empty_like(y)
    if fused:
        wq, w2 = conv._fused_weight()
        lin2 = (w2, conv.mlp[0].bias.detach(), yin)
    prepared = emb.prepare(x, graph, linear2=lin2 if fused else None)
    st = prepared["stats"]
    assert st[_lib.STAT_NUM_LDS] == 0 and st[_lib.STAT_NUM_GLOBAL] == 0, "config 3 is expected to sit on the register path"
    Xp, ldp, table = prepared["Xp"], prepared["ldp"], prepared["table"]
    V, fr = emb.projVecs.detach(), emb.freqs.detach()
    out = torch.empty((n, conv.embed_dim + conv.in_channels), dtype=torch.float32, device=dev)
    def project():
        if fused:   # projection + the x . W2^T + b half of the first Linear layer
            emb.prepare(x, graph, linear2=lin2)
        else:
            _lib.check(L.fsw_project_f32(x.data_ptr(), n, D_FEAT, D_FEAT, V.data_ptr(), S, D_FEAT, Xp.data_ptr(), ldp,
                                         out.data_ptr() + 4 * conv.embed_dim, out.stride(0), None, stream), "project")

    a = emb.make_args(graph, st, Xp, ldp, fr, S, table, out.data_ptr(), out.stride(0), None, 1.0, 1)
    a.num_zero_rows = 0                                # the timed call is exactly one launch: k_embed_reg_unit

    def embed():
        _lib.check(L.fsw_embed_f32(ctypes.byref(a), stream), "embed")

    ms = {
        "csr_build": timed_ms(lambda: conv.build_graph(ei, n), max(3, reps // 4), dev),
        "project": timed_ms(project, max(3, reps // 4), dev),
        "embed_reg_unit": timed_ms(embed, reps, dev),
    }
    rows_reg = st[_lib.STAT_NUM_REG]
    edges = int(graph.rowptr[-1])
    gather_bytes = 4.0 * edges * S + 4.0 * edges + 8.0 * n        # Xp gather + col + rowptr/perm
    if fused:
        ms["conv_fused_unit"] = timed_ms(lambda: conv._fused_linear(graph, prepared, 1.0, wq, yin, y), reps, dev)
        kernel, kms = "k_conv_fused_unit", ms["conv_fused_unit"]
        # + Y (= x . W2^T + b from the projection kernel) read once and written once (packed W1^T, 0.13 MB, stays in L2)
        alg_bytes = gather_bytes + 8.0 * n * conv.mlp[0].out_features
    else:
        ms["mlp"] = timed_ms(lambda: conv.mlp(out), max(3, reps // 4), dev)
        kernel, kms = "k_embed_reg_unit", ms["embed_reg_unit"]
        alg_bytes = gather_bytes + 4.0 * rows_reg * (S + 1)       # + embedding written once
    secs = kms * 1e-3
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic_latest.json")
    if os.path.isfile(tpath):
        try:
            traffic = json.load(open(tpath)).get(kernel + "_hbm_bytes_per_launch")
        except Exception:
            traffic = None
    roof = {"bound": "hbm", "kernel": kernel, "achieved": alg_bytes / secs / 1e9, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": alg_bytes / secs / 1e9 / HBM_PEAK_GBS, "traffic": traffic,
            "algorithmic_bytes_per_launch": alg_bytes, "ms_per_launch": kms,
            "bytes_per_edge_slice": alg_bytes / (float(e_coalesced) * S)}
    return roof, ms


def sharded_kernel_roofline(conv, x, ei, n, reps, dev, rank, world):
    """N > 1: the dominant kernel of a rank is k_embed_reg_unit on its own block of slices (unfused kernels + all-gather)."""
    from fsw_gnn_amd import _lib
    from fsw_gnn_amd.dist import slice_partition
    L = _lib.lib()
    emb = conv.fsw_embed
    ka, kb = slice_partition(emb.nSlices, world)[rank]
    Sl = kb - ka
    stream = torch.cuda.current_stream(dev).cuda_stream
    graph = conv.build_graph(ei, n)
    st = graph.stats()
    ldp = (Sl + 63) // 64 * 64
    Xp = torch.empty((n, ldp), dtype=torch.float32, device=dev)
    V, fr = emb.projVecs.detach()[ka:kb], emb.freqs.detach()[ka:kb]
    table = torch.empty((int(L.fsw_unit_table_rows(32)), ldp), dtype=torch.float32, device=dev)
    out = torch.empty((n, 1 + Sl), dtype=torch.float32, device=dev)
    _lib.check(L.fsw_project_f32(x.data_ptr(), n, D_FEAT, D_FEAT, V.data_ptr(), Sl, D_FEAT, Xp.data_ptr(), ldp, None, 0, None, stream), "project")
    _lib.check(L.fsw_unit_coeff_table(fr.data_ptr(), Sl, 32, table.data_ptr(), ldp, stream), "table")
    a = emb.make_args(graph, st, Xp, ldp, fr, Sl, table, out.data_ptr(), out.stride(0), None, 1.0, 1)
    a.num_zero_rows = 0
    kms = timed_ms(lambda: _lib.check(L.fsw_embed_f32(ctypes.byref(a), stream), "embed"), reps, dev)
    edges = int(graph.rowptr[-1])
    alg_bytes = 4.0 * edges * Sl + 4.0 * edges + 8.0 * n + 4.0 * st[_lib.STAT_NUM_REG] * (Sl + 1)
    return {"bound": "hbm", "kernel": "k_embed_reg_unit (rank 0, %d of %d slices)" % (Sl, emb.nSlices),
            "achieved": alg_bytes / (kms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": alg_bytes / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_launch": alg_bytes,
            "ms_per_launch": kms}


def node_sharded_kernel_roofline(conv, x, ei, n, reps, dev, rank, world):
    """N > 1, recipient-row sharding: the dominant kernel of a rank is k_conv_fused_unit over its block of rows."""
    from fsw_gnn_amd import _lib
    from fsw_gnn_amd.graph import build_csr
    emb = conv.fsw_embed
    per = -(-n // world)
    r0 = min(rank * per, n)
    nl = min(r0 + per, n) - r0
    mine = (ei[1] >= r0) & (ei[1] < r0 + nl)
    graph = build_csr(ei[1][mine] - r0, ei[0][mine], None, nl, n, want_invperm=True)
    wq, w2 = conv._fused_weight()
    prepared = emb.prepare(x, graph)
    st = prepared["stats"]
    assert st[_lib.STAT_NUM_LDS] == 0 and st[_lib.STAT_NUM_GLOBAL] == 0
    H = conv.mlp[0].out_features
    yin = torch.empty((nl, H), dtype=torch.float32, device=dev)
    y = torch.empty((nl, H), dtype=torch.float32, device=dev)
    kms = timed_ms(lambda: conv._fused_linear(graph, prepared, 1.0, wq, yin, y), reps, dev)
    edges = int(graph.rowptr[-1])
    alg_bytes = 4.0 * edges * emb.nSlices + 4.0 * edges + 8.0 * nl + 8.0 * nl * H
    return {"bound": "hbm", "kernel": "k_conv_fused_unit (rank 0: %d of %d rows)" % (nl, n),
            "achieved": alg_bytes / (kms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": alg_bytes / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_launch": alg_bytes,
            "ms_per_launch": kms}


def cpu_baseline(x, ei, conv, n, nslices, max_threads):
    """C oracle (port of the reference algorithm) on the host cores: slices [0, nslices) of the same workload."""
    from oracle import c_oracle as C
    from oracle import fsw_oracle as O
    eih = ei.cpu().numpy()
    order = np.argsort(eih[1], kind="stable")
    rowptr = np.concatenate([[0], np.cumsum(np.bincount(eih[1], minlength=n))]).astype(np.int64)
    col = eih[0][order]
    xh = x.cpu().numpy()
    V = conv.fsw_embed.projVecs.detach().cpu().numpy()
    fr = conv.fsw_embed.freqs.detach().cpu().numpy()
    threads = min(C.max_threads(), max_threads)        # the GPU box's CPU share for one GPU is 16 cores
    t0 = time.time()
    C.embed(xh, rowptr, col, None, V, fr, s0=0, s1=nslices, nthreads=threads)
    secs = time.time() - t0
    return {"value": float(eih.shape[1]) * nslices / secs, "unit": "edges*slices/sec", "cores": threads, "kind": "port",
            "sample": "slices 0..%d of %d, all %d rows / %d edges, embedding core only (projection + sort + cumsum + readout), "
                      "oracle/fsw_oracle.c with OpenMP, %.1f s" % (nslices - 1, V.shape[0], n, eih.shape[1], secs)}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % args.gpus)
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    dev = torch.device("cuda", local_rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    if world > 1:
        # RCCL ("nccl") over xGMI, one rank per GPU.  FSW_BENCH_BACKEND=gloo is a functional rehearsal on a box with
        # fewer GPUs than ranks (ranks then share a device; numbers from it mean nothing).
        backend = os.environ.get("FSW_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from fsw_gnn_amd import FSW_conv
    n, E = args.nodes, args.edges
    x, ei = make_inputs(n, E, dev)
    torch.manual_seed(4321)
    conv = FSW_conv(D_FEAT, OUT_CH, embed_dim=EMBED_DIM, device=dev)
    if args.no_fuse:
        conv.fuse_linear = False
    if world > 1:
        if args.shard == "nodes" and not args.no_fuse:
            conv.enable_node_parallel(None)
        else:
            conv.enable_slice_parallel(None)
    S = conv.fsw_embed.nSlices
    keys = ei[1] * n + ei[0]
    e_coalesced = int(torch.unique(keys).numel())          # E' of SURVEY 8(d): edges after the reference's coalesce()
    del keys

    def step():
        with torch.no_grad():
            return conv(x, ei)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        y = step()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax)
    assert torch.isfinite(y).all()

    ms_per_step = elapsed / args.steps * 1e3
    value = float(e_coalesced) * S * args.steps / elapsed
    result = {
        "metric": "edges*slices/sec FSW_conv forward, 1M-node/10M-edge, 256 slices", "value": value,
        "unit": "edges*slices/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE config 3: ER multigraph %d nodes / %d edges (%d after coalesce), %d feat, %d slices, "
                               "FSW_conv(%d->%d, embed_dim=%d), full forward incl. CSR build and Linear layer"
                               % (n, E, e_coalesced, D_FEAT, S, D_FEAT, OUT_CH, EMBED_DIM),
                   "nodes": n, "edges": E, "edges_coalesced": e_coalesced, "slices": S, "features": D_FEAT,
                   "parallelism": ("single GPU" if world == 1 else
                                   "recipient-row shard x%d, all-gather of the output rows" % world if getattr(conv, "_node_parallel", False)
                                   else "slice-shard x%d, all-gather of the embedding" % world)},
        # fraction of the 8 TB/s roofline for the WHOLE forward at SURVEY 8(d)'s 5.13 B per edge*slice (incl. CSR build)
        "path_roofline_frac": value * 5.13 / (HBM_PEAK_GBS * 1e9),
    }
    if rank == 0 and world == 1:
        roof, ms = dominant_kernel_roofline(conv, x, ei, n, e_coalesced, args.kernel_reps, dev)
        result["roofline"] = roof
        result["stage_ms"] = ms
        if not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(x, ei, conv, n, args.cpu_slices, args.cpu_threads)
    if world > 1:
        fn = node_sharded_kernel_roofline if getattr(conv, "_node_parallel", False) else sharded_kernel_roofline
        roof = fn(conv, x, ei, n, max(3, args.kernel_reps // 2), dev, rank, world)   # every rank runs it
        if rank == 0:
            result["roofline"] = roof
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
