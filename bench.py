#!/usr/bin/env python3
"""Headline benchmark: edges*slices/sec of FSW_conv.forward on the BASELINE config-3 workload.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): ER-style directed multigraph with
1,000,000 nodes / 10,000,000 edges, 128 features, FSW_conv(128 -> 128, embed_dim = 257) = 256 slices + degree
column, 'spread' frequencies, one Linear + LeakyReLU layer -- synthetic data, random-init weights.
A step is ONE full FSW_conv.forward(x, edge_index): CSR build from the int64 edge_index (rebuilt every step like
the reference, fsw_conv.py:352 -- nothing is cached), fp32-MFMA projection, fused neighbourhood sort / cumulative
sum / Fourier readout, concat with the vertex features and the Linear layer.  Inputs are resident in HBM before
the timed region.  With N > 1 the SLICE axis is sharded over the ranks (BASELINE north_star; 256 / N slices each, total
work fixed -> strong scaling): by default the sharded-consumer form of fsw_gnn_amd/dist.py (every rank multiplies its
slice block by its columns of the first Linear layer inside the fused kernel, the n x 128 partial sums are
reduce-scattered, finished rows all-gathered, node-range chunks pipelined); --mode gather runs the contracted
all-gather of the embedding instead.  The line then also carries compute_ms (the same step with the collectives
replaced by local copies), collective_ms / collective_GBps (the step's collectives alone on same-size buffers) and
bytes per rank.  --shard nodes (recipient-row sharding, not the contracted partition) stays behind its flag.

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
    ap.add_argument("--no-segcumsum", action="store_true", help="skip the stand-alone segmented-cumsum throughput leg")
    ap.add_argument("--cpu-slices", type=int, default=256)
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--no-fuse", action="store_true", help="force the unfused kernels (embedding written to HBM, torch Linear)")
    ap.add_argument("--shard", choices=("slices", "nodes"), default="slices",
                    help="N > 1: 'slices' = BASELINE north_star's slice-axis shard (default); 'nodes' = every rank runs the fused "
                         "layer on its block of recipient rows, all-gather of the 128-wide output (extra, not the contracted partition)")
    ap.add_argument("--mode", choices=("auto", "consumer", "gather"), default="auto",
                    help="slice shard: 'consumer' = sharded first Linear layer + reduce-scatter (auto picks it), 'gather' = all-gather of the embedding")
    ap.add_argument("--chunks", type=int, default=0, help="node-range chunks of the multi-GPU pipeline (0 = by size)")
    ap.add_argument("--output", choices=("replicated", "sharded"), default="replicated",
                    help="consumer form: 'sharded' stops after the reduce-scatter (every rank keeps its finished rows)")
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
    traffic, traffic_source = None, None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic_latest.json")
    if os.path.isfile(tpath):
        try:
            tj = json.load(open(tpath))
            traffic = tj.get(kernel + "_hbm_bytes_per_launch")
            # NOT measured in this run: PMC passes are separate rocprofv3 runs (tools/refresh_profiles.sh); say where from
            traffic_source = dict(tj.get("source", {}), file="profiles/pmc_traffic_latest.json")
        except Exception:
            traffic = None
    roof = {"bound": "hbm", "kernel": kernel, "achieved": alg_bytes / secs / 1e9, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": alg_bytes / secs / 1e9 / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
            "algorithmic_bytes_per_launch": alg_bytes, "ms_per_launch": kms,
            "bytes_per_edge_slice": alg_bytes / (float(e_coalesced) * S)}
    return roof, ms


def sharded_kernel_roofline(conv, x, ei, n, reps, dev, rank, world, consumer):
    """N > 1: the dominant kernel of a rank over its own block of slices -- k_conv_fused_unit (sharded-consumer form:
    gather of the block + the n x H partial sums written) or k_embed_reg_unit (gather form: block of the embedding written)."""
    from fsw_gnn_amd import _lib
    from fsw_gnn_amd.dist import slice_partition
    L = _lib.lib()
    emb = conv.fsw_embed
    ka, kb = slice_partition(emb.nSlices, world)[rank]
    Sl = kb - ka
    stream = torch.cuda.current_stream(dev).cuda_stream
    graph = conv.build_graph(ei, n)
    prepared = emb.prepare(x, graph, slice_range=(ka, kb))
    st = prepared["stats"]
    fr = emb.freqs.detach()[ka:kb]
    edges = int(graph.rowptr[-1])
    H = conv.mlp[0].out_features
    if consumer:
        hm = 1 if rank == 0 else 0
        wq, _ = conv._fused_weight(col0=0 if rank == 0 else 1 + ka, K=hm + Sl, want_w2=False)
        y = torch.empty((n, H), dtype=torch.float32, device=dev)
        a = emb.make_args(graph, st, prepared["Xp"], prepared["ldp"], fr, Sl, prepared["table"], None, 0, None, 1.0, hm, slice_offset=ka)
        kms = timed_ms(lambda: _lib.check(L.fsw_conv_fused_f32(ctypes.byref(a), wq.data_ptr(), wq.shape[1], None, H, None, 0, 0, 0.0,
                                                               y.data_ptr(), y.stride(0), stream), "fused"), reps, dev)
        alg_bytes = 4.0 * edges * Sl + 4.0 * edges + 8.0 * n + 4.0 * n * H
        kernel = "k_conv_fused_unit"
    else:
        out = torch.empty((n, 1 + Sl), dtype=torch.float32, device=dev)
        a = emb.make_args(graph, st, prepared["Xp"], prepared["ldp"], fr, Sl, prepared["table"], out.data_ptr(), out.stride(0), None, 1.0, 1,
                          slice_offset=ka)
        a.num_zero_rows = 0
        kms = timed_ms(lambda: _lib.check(L.fsw_embed_f32(ctypes.byref(a), stream), "embed"), reps, dev)
        alg_bytes = 4.0 * edges * Sl + 4.0 * edges + 8.0 * n + 4.0 * st[_lib.STAT_NUM_REG] * (Sl + 1)
        kernel = "k_embed_reg_unit"
    return {"bound": "hbm", "kernel": "%s (rank 0, %d of %d slices)" % (kernel, Sl, emb.nSlices),
            "achieved": alg_bytes / (kms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": alg_bytes / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_launch": alg_bytes,
            "ms_per_launch": kms}


def collectives_alone(stats, n, H, width, world, reps, dev):
    """The collectives of one step on same-size dummy buffers, nothing else on the GPU: milliseconds and the bus rate
    (bytes a rank receives per second).  Un-overlapped: the step hides part of it under the kernels."""
    if not stats:
        return None
    from fsw_gnn_amd import dist as D
    if stats.get("mode") == "consumer":
        rows = -(-n // world) * world
        P = torch.zeros((rows, H), dtype=torch.float32, device=dev)
        R = torch.empty((rows // world, H), dtype=torch.float32, device=dev)

        def fn():
            D._reduce_scatter(R, P, None, False)
            if "all_gather" in stats.get("collective", ""):
                D._all_gather(P, R, None, False)
    else:
        loc = torch.zeros((n, width), dtype=torch.float32, device=dev)
        flat = torch.empty((world * n, width), dtype=torch.float32, device=dev)

        def fn():
            D._all_gather(flat, loc, None, False)
    ms = timed_ms(fn, reps, dev)
    t = torch.tensor([ms], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ms = float(t)
    return {"collective_ms": ms, "collective_GBps_per_rank": stats["bytes_received_per_rank"] / (ms * 1e-3) / 1e9}


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


def segcumsum_leg(dev, reps=5, elems=256_000_000, mean_seg=10.0):
    """The stand-alone segmented cumulative sum (fsw_segcumsum, the replacement of the reference's only native kernel,
    fsw_embedding.cu) on float32 values / int64 ids with neighbourhood-sized segments: GB/s of algorithmic bytes
    (value read + id read + value written = 16 B per element) against the HBM peak."""
    from fsw_gnn_amd import segcumsum
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    ids = torch.cumsum(torch.rand(elems, device=dev, generator=g) < (1.0 / mean_seg), 0, dtype=torch.int64)
    vals = torch.rand(elems, device=dev, generator=g)
    ms = timed_ms(lambda: segcumsum(vals, ids), reps, dev)
    gbs = elems * 16.0 / ms / 1e6
    return {"kernel": "k_segscan_chained", "elements": elems, "mean_segment": mean_seg, "values": "f32", "ids": "i64", "ms": ms,
            "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
            "algorithmic_bytes_per_element": 16}


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
    res = {"value": float(eih.shape[1]) * nslices / secs, "unit": "edges*slices/sec", "cores": threads, "kind": "port",
           "sample": "slices 0..%d of %d, all %d rows / %d edges, embedding core only (projection + sort + cumsum + readout), "
                     "oracle/fsw_oracle.c with OpenMP, %.1f s" % (nslices - 1, V.shape[0], n, eih.shape[1], secs)}
    # the reference's own CPU path (fsw_embedding.py, CUDA library disabled) cannot travel to this box; its figure was
    # measured once in the build container by oracle/make_goldens.py and is quoted here with its provenance
    try:
        rt = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_timings.json")))
        secs_ref = float(rt["er1m_fp64_forward_seconds"])
        res["reference_cpu"] = {
            "value": float(rt["er1m_edges_coalesced"]) * rt["er1m_slices"] / secs_ref, "unit": "edges*slices/sec",
            "cores": rt["cpu_threads"], "seconds": secs_ref,
            "provenance": "the unmodified reference FSW_embedding.forward (device='cpu', load_custom_cuda_lib=False, float64, "
                          "serialize_num_slices=%d) on BASELINE config 3, build container (8 cores, torch %s), "
                          "tests/golden/ref_timings.json written by oracle/make_goldens.py -- NOT measured on this box"
                          % (rt["er1m_serialize_num_slices"], rt["torch"])}
    except Exception:
        pass
    return res


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
    sp_stats = {}
    if world > 1:
        if args.shard == "nodes" and not args.no_fuse:
            conv.enable_node_parallel(None)
        else:
            conv.enable_slice_parallel(None, mode="gather" if args.no_fuse else args.mode, chunks=args.chunks or None,
                                       output=args.output, stats=sp_stats)
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
    assert torch.isfinite(y[0] if isinstance(y, tuple) else y).all()

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
                                   else "slice-axis shard x%d (%d slices per rank), %s form: %s" % (
                                       world, S // world, sp_stats.get("mode", "?"), sp_stats.get("collective", "?")))},
        # fraction of the 8 TB/s roofline for the WHOLE forward at SURVEY 8(d)'s 5.13 B per edge*slice (incl. CSR build)
        "path_roofline_frac": value * 5.13 / (HBM_PEAK_GBS * 1e9),
    }
    if rank == 0 and world == 1:
        roof, ms = dominant_kernel_roofline(conv, x, ei, n, e_coalesced, args.kernel_reps, dev)
        result["roofline"] = roof
        result["stage_ms"] = ms
        if not args.no_segcumsum:
            del y
            torch.cuda.empty_cache()
            try:
                result["segcumsum"] = segcumsum_leg(dev)
            except Exception as e:   # noqa: BLE001 -- a side leg must not lose the line
                result["segcumsum"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if not args.no_cpu_baseline:
            try:
                result["cpu_baseline"] = cpu_baseline(x, ei, conv, n, args.cpu_slices, args.cpu_threads)
            except Exception as e:   # noqa: BLE001
                result["cpu_baseline"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if world > 1:
        # the legs below only decorate the line (roofline of the sharded kernel, compute / collective split): a failure in
        # one of them must not lose the timed result above
        try:
            if getattr(conv, "_node_parallel", False):
                roof = node_sharded_kernel_roofline(conv, x, ei, n, max(3, args.kernel_reps // 2), dev, rank, world)
            else:
                from fsw_gnn_amd import dist as D
                consumer = sp_stats.get("mode") == "consumer"
                roof = sharded_kernel_roofline(conv, x, ei, n, max(3, args.kernel_reps // 2), dev, rank, world, consumer)   # every rank runs it
                # the same step with every collective replaced by a local copy = this rank's compute
                D.COLLECTIVES_ENABLED = False
                cms = timed_ms(step, max(3, args.kernel_reps // 4), dev)
                D.COLLECTIVES_ENABLED = True
                t = torch.tensor([cms], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                result["compute_ms"] = float(t)
                H = conv.mlp[0].out_features
                width = 1 + max(b - a for a, b in D.slice_partition(S, world))
                coll = collectives_alone(sp_stats, n, H, width, world, max(3, args.kernel_reps // 4), dev)
                if coll:
                    result.update(coll)
                result["bytes_sent_per_rank"] = sp_stats.get("bytes_sent_per_rank")
                result["bytes_received_per_rank"] = sp_stats.get("bytes_received_per_rank")
                result["slice_parallel"] = {k: sp_stats.get(k) for k in ("mode", "collective")}
                result["slice_parallel"]["output"] = args.output
            if rank == 0:
                result["roofline"] = roof
        except Exception as e:   # noqa: BLE001 -- reported in the line, the headline number stands
            result["extras_error"] = "%s: %s" % (type(e).__name__, e)
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
