#!/usr/bin/env python3
"""Headline benchmark: edges*slices/sec of FSW_conv.forward on the BASELINE config-3 workload.

    python bench.py --gpus N --steps K --warmup W
        N > 1 without WORLD_SIZE in the environment: bench.py starts its own ranks -- a CHILD process
        `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <same args>`
        (one rank per GPU over RCCL), before anything in this process touches the GPU -- relays rank 0's JSON line as the last
        stdout line and exits with the child's code.  Launched by an outer torch.distributed.run it runs as a rank as before.

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): ER-style directed multigraph with
1,000,000 nodes / 10,000,000 edges, 128 features, FSW_conv(128 -> 128, embed_dim = 257) = 256 slices + degree
column, 'spread' frequencies, one Linear + LeakyReLU layer -- synthetic data, random-init weights.
A step is ONE full FSW_conv.forward(x, edge_index): CSR build from the int64 edge_index (rebuilt every step like
the reference, fsw_conv.py:352 -- nothing is cached), fp32-MFMA projection, fused neighbourhood sort / cumulative
sum / Fourier readout, concat with the vertex features and the Linear layer.  Inputs are resident in HBM before
the timed region.  With N > 1 the SLICE axis is sharded over the ranks (BASELINE north_star; 256 / N slices each, total
work fixed -> strong scaling) and EVERY form of fsw_gnn_amd/dist.py is timed over the same W + K steps: 'gather' (north_star's
all-gather of the embedding), 'consumer' (every rank multiplies its slice block by its columns of the first Linear layer inside
the fused kernel, the n x 128 partial sums are reduce-scattered, finished rows all-gathered) and 'exchange' (one all-to-all of
slice blocks to the row owners); node-range chunks pipelined; the headline is the fastest form that leaves the full output on
every rank.  Each form carries compute_ms (the same step with the collectives replaced by local copies), collective_ms /
collective_GBps (the form's collectives alone on same-size buffers) and
bytes per rank.  Recipient-row sharding (not the contracted partition) is timed in the same run as `other_partitions` and is
never the headline; --shard nodes makes it the only thing that runs.

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
import socket
import subprocess
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_NODES, N_EDGES, D_FEAT, N_SLICES, OUT_CH = 1_000_000, 10_000_000, 128, 256, 128
EMBED_DIM = N_SLICES + 1            # + the degree column (FSW_conv: encode_vertex_degrees)
WEAK_SLICES_PER_GPU = 128           # BASELINE config 4: 1024 slices sharded 128 per GPU across 8
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


FORM_ARGS = {"gather": ("gather", "replicated"), "consumer": ("consumer", "replicated"), "consumer_sharded": ("consumer", "sharded"),
             "exchange": ("exchange", "replicated"), "exchange_sharded": ("exchange", "sharded")}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--nodes", type=int, default=N_NODES)
    ap.add_argument("--edges", type=int, default=N_EDGES)
    ap.add_argument("--slices", type=int, default=N_SLICES,
                    help="slices of the layer (embed_dim = slices + 1).  256 = BASELINE config 3 (the metric's configuration); "
                         "1024 = config 4 (128 slices per GPU at 8 GPUs)")
    ap.add_argument("--kernel-reps", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-segcumsum", action="store_true", help="skip the stand-alone segmented-cumsum throughput leg")
    ap.add_argument("--no-weak", action="store_true", help="skip the weak-scaling leg (128 slices per GPU, BASELINE config 4's shape)")
    ap.add_argument("--no-other-partitions", action="store_true",
                    help="N > 1: skip the recipient-row sharded leg (reported beside the slice-axis forms, never the headline)")
    ap.add_argument("--cpu-slices", type=int, default=256)
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--no-fuse", action="store_true", help="force the unfused kernels (embedding written to HBM, torch Linear)")
    ap.add_argument("--shard", choices=("slices", "nodes"), default="slices",
                    help="N > 1: 'slices' = BASELINE north_star's slice-axis shard (default); 'nodes' = every rank runs the fused "
                         "layer on its block of recipient rows, all-gather of the 128-wide output (extra, not the contracted partition)")
    ap.add_argument("--forms", default="gather,consumer,exchange",
                    help="N > 1, slice shard: comma-separated forms to time, each over the same K steps: 'gather' = north_star's "
                         "all-gather of the embedding, 'consumer' = sharded first Linear layer + reduce-scatter + all-gather of the "
                         "output rows, 'exchange' = all-to-all of the slice blocks to the row owners + tail on the owned rows + all-gather of "
                         "the output rows, 'consumer_sharded' / 'exchange_sharded' = the same without the final all-gather (rows stay "
                         "with their owner). "
                         "The headline value is the fastest form that returns the full output on every rank.")
    ap.add_argument("--mode", choices=("auto", "consumer", "gather", "exchange"), default=None,
                    help="(older spelling) time ONE form: consumer / gather / exchange; auto = consumer")
    ap.add_argument("--chunks", type=int, default=0, help="node-range chunks of the multi-GPU pipeline (0 = by size)")
    ap.add_argument("--output", choices=("replicated", "sharded"), default="replicated",
                    help="with --mode consumer: 'sharded' stops after the reduce-scatter (every rank keeps its finished rows)")
    args = ap.parse_args(argv)
    if args.mode is not None:
        base = "consumer" if args.mode == "auto" else args.mode
        args.forms = base + ("_sharded" if args.output == "sharded" and base != "gather" else "")
    if args.no_fuse:
        args.forms = "gather,exchange"
    args.forms = [f for f in args.forms.split(",") if f]
    for f in args.forms:
        if f not in FORM_ARGS:
            ap.error("unknown form %r" % f)
    return args


# ---------------------------------------------------------------------------------------------------------------------
# self-launch: `python bench.py --gpus N` with no launcher around it
# ---------------------------------------------------------------------------------------------------------------------
def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_command(argv, gpus, port, script=None, python=None):
    """The child command that runs this script as `gpus` ranks of one node (one rank per GPU, RCCL): exactly the line the
    driver uses for N > 1.  argv = this process's arguments, passed through unchanged."""
    return [python or sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(int(gpus)),
            "--master-addr", "127.0.0.1", "--master-port", str(int(port)), script or os.path.abspath(__file__)] + list(argv)


def self_launch(args, argv, script=None):
    """Parent of a self-launched N > 1 run.  Nothing here initialises the GPU (no torch.cuda call): the ranks are children.
    Rank 0's JSON line is relayed as the LAST stdout line, everything else the children print goes to stderr."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this pool (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = launch_command(argv, args.gpus, free_port(), script=script)
    print("bench.py: starting %d ranks: %s" % (args.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for out in proc.stdout:
        if out.startswith('{"metric"'):
            line = out.rstrip("\n")
        else:
            sys.stderr.write(out)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    return rc if rc != 0 or line is not None else 1


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
        kms = timed_ms(lambda: _lib.check(L.fsw_conv_fused_f32(ctypes.byref(a), wq.data_ptr(), wq.shape[1], None, H, None, 0, 0, 0, 0.0,
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
    fn, err = None, None
    try:          # allocation is the part that can fail on one rank alone
        if stats.get("mode") == "consumer":
            rows = -(-n // world) * world
            P = torch.zeros((rows, H), dtype=torch.float32, device=dev)
            R = torch.empty((rows // world, H), dtype=torch.float32, device=dev)

            def fn():
                D._reduce_scatter(R, P, None, False)
                if "all_gather" in stats.get("collective", ""):
                    D._all_gather(P, R, None, False)
        elif stats.get("mode") == "exchange":
            rows = -(-n // world) * world
            A = torch.zeros((world, rows // world, width), dtype=torch.float32, device=dev)
            B = torch.empty_like(A)
            R = torch.zeros((rows // world, H), dtype=torch.float32, device=dev)
            Y = torch.empty((rows, H), dtype=torch.float32, device=dev)

            def fn():
                D._all_to_all(B, A, None, False)
                if "all_gather" in stats.get("collective", ""):
                    D._all_gather(Y, R, None, False)
        else:
            loc = torch.zeros((n, width), dtype=torch.float32, device=dev)
            flat = torch.empty((world * n, width), dtype=torch.float32, device=dev)

            def fn():
                D._all_gather(flat, loc, None, False)
    except Exception as e:   # noqa: BLE001
        err = "%s: %s" % (type(e).__name__, e)
    if not all_ranks_ok(err is None, dev, world):
        return {"collective_error": err or "another rank could not allocate the buffers"}
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


def timed_steps(step, warmup, steps, dev, world):
    """W untimed steps, then EXACTLY K steps bracketed by barrier + synchronize on both sides; seconds, MAX over ranks."""
    y = None
    for _ in range(warmup):
        y = step()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
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
    return elapsed, y


def all_ranks_ok(ok, dev, world):
    """True only if EVERY rank passes ok = True: called before a leg that contains collectives, so that a rank which failed
    while preparing it (out of memory, ...) makes all ranks skip the leg instead of leaving the others blocked in RCCL."""
    if world == 1:
        return bool(ok)
    t = torch.tensor([0 if ok else 1], dtype=torch.int32, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(t) == 0





def world_record(dev, world, rank, local_rank):
    """What the process group really is: proves in the record that RCCL saw N ranks, each on its own device."""
    me = {"rank": rank, "local_rank": local_rank, "device_index": dev.index, "device_name": torch.cuda.get_device_name(dev),
          "pid": os.getpid(), "host": socket.gethostname()}
    if world == 1:
        return {"world_size": 1, "backend": None, "ranks": [me]}
    ranks = [None] * world
    dist.all_gather_object(ranks, me)
    ones = torch.ones(1, dtype=torch.float32, device=dev)
    dist.all_reduce(ones)
    return {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "ranks": ranks, "allreduce_of_ones": float(ones),
            "distinct_devices": len({(r["host"], r["device_index"]) for r in ranks}),
            "launcher": "self (bench.py started torch.distributed.run)" if os.environ.get("FSW_BENCH_SELF_LAUNCHED") else "outer torch.distributed.run"}


def weak_scaling_leg(args, x, ei, e_coalesced, dev, world, rank):
    """BASELINE config 4's shape: the same graph with WEAK_SLICES_PER_GPU slices per GPU (1024 at 8 GPUs), i.e. per-GPU work
    fixed as N grows.  N > 1: the consumer form (its collectives do not grow with the slice count)."""
    from fsw_gnn_amd import FSW_conv
    S = WEAK_SLICES_PER_GPU * world
    ok, conv, st = True, None, {}
    try:
        torch.manual_seed(4321)
        conv = FSW_conv(D_FEAT, OUT_CH, embed_dim=S + 1, device=dev)
        if world > 1:
            conv.enable_slice_parallel(None, mode="consumer", chunks=args.chunks or None, output="replicated", stats=st)
    except Exception as e:   # noqa: BLE001
        ok, err = False, "%s: %s" % (type(e).__name__, e)
    def step():
        with torch.no_grad():
            return conv(x, ei)

    if ok and world == 1:
        try:
            step()                      # configuration errors show here, before the timed region
        except Exception as e:   # noqa: BLE001
            ok, err = False, "%s: %s" % (type(e).__name__, e)
    if not all_ranks_ok(ok, dev, world):
        return {"error": err if not ok else "another rank failed to set the leg up"}
    try:
        # N > 1: a configuration error raises on every rank alike, before the step's first collective
        elapsed, _ = timed_steps(step, args.warmup, args.steps, dev, world)
    except NotImplementedError as e:
        return {"error": "NotImplementedError: %s" % e}
    return {"scaling": "weak", "slices_per_gpu": WEAK_SLICES_PER_GPU, "slices": S, "n_gpus": world,
            "ms_per_step": elapsed / args.steps * 1e3, "value": float(e_coalesced) * S * args.steps / elapsed,
            "unit": "edges*slices/sec", "form": st.get("mode", "single GPU"), "collective": st.get("collective"),
            "bytes_received_per_rank": st.get("bytes_received_per_rank")}


def recipient_rows_leg(args, x, ei, e_coalesced, S, dev, world, rank):
    """Outside north_star's slice-axis contract, for comparison in the same record: the RECIPIENT rows sharded over the ranks
    (FSW_conv.enable_node_parallel: projection replicated, CSR + fused kernel on n / N rows, ONE all-gather of the output
    rows).  Same W + K steps and the same max-over-ranks clock as the forms above; compute_ms = the step without the collective."""
    from fsw_gnn_amd import FSW_conv
    ok, err, conv = True, None, None
    try:
        torch.manual_seed(4321)
        conv = FSW_conv(D_FEAT, OUT_CH, embed_dim=S + 1, device=dev)
        conv.enable_node_parallel(None)
    except Exception as e:   # noqa: BLE001
        ok, err = False, "%s: %s" % (type(e).__name__, e)
    if not all_ranks_ok(ok, dev, world):
        return {"error": err or "another rank failed to set the leg up"}

    def step():
        with torch.no_grad():
            return conv(x, ei)

    elapsed, y = timed_steps(step, args.warmup, args.steps, dev, world)
    n, H = x.shape[0], conv.mlp[0].out_features
    per = -(-n // world)
    out = {"partition": "recipient rows (not the contracted slice-axis shard)", "collective": "all_gather", "output": "replicated",
           "ms_per_step": elapsed / args.steps * 1e3, "value": float(e_coalesced) * S * args.steps / elapsed,
           "bytes_sent_per_rank": 4 * per * H * (world - 1), "bytes_received_per_rank": 4 * per * H * (world - 1),
           "finite": bool(torch.isfinite(y).all())}
    cms, err = 0.0, None
    try:
        with torch.no_grad():
            cms = timed_ms(lambda: conv._forward_node_parallel(x, ei, _emulate=(rank, world)), max(3, args.kernel_reps // 4), dev)
    except Exception as e:   # noqa: BLE001
        err = "%s: %s" % (type(e).__name__, e)
    if all_ranks_ok(err is None, dev, world):
        t = torch.tensor([cms], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        out["compute_ms"] = float(t)
    else:
        out["compute_ms_error"] = err or "another rank failed"
    return out


def main():
    argv = sys.argv[1:]
    args = parse(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        os.environ["FSW_BENCH_SELF_LAUNCHED"] = "1"
        sys.exit(self_launch(args, argv))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("bench.py --gpus %d inside a launcher with WORLD_SIZE=%d: the two must agree" % (args.gpus, world))
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    dev = torch.device("cuda", local_rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    if world > 1:
        # RCCL ("nccl") over xGMI, one rank per GPU.  FSW_BENCH_BACKEND=gloo is a functional rehearsal on a box with
        # fewer GPUs than ranks (ranks then share a device; numbers from it mean nothing).
        import datetime
        backend = os.environ.get("FSW_BENCH_BACKEND", "nccl")
        tmo = datetime.timedelta(seconds=300)    # a rank that died must not leave the others waiting for the default 10 minutes
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=tmo)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=tmo)

    from fsw_gnn_amd import FSW_conv
    from fsw_gnn_amd import dist as D
    n, E = args.nodes, args.edges
    embed_dim = args.slices + 1
    x, ei = make_inputs(n, E, dev)
    torch.manual_seed(4321)
    conv = FSW_conv(D_FEAT, OUT_CH, embed_dim=embed_dim, device=dev)
    if args.no_fuse:
        conv.fuse_linear = False
    S = conv.fsw_embed.nSlices
    keys = ei[1] * n + ei[0]
    e_coalesced = int(torch.unique(keys).numel())          # E' of SURVEY 8(d): edges after the reference's coalesce()
    del keys
    wrec = world_record(dev, world, rank, local_rank)

    def step():
        with torch.no_grad():
            return conv(x, ei)

    forms, sp_stats = {}, {}
    node_parallel = world > 1 and args.shard == "nodes" and not args.no_fuse
    if world == 1:
        elapsed, y = timed_steps(step, args.warmup, args.steps, dev, world)
        assert torch.isfinite(y).all()
        headline = None
    elif node_parallel:
        conv.enable_node_parallel(None)
        elapsed, y = timed_steps(step, args.warmup, args.steps, dev, world)
        assert torch.isfinite(y).all()
        headline = None
    else:
        # every form over the same W + K steps; the headline is the fastest one that leaves the full output on every rank
        for form in args.forms:
            mode, output = FORM_ARGS[form]
            st = {}
            conv.enable_slice_parallel(None, mode=mode, chunks=args.chunks or None, output=output, stats=st)
            el, y = timed_steps(step, args.warmup, args.steps, dev, world)
            assert torch.isfinite(y[0] if isinstance(y, tuple) else y).all()
            forms[form] = {"ms_per_step": el / args.steps * 1e3, "value": float(e_coalesced) * S * args.steps / el,
                           "form_taken": st.get("mode"), "collective": st.get("collective"), "output": output,
                           "bytes_sent_per_rank": st.get("bytes_sent_per_rank"),
                           "bytes_received_per_rank": st.get("bytes_received_per_rank"), "_elapsed": el, "_stats": st}
            del y
        full = [f for f in args.forms if FORM_ARGS[f][1] == "replicated"] or list(args.forms)
        headline = min(full, key=lambda f: forms[f]["_elapsed"])
        elapsed, sp_stats = forms[headline]["_elapsed"], forms[headline]["_stats"]

    ms_per_step = elapsed / args.steps * 1e3
    value = float(e_coalesced) * S * args.steps / elapsed
    cfg_name = {256: "BASELINE config 3", 1024: "BASELINE config 4 (1024 slices)"}.get(S, "config-3 graph, %d slices" % S)
    result = {
        "metric": "edges*slices/sec FSW_conv forward, 1M-node/10M-edge, %d slices" % S, "value": value,
        "unit": "edges*slices/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s: ER multigraph %d nodes / %d edges (%d after coalesce), %d feat, %d slices, "
                               "FSW_conv(%d->%d, embed_dim=%d), full forward incl. CSR build and Linear layer"
                               % (cfg_name, n, E, e_coalesced, D_FEAT, S, D_FEAT, OUT_CH, embed_dim),
                   "nodes": n, "edges": E, "edges_coalesced": e_coalesced, "slices": S, "features": D_FEAT,
                   "parallelism": ("single GPU" if world == 1 else
                                   "recipient-row shard x%d, all-gather of the output rows" % world if node_parallel
                                   else "slice-axis shard x%d (%d slices per rank), headline = %s form: %s" % (
                                       world, S // world, headline, sp_stats.get("collective", "?")))},
        # fraction of the 8 TB/s roofline for the WHOLE forward at SURVEY 8(d)'s 5.13 B per edge*slice (incl. CSR build)
        "path_roofline_frac": value * 5.13 / (HBM_PEAK_GBS * 1e9),
        "world": wrec,
    }
    if rank == 0 and world == 1:
        roof, ms = dominant_kernel_roofline(conv, x, ei, n, e_coalesced, args.kernel_reps, dev)
        result["roofline"] = roof
        result["stage_ms"] = ms
    if world > 1 and not node_parallel:
        # the legs below only decorate the line (roofline of the sharded kernel, compute / collective split per form).  Local
        # work sits in try blocks; before anything with a collective in it the ranks agree that all of them got there.
        H = conv.mlp[0].out_features
        width = 1 + max(b - a for a, b in D.slice_partition(S, world))
        reps = max(3, args.kernel_reps // 4)
        for form in args.forms:
            f = forms[form]
            st = f.pop("_stats")
            f.pop("_elapsed")
            mode, output = FORM_ARGS[form]
            conv.enable_slice_parallel(None, mode=mode, chunks=args.chunks or None, output=output, stats={})
            cms, err = 0.0, None
            try:       # the same step with every collective replaced by a local copy = this rank's compute (no RCCL call inside)
                D.COLLECTIVES_ENABLED = False
                cms = timed_ms(step, reps, dev)
            except Exception as e:   # noqa: BLE001
                err = "%s: %s" % (type(e).__name__, e)
            finally:
                D.COLLECTIVES_ENABLED = True
            if all_ranks_ok(err is None, dev, world):
                t = torch.tensor([cms], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                f["compute_ms"] = float(t)
            else:
                f["compute_ms_error"] = err or "another rank failed"
            try:
                coll = collectives_alone(st, n, H, width, world, reps, dev)
            except Exception as e:   # noqa: BLE001 -- collectives_alone agrees across ranks before its first collective
                coll = {"collective_error": "%s: %s" % (type(e).__name__, e)}
            if coll:
                f.update(coll)
        result["forms"] = forms
        result["headline_form"] = headline
        for k in ("compute_ms", "collective_ms", "collective_GBps_per_rank", "bytes_sent_per_rank", "bytes_received_per_rank"):
            if k in forms[headline]:
                result[k] = forms[headline][k]
        result["slice_parallel"] = {"mode": forms[headline].get("form_taken"), "collective": forms[headline].get("collective"),
                                    "output": forms[headline]["output"]}
        try:
            roof = sharded_kernel_roofline(conv, x, ei, n, max(3, args.kernel_reps // 2), dev, rank, world,
                                           forms[headline].get("form_taken") == "consumer")   # every rank runs it; no collective
            if rank == 0:
                result["roofline"] = roof
        except Exception as e:   # noqa: BLE001 -- reported in the line, the headline number stands
            result["extras_error"] = "%s: %s" % (type(e).__name__, e)
    elif world > 1:
        try:
            roof = node_sharded_kernel_roofline(conv, x, ei, n, max(3, args.kernel_reps // 2), dev, rank, world)
            if rank == 0:
                result["roofline"] = roof
        except Exception as e:   # noqa: BLE001
            result["extras_error"] = "%s: %s" % (type(e).__name__, e)
    if world > 1 and not node_parallel and not args.no_fuse and not args.no_other_partitions:
        result["other_partitions"] = {"recipient_rows": recipient_rows_leg(args, x, ei, e_coalesced, S, dev, world, rank)}
    if not args.no_weak and not node_parallel and args.slices == N_SLICES:
        del conv
        torch.cuda.empty_cache()
        result["weak_scaling"] = weak_scaling_leg(args, x, ei, e_coalesced, dev, world, rank)
    if rank == 0 and world == 1:
        if not args.no_segcumsum:
            torch.cuda.empty_cache()
            try:
                result["segcumsum"] = segcumsum_leg(dev)
            except Exception as e:   # noqa: BLE001 -- a side leg must not lose the line
                result["segcumsum"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if not args.no_cpu_baseline:
            try:
                torch.manual_seed(4321)
                conv = FSW_conv(D_FEAT, OUT_CH, embed_dim=embed_dim, device=dev)
                result["cpu_baseline"] = cpu_baseline(x, ei, conv, n, min(args.cpu_slices, S), args.cpu_threads)
            except Exception as e:   # noqa: BLE001
                result["cpu_baseline"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
