"""CPU-side tests (no GPU): C-ABI symbol export, host logic of the Python mirror, the register sorting networks,
and the multi-GPU slice sharding exercised with gloo at world_size 2."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from tests.conftest import ROOT, golden, relerr

LIB = os.path.join(ROOT, "fsw_gnn_amd", "libfsw_hip.so")


def _need_lib():
    if not os.path.isfile(LIB):
        pytest.skip("libfsw_hip.so not built (run __graft_entry__.build())")


def test_library_exports_every_declared_symbol():
    """Every function declared in include/fsw_hip.h is exported by the shared library and bound by _lib.py."""
    _need_lib()
    header = open(os.path.join(ROOT, "include", "fsw_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b([a-z_0-9]+)\s*\(", header)) - {"defined", "sizeof"}
    declared = {d for d in declared if d.startswith("fsw_") or d.startswith("launch_") or
                d in ("segcumsum_wrapper", "add_block_sums_wrapper", "get_max_threads_per_block")}
    from fsw_gnn_amd import _lib
    L = _lib.lib()                      # loads without a GPU; binds argtypes for every symbol
    assert declared == set(_lib.EXPORTED_SYMBOLS)
    for name in declared:
        assert hasattr(L, name), name
    abi = int(re.search(r"#define FSW_ABI_VERSION (\d+)", header).group(1))
    assert L.fsw_abi_version() == abi == _lib.FSW_ABI_VERSION and L.fsw_arch() == b"gfx950"
    sizes = tuple(int(v) for v in re.search(r"#define FSW_MID_SIZES \{([^}]*)\}", header).group(1).split(","))
    assert sizes == _lib.MID_SIZES and _lib.NUM_BINS == _lib.REG_MAX_DEG + 1 + len(sizes) + _lib.NUM_LDS_BINS + _lib.NUM_HUB_BINS + 1
    # size helpers are pure host functions
    assert L.fsw_unit_table_rows(32) == 528
    assert L.fsw_graph_workspace_bytes(1000, 10_000) >= 24 * 10_000      # two (key, value) ping-pong buffers
    assert L.fsw_embed_scratch_bytes(100) == 0 and L.fsw_embed_scratch_bytes(5000) > 0
    assert L.fsw_segcumsum_workspace_bytes(10_000) >= 5 * 16


def test_embed_args_struct_matches_header_layout():
    import ctypes
    from fsw_gnn_amd import _lib
    header = open(os.path.join(ROOT, "include", "fsw_hip.h")).read()
    body = header[header.index("typedef struct {"):header.index("} fsw_embed_args;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for stmt in body.split(";"):
        stmt = stmt.replace("typedef struct {", "").strip()
        if not stmt:
            continue
        first, *rest = stmt.split(",")
        names.append(re.findall(r"[A-Za-z_0-9]+", first)[-1])
        names += [re.findall(r"[A-Za-z_0-9]+", r)[-1] for r in rest]
    assert names == [f[0] for f in _lib.EmbedArgs._fields_]
    assert ctypes.sizeof(_lib.EmbedArgs) == 8 * 6 + 8 * 3 + 8 + 8 * 2 + 8 * 3 + 16 + 8 * 5 + 16 + 8 * 3 + 8 + 8


def test_sorting_networks_native():
    exe = "/tmp/fsw_test_sortnet"
    src = os.path.join(ROOT, "tests", "native", "test_sortnet.cpp")
    subprocess.run(["g++", "-O1", "-std=c++17", src, "-o", exe], check=True)
    assert subprocess.run([exe], capture_output=True, text=True).stdout.strip().endswith("OK")


def test_module_surface_and_loud_failures():
    _need_lib()
    from fsw_gnn_amd import FSW_conv, FSW_embedding, segcumsum
    E = FSW_embedding(d_in=5, d_out=9, encode_total_mass=True, freqs_init="spread", device="cpu")
    assert E.nSlices == 8 and E.d_out == 9 and tuple(E.projVecs.shape) == (8, 5) and tuple(E.bias.shape) == (9,)
    assert set(E.state_dict()) == {"projVecs", "freqs", "bias", "total_mass_encoding_scale"}
    k = np.arange(8)
    assert np.allclose(E.freqs.detach().numpy(), ((k + .5) / 8) / (1 - (k + .5) / 8), rtol=1e-6)   # fsw_embedding.py:529-531
    assert np.allclose(E.projVecs.detach().norm(dim=1).numpy(), 1.0, atol=1e-6)
    assert (E.bias == 0).all()
    with pytest.raises(RuntimeError, match="no CPU path"):                  # no silent CPU fallback
        with torch.no_grad():
            E(torch.zeros(4, 5))
    with pytest.raises(RuntimeError, match="no pure-PyTorch path"):
        FSW_embedding(d_in=5, d_out=9, load_custom_cuda_lib=False, device="cpu")
    with pytest.raises(RuntimeError, match="no CPU path"):
        segcumsum(torch.zeros(3), torch.zeros(3, dtype=torch.int64))
    with pytest.raises(NotImplementedError):
        FSW_embedding(d_in=5, nSlices=3, nFreqs=4, device="cpu")
    C = FSW_conv(6, 10, device="cpu")
    assert C.embed_dim == 20 and C.fsw_embed.nSlices == 19 and not C.fsw_embed.enable_bias        # fsw_conv.py:231-237
    assert {"size_coeff", "mlp.0.weight", "mlp.0.bias", "fsw_embed.projVecs", "fsw_embed.freqs",
            "fsw_embed.total_mass_encoding_scale"} == set(C.state_dict())
    assert tuple(C.mlp[0].weight.shape) == (10, 26)
    with pytest.raises(ValueError, match="Invalid argument"):
        FSW_conv(6, 10, device="cpu", config={"no_such_option": 1})
    C2 = FSW_conv(6, 10, device="cpu", config={"mlp_layers": 0, "concat_self": False})
    assert C2.embed_dim == 10 and C2.mlp is None and C2.fsw_embed.enable_bias
    C3 = FSW_conv(6, 10, edgefeat_dim=3, device="cpu")                    # projVecs cover vertex + edge features
    assert tuple(C3.fsw_embed.projVecs.shape) == (19, 9) and C3.fsw_embed.d_edge == 3 and not C3._fusable()


def test_slice_partition():
    from fsw_gnn_amd.dist import slice_partition
    assert slice_partition(256, 8) == [(32 * r, 32 * r + 32) for r in range(8)]
    parts = slice_partition(255, 4)
    assert parts[0] == (0, 64) and parts[-1][1] == 255 and all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
    assert slice_partition(3, 4)[-1] == (3, 3)


_WORKER = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from fsw_gnn_amd.dist import (slice_partition, all_gather_slice_blocks, pipelined_gather, reduce_scatter_pipeline,
                              all_to_all_pipeline, interleave_blocks)
from oracle import fsw_oracle as O
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
g = np.load(os.path.join(sys.argv[1], "tests", "golden", "tiny_graph.npz"))
n = 64
rp, cl, vv = O.csr_from_coo(g["adj_indices"][0], g["adj_indices"][1], g["adj_values"], n)
S = g["V"].shape[0]
parts = slice_partition(S, world)
ka, kb = parts[rank]
wmax = max(b - a for a, b in parts)
# stand-in for the per-rank HIP call: the oracle restricted to this rank's block of slices (plus mass column)
full = O.fsw_embedding_forward(g["X"], rp, cl, vv, g["V"], g["freqs"], encode_total_mass=True)
local = np.zeros((n, 1 + wmax))
local[:, 0] = full[:, 0]
blk = O.fsw_embedding_forward(g["X"], rp, cl, vv, g["V"][ka:kb], g["freqs"][ka:kb])
local[:, 1:1 + (kb - ka)] = blk
localt = torch.from_numpy(local).contiguous()

# 1. gather form, one collective
out = torch.full((n, 1 + S + 3), -7.0, dtype=torch.float64)          # wider buffer, like FSW_conv's concat buffer
all_gather_slice_blocks(localt, parts, 1, out)
err = float(np.abs(out[:, :1 + S].numpy() - full).max())
assert err == 0.0, err                                               # no reduction: bit-identical reassembly
assert (out[:, 1 + S:] == -7.0).all()

# 2. gather form, pipelined over node-range chunks (ragged last chunk)
for cs in (16, 24, 64):
    nch = -(-n // cs)
    out2 = torch.full((n, 1 + S + 3), -7.0, dtype=torch.float64)
    def compute(c, lc):
        r0, r1 = c * cs, min((c + 1) * cs, n)
        lc[:r1 - r0] = localt[r0:r1]
    stats = {}
    pipelined_gather(compute, nch, cs, n, parts, 1, out2, None, stats)
    assert torch.equal(out2, out), cs
    assert stats["collective"] == "all_gather" and stats["bytes_sent_per_rank"] == nch * cs * (1 + wmax) * 8 * (world - 1)

# 3. consumer form: sharded first Linear layer, reduce-scatter of the partial sums, all-gather of the finished rows
rng = np.random.default_rng(5)
H, d = 10, g["X"].shape[1]
W = rng.standard_normal((H, 1 + S + d)); bvec = rng.standard_normal(H)
pre = full @ W[:, :1 + S].T + g["X"] @ W[:, 1 + S:].T + bvec
ref = np.where(pre >= 0, pre, 0.2 * pre)
hm = 1 if rank == 0 else 0
Eblk = torch.from_numpy(np.concatenate([full[:, :hm], full[:, 1 + ka:1 + kb]], axis=1))
W1blk = torch.from_numpy(np.concatenate([W[:, :hm], W[:, 1 + ka:1 + kb]], axis=1))
Xt, W2t, bt = torch.from_numpy(g["X"].astype(np.float64)), torch.from_numpy(W[:, 1 + S:]), torch.from_numpy(bvec)
def finish(r0, r1, R):
    R.addmm_(Xt[r0:r1], W2t.t()); R.add_(bt)
    torch.nn.functional.leaky_relu(R, 0.2, inplace=True)
for cs in (8 * world, 16 * world, 64 * world):
    nch = -(-n // cs)
    def partial(c, Pc):
        r0, r1 = c * cs, min((c + 1) * cs, n)
        Pc[:r1 - r0] = Eblk[r0:r1] @ W1blk.t()
    stats = {}
    Y = reduce_scatter_pipeline(partial, finish, nch, cs, n, H, torch.float64, torch.device("cpu"), None, "replicated", stats)
    assert tuple(Y.shape) == (n, H)
    e = float(np.abs(Y.numpy() - ref).max() / np.abs(ref).max())
    assert e < 1e-13, e                                             # a sum over the ranks: summation order differs, nothing else
    assert stats["collective"] == "reduce_scatter+all_gather"
    R, row0 = reduce_scatter_pipeline(partial, finish, nch, cs, n, H, torch.float64, torch.device("cpu"), None, "sharded")
    m = cs // world
    for c in range(nch):
        r0 = int(row0[c]); r1 = min(r0 + m, n)
        assert r0 == c * cs + rank * m
        if r1 > r0:
            assert float(np.abs(R[c, :r1 - r0].numpy() - ref[r0:r1]).max() / np.abs(ref).max()) < 1e-13

# 4. exchange form: all-to-all of the slice blocks to the owners of the rows, the tail on the owned rows, all-gather of the output
Wt = torch.from_numpy(W)
def finish_x(c, r0, r1, blocks, Yc):
    rows = r1 - r0
    buf = torch.empty((rows, 1 + S + d), dtype=torch.float64)
    interleave_blocks(blocks[:, :rows], parts, 1, buf)
    assert float(np.abs(buf[:, :1 + S].numpy() - full[r0:r1]).max()) == 0.0     # the owner holds the single-GPU embedding rows
    buf[:, 1 + S:] = Xt[r0:r1]
    Yc[:rows] = torch.nn.functional.leaky_relu(buf @ Wt.t() + bt, 0.2)
for cs in (8 * world, 16 * world, 64 * world):
    nch = -(-n // cs)
    def block(c, lc):
        r0, r1 = c * cs, min((c + 1) * cs, n)
        lc[:r1 - r0] = localt[r0:r1]
    stats = {}
    Y = all_to_all_pipeline(block, finish_x, nch, cs, n, 1 + wmax, H, torch.float64, torch.device("cpu"), None, "replicated", stats)
    assert tuple(Y.shape) == (n, H)
    assert float(np.abs(Y.numpy() - ref).max() / np.abs(ref).max()) < 1e-13
    assert stats["collective"] == "all_to_all+all_gather"
    assert stats["bytes_sent_per_rank"] == nch * cs * ((1 + wmax) + H) * 8 * (world - 1) // world
    R, row0 = all_to_all_pipeline(block, finish_x, nch, cs, n, 1 + wmax, H, torch.float64, torch.device("cpu"), None, "sharded", stats)
    assert stats["collective"] == "all_to_all"
    m = cs // world
    for c in range(nch):
        r0 = int(row0[c]); r1 = min(r0 + m, n)
        assert r0 == c * cs + rank * m
        if r1 > r0:
            assert float(np.abs(R[c, :r1 - r0].numpy() - ref[r0:r1]).max() / np.abs(ref).max()) < 1e-13
        assert not R[c, max(r1 - r0, 0):].any()
dist.barrier()
if rank == 0:
    print("SHARD_OK")
'''


@pytest.mark.parametrize("world", [2, 3, 4])
def test_slice_sharding_collective_forms_gloo(world, tmp_path):
    """dist.py's ways of reassembling a slice-sharded layer (all-gather, pipelined all-gather, sharded consumer with
    reduce-scatter, all-to-all exchange to the row owners), the kernels replaced by the oracle restricted to the rank's slices."""
    golden("tiny_graph")
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29600 + world), WORLD_SIZE=str(world))
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = [p.communicate(timeout=180) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-1500:] for o in outs]
    assert "SHARD_OK" in outs[0][0]


def test_chunk_plan():
    from fsw_gnn_amd.dist import chunk_plan
    from fsw_gnn_amd.graph import round_chunk_rows
    assert round_chunk_rows(1) == 2048 and round_chunk_rows(2049) == 4096 and round_chunk_rows(5000, 3) == 6144
    for n, world, want in ((1_000_000, 8, 4), (1_000_000, 2, 4), (3000, 4, 2), (10, 3, 5), (1_000_000, 8, 1)):
        cs, nch = chunk_plan(n, world, want)
        assert cs % 2048 == 0 and cs % world == 0 and nch * cs >= n and (nch - 1) * cs < n and nch <= want


def test_coherence_minimisation_matches_reference():
    """coherence.py against the reference's minimize_mutual_coherence on the same starting points (float64 goldens)."""
    from fsw_gnn_amd.coherence import minimize_mutual_coherence, mutual_coherence
    g = golden("coherence")
    for name in ("a", "b", "c"):
        X0 = torch.from_numpy(g["X0_" + name])
        X = minimize_mutual_coherence(X0.clone())
        assert torch.allclose(X.norm(dim=1), torch.ones(X.shape[0], dtype=X.dtype), atol=1e-12)
        mu0, mu, mu_ref = float(mutual_coherence(X0)), float(mutual_coherence(X)), float(g["mu_" + name])
        assert mu < mu0 and abs(mu - mu_ref) < 1e-6
        assert float((X - torch.from_numpy(g["Xref_" + name])).abs().max()) < 1e-6
    # wired into the constructors like the reference (FSW_conv always minimises, fsw_conv.py:321)
    from fsw_gnn_amd import FSW_conv
    C = FSW_conv(6, 10, device="cpu")
    assert C.fsw_embed.minimize_slice_coherence
    rnd = torch.nn.functional.normalize(torch.randn(19, 6, dtype=torch.float64), dim=1)
    assert float(mutual_coherence(C.fsw_embed.projVecs.detach().double())) < float(mutual_coherence(rnd))


def test_node_block_partition_covers_every_row_once():
    from fsw_gnn_amd.dist import node_block
    for n, world in ((1_000_000, 8), (10, 4), (3, 8), (7, 7), (1, 2)):
        blocks = [node_block(n, world, r) for r in range(world)]
        per = blocks[0][0]
        assert all(b[0] == per for b in blocks) and per * world >= n
        covered = []
        for _, r0, nl in blocks:
            assert 0 <= nl <= per
            covered += list(range(r0, r0 + nl))
        assert covered == list(range(n))


def test_bench_self_launch_command_and_relay(tmp_path, capfd):
    """`python bench.py --gpus N` without a launcher: the parent builds the driver's own torch.distributed.run line
    (one rank per GPU, 127.0.0.1 rendezvous), starts it as a CHILD process without touching the GPU itself, relays rank 0's
    JSON line as the last stdout line and returns the child's exit code."""
    import bench
    argv = ["--gpus", "8", "--steps", "7", "--warmup", "2"]
    cmd = bench.launch_command(argv, 8, 29517, script="/x/bench.py", python="python3")
    assert cmd == ["python3", "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                   "--master-port", "29517", "/x/bench.py"] + argv
    args = bench.parse(argv)
    assert args.gpus == 8 and args.forms == ["gather", "consumer", "exchange"] and args.slices == 256
    assert bench.parse(["--mode", "consumer", "--output", "sharded"]).forms == ["consumer_sharded"]
    assert bench.parse(["--no-fuse"]).forms == ["gather", "exchange"]
    assert bench.parse(["--slices", "1024"]).slices == 1024
    # the relay, with a stand-in for the rank program: two ranks, rank 0 prints the line, everybody prints noise
    script = tmp_path / "fake_rank.py"
    script.write_text(
        "import os, sys\n"
        "r = int(os.environ['RANK']); w = int(os.environ['WORLD_SIZE'])\n"
        "print('noise from rank %d of %d' % (r, w), flush=True)\n"
        "assert os.environ.get('MASTER_ADDR') == '127.0.0.1' and '--gpus' in sys.argv\n"
        "if r == 0:\n"
        "    print('{\"metric\": \"m\", \"n_gpus\": %d}' % w, flush=True)\n"
        "sys.exit(int(os.environ.get('FAKE_RC', '0')) if r == 1 else 0)\n")
    a2 = bench.parse(["--gpus", "2"])
    rc = bench.self_launch(a2, ["--gpus", "2"], script=str(script))
    out = capfd.readouterr()
    assert rc == 0
    assert out.out.strip().splitlines()[-1] == '{"metric": "m", "n_gpus": 2}' and "noise" not in out.out and "noise from rank 1 of 2" in out.err
    os.environ["FAKE_RC"] = "3"
    try:
        assert bench.self_launch(a2, ["--gpus", "2"], script=str(script)) != 0      # a failing rank fails the run
    finally:
        del os.environ["FAKE_RC"]


def test_build_hint_is_per_owner_and_never_risks_an_unknown_graph():
    """graph.BuildHint (replaces round 2's module-global heuristic): the two-level CSR build only once THIS owner has seen a graph
    without hub rows, back to the LSD build after a skewed one, per shape memory, and no coupling between two owners."""
    from fsw_gnn_amd.graph import BuildHint, TWO_LEVEL_MAX_DEGREE
    a, b = BuildHint(), BuildHint()
    assert not a.two_level(1000, 5000)                       # nothing seen yet: the safe build
    a.note(1000, 5000, 30)
    assert a.two_level(1000, 5000) and a.two_level(2000, 9000)
    assert not b.two_level(1000, 5000)                       # another layer has learnt nothing from it
    a.note(4000, 64000, TWO_LEVEL_MAX_DEGREE + 1)            # a skewed graph
    assert not a.two_level(4000, 64000) and not a.two_level(1000, 5000)
    a.note(1000, 5000, 27)                                   # the stream of graphs is tame again ...
    assert a.two_level(1000, 5000) and not a.two_level(4000, 64000)   # ... but the shape that showed hubs stays on the LSD build
    a.note(4000, 64000, 100)
    assert a.two_level(4000, 64000)
    import fsw_gnn_amd.graph as G
    assert not hasattr(G, "_skewed_shapes") and not hasattr(G, "_last_graph_skewed")
