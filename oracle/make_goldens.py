"""Generates tests/golden/*.npz by running the UNMODIFIED reference on CPU (build container only).

    python -m oracle.make_goldens small      # everything except the 1M-node case (about a minute)
    python -m oracle.make_goldens er1m       # BASELINE config 3 in float64 (tens of minutes, ~30 GB RSS)

Inputs come from fsw_gnn_amd/synth.py (counter-based, reproducible anywhere), parameters
(projVecs, freqs, bias, MLP weights) are written into the reference modules explicitly because the
reference draws them from torch's global RNG (fsw_embedding.py:455, 521).  The fixtures hold data
only: small inputs, expected outputs, sampled rows and norms of the large cases.
"""
import json
import os
import sys
import time
import warnings

import numpy as np
import torch

from fsw_gnn_amd import synth
from oracle import ref_harness
from tests import cases
from tests.cases import conv_params, random_freqs

GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
warnings.filterwarnings("ignore")


def T(a, dtype):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype)


def set_params(embmod, V, freqs, bias=None, scale=None):
    with torch.no_grad():
        embmod.projVecs.copy_(T(V, embmod.projVecs.dtype))
        embmod.freqs.copy_(T(freqs, embmod.freqs.dtype))
        if bias is not None:
            embmod.bias.copy_(T(bias, embmod.bias.dtype))
        if scale is not None:
            embmod.total_mass_encoding_scale.fill_(scale)


def save(name, **arrays):
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB", flush=True)


# ------------------------------------------------------------------------------------------------
def case_pointcloud(emb):
    # BASELINE config 1: 1k points x 64 dims x 32 slices, W='unit', non-graph mode
    c = cases.pointcloud_1k()
    X, V, fr = c["X"], c["V"], c["freqs"]
    out = {}
    for dt, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
        E = emb.FSW_embedding(d_in=64, d_out=32, device="cpu", dtype=dt, load_custom_cuda_lib=False)
        set_params(E, V, fr)
        with torch.no_grad():
            out[tag] = E(T(X, dt)).numpy()
    save("pointcloud_1k", out_f64=out["f64"], out_f32=out["f32"])

    # batched weighted point clouds, one batch entry with mass < tau (padding path), one with a zero weight
    Xb = synth.features(3 * 50, 8, seed=21).reshape(3, 50, 8)
    Wb = synth.edge_weights(150, seed=22).reshape(3, 50).copy()
    Wb[1] *= 0.4 / Wb[1].sum()
    Wb[2, 7] = 0.0
    Vb = synth.unit_slices(16, 8, seed=23)
    frb = random_freqs(16, seed=24)
    bias = (0.1 * synth.normal(25, 1, (16,), dtype=np.float64)).astype(np.float32)
    res = {}
    for dt, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
        E = emb.FSW_embedding(d_in=8, d_out=16, device="cpu", dtype=dt, load_custom_cuda_lib=False)
        set_params(E, Vb, frb, bias=bias)
        with torch.no_grad():
            res[tag] = E(T(Xb, dt), T(Wb, dt)).numpy()
            res[tag + "_uniform"] = E(T(Xb, dt), "uniform").numpy()
    save("pointcloud_batch", X=Xb, W=Wb, V=Vb, freqs=frb, bias=bias, **{"out_" + k: v for k, v in res.items()})


def case_tiny_graph(emb, conv):
    n, E_, d, S = 64, 400, 8, 16
    ei = synth.er_multigraph(n, E_, seed=31)
    ei[1] = ei[1] % 56                       # recipients 56..63 stay isolated (zero in-degree)
    ei[:, 390:400] = ei[:, 0:10]             # guaranteed duplicate edges
    X = synth.features(n, d, seed=32)
    V = synth.unit_slices(S, d, seed=33)
    fr = random_freqs(S, seed=34)
    fr[0] = 0.0                              # exercises the xi = 0 (sinc(0) = 1) branch
    bias = (0.1 * synth.normal(35, 1, (S + 1,), dtype=np.float64)).astype(np.float32)
    arrays = dict(edge_index=ei, X=X, V=V, freqs=fr, bias=bias)
    dt = torch.float64
    eit = torch.from_numpy(ei)

    def run(adj, **kw):
        encode = kw.get("encode_total_mass", False)
        Em = emb.FSW_embedding(d_in=d, d_out=S + (1 if encode else 0), device="cpu", dtype=dt,
                               load_custom_cuda_lib=False, **kw)
        set_params(Em, V, fr, bias=(bias if encode else bias[1:]) if kw.get("enable_bias", True) else None,
                   scale=kw.get("total_mass_encoding_scale"))
        with torch.no_grad():
            return Em(T(X, dt), adj, graph_mode=True).numpy()

    adj, _, indeg = conv.FSW_conv.edge_index_to_adj(eit, None, n, 0, dt)
    arrays["in_degrees"] = indeg.numpy().reshape(-1)
    arrays["adj_indices"] = adj.indices().numpy()
    arrays["adj_values"] = adj.values().numpy()
    arrays["out_plain_nomass_nobias"] = run(adj, enable_bias=False)
    for fn in ("identity", "sqrt", "log"):
        for method in ("plain", "homog", "homog_alt"):
            arrays["out_%s_%s" % (fn, method)] = run(
                adj, encode_total_mass=True, total_mass_encoding_function=fn, total_mass_encoding_method=method,
                total_mass_encoding_scale=0.7)
    # larger pad threshold: rows with 1 <= deg < 3 get a non-trivial pad element
    arrays["out_tau3"] = run(adj, enable_bias=False, total_mass_pad_thresh=3.0)
    # serialize_num_slices invariance is a property test; here: gcn weighting + self loops
    adj2, _, indeg2 = conv.FSW_conv.edge_index_to_adj(eit, None, n, 0, dt, self_loop_weight=0.5, edge_weighting="gcn")
    arrays["adj2_indices"] = adj2.indices().numpy()
    arrays["adj2_values"] = adj2.values().numpy()
    arrays["out_gcn_selfloop"] = run(adj2, enable_bias=False)
    # explicit non-unit sparse weights with some rows below tau = 1
    wv = synth.edge_weights(adj.values().numel(), seed=36).astype(np.float64) * 0.25
    adj3 = torch.sparse_coo_tensor(adj.indices(), torch.from_numpy(wv), adj.shape).coalesce()
    arrays["adj3_values"] = wv
    arrays["out_weighted"] = run(adj3, enable_bias=False)
    arrays["out_weighted_mass"] = run(adj3, encode_total_mass=True, total_mass_encoding_scale=0.7)
    save("tiny_graph", **arrays)


def case_conv10k(emb, conv):
    # BASELINE config 2: 10k nodes / 100k edges, 64 feat, 128 slices (+ degree column), 1-layer MLP
    c = cases.conv10k()
    n, d, out_ch, embed_dim = c["n"], c["d"], c["out_ch"], c["embed_dim"]
    ei, X, V, fr, Wl, bl = c["edge_index"], c["X"], c["V"], c["freqs"], c["lin_w"], c["lin_b"]
    rows = np.unique(synth.randint(44, 1, n, 256))
    res, timing = {}, {}
    for dt, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
        C = conv.FSW_conv(d, out_ch, embed_dim=embed_dim, device="cpu", dtype=dt)
        set_params(C.fsw_embed, V, fr)
        with torch.no_grad():
            C.mlp[0].weight.copy_(T(Wl, dt))
            C.mlp[0].bias.copy_(T(bl, dt))
            t0 = time.time()
            y = C(T(X, dt), torch.from_numpy(ei))
            timing[tag] = time.time() - t0
            adj, _, indeg = conv.FSW_conv.edge_index_to_adj(torch.from_numpy(ei), None, n, 0, dt)
            e = C.fsw_embed(T(X, dt), adj, graph_mode=True)
        res["conv_rows_" + tag] = y.numpy()[rows]
        res["emb_rows_" + tag] = e.numpy()[rows]
        res["emb_colnorm_" + tag] = e.norm(dim=0).numpy()
        res["conv_norm_" + tag] = np.array(float(y.norm()))
        res["emb_norm_" + tag] = np.array(float(e.norm()))
        if tag == "f64":
            res["nnz_coalesced"] = np.array(adj.values().numel())
            res["in_degrees"] = indeg.numpy().reshape(-1).astype(np.float32)
    save("conv10k", rows=rows, **res)
    return {"conv10k_forward_seconds": timing, "conv10k_edges_coalesced": int(res["nnz_coalesced"]),
            "conv10k_slices": embed_dim - 1}


def case_rmat(emb, conv, scale=14, edge_factor=16, d=32, S=64):
    c = cases.rmat(scale, edge_factor, d, S)
    n, ei, X, V, fr = c["n"], c["edge_index"], c["X"], c["V"], c["freqs"]
    dt = torch.float64
    adj, _, indeg = conv.FSW_conv.edge_index_to_adj(torch.from_numpy(ei), None, n, 0, dt)
    Em = emb.FSW_embedding(d_in=d, d_out=S + 1, encode_total_mass=True, enable_bias=False, device="cpu", dtype=dt,
                           load_custom_cuda_lib=False)
    set_params(Em, V, fr)
    with torch.no_grad():
        e = Em(T(X, dt), adj, graph_mode=True, serialize_num_slices=16)
    deg = indeg.numpy().reshape(-1)
    hubs = np.argsort(-deg)[:16]
    rows = np.unique(np.concatenate([hubs, synth.randint(54, 1, n, 496)]))
    save("rmat%d" % scale, rows=rows, emb_rows_f64=e.numpy()[rows], emb_colnorm_f64=e.norm(dim=0).numpy(),
         emb_norm_f64=np.array(float(e.norm())), in_degrees=deg.astype(np.float32),
         nnz_coalesced=np.array(adj.values().numel()))


def case_segcumsum(emb):
    arrays = {}
    for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        lens = synth.randint(61, 1, 40, 300) + 1
        lens[5] = 1
        lens[17] = 700                     # longer than one 256-thread block of the reference kernel
        lens[100] = 1300
        ids = np.repeat(np.arange(lens.shape[0], dtype=np.int64) * 3 + 5, lens)
        vals = synth.normal(62, 1, (ids.shape[0],), dtype=np.float64)
        v = T(vals, dt)
        got = emb.segcumsum(v.clone(), torch.from_numpy(ids), always_use_pure_torch=True)
        slow = emb.segcumsum_slow(v, torch.from_numpy(ids))
        arrays.update({"ids": ids, "values_" + tag: v.numpy(), "out_" + tag: got.numpy(), "slow_" + tag: slow.numpy()})
    save("segcumsum", **arrays)


def case_grads(emb, conv):
    """Backward goldens: d(sum(out * R))/d{X, projVecs, freqs, bias, mass scale} in float64 from the
    reference's own autograd Functions (fsw_embedding.py:1232-2257), tiny graph and the 10k-node conv layer."""
    g = np.load(os.path.join(GOLD, "tiny_graph.npz"))
    dt = torch.float64
    n, d, S = 64, 8, 16
    X0, V, fr, bias, ei = g["X"], g["V"], g["freqs"].copy(), g["bias"], g["edge_index"]
    fr[0] = 0.37                                 # keep xi > 0 here: d/dxi at xi = 0 is covered by the oracle only
    R = synth.normal(91, 1, (n, S + 1), dtype=np.float64)
    arrays = {"R": R, "freqs": fr}
    adj, _, _ = conv.FSW_conv.edge_index_to_adj(torch.from_numpy(ei), None, n, 0, dt)
    wv = g["adj3_values"]
    adj3 = torch.sparse_coo_tensor(adj.indices(), torch.from_numpy(wv), adj.shape).coalesce()
    for tag, A in (("unit", adj), ("weighted", adj3)):
        Em = emb.FSW_embedding(d_in=d, d_out=S + 1, encode_total_mass=True, total_mass_encoding_scale=0.7, learnable_slices=True,
                               learnable_freqs=True, learnable_total_mass_encoding_scale=True, device="cpu", dtype=dt,
                               load_custom_cuda_lib=False)
        set_params(Em, V, fr, bias=bias, scale=0.7)
        X = T(X0, dt).requires_grad_(True)
        out = Em(X, A, graph_mode=True)
        (out * torch.from_numpy(R)).sum().backward()
        arrays.update({"out_" + tag: out.detach().numpy(), "gX_" + tag: X.grad.numpy(), "gV_" + tag: Em.projVecs.grad.numpy(),
                       "gfreqs_" + tag: Em.freqs.grad.numpy(), "gbias_" + tag: Em.bias.grad.numpy(),
                       "gscale_" + tag: np.array(float(Em.total_mass_encoding_scale.grad))})
    save("grads_tiny", **arrays)

    c = cases.conv10k()
    n, d, out_ch, embed_dim = c["n"], c["d"], c["out_ch"], c["embed_dim"]
    C = conv.FSW_conv(d, out_ch, embed_dim=embed_dim, device="cpu", dtype=dt)
    set_params(C.fsw_embed, c["V"], c["freqs"])
    with torch.no_grad():
        C.mlp[0].weight.copy_(T(c["lin_w"], dt))
        C.mlp[0].bias.copy_(T(c["lin_b"], dt))
    X = T(c["X"], dt).requires_grad_(True)
    Rc = synth.normal(92, 1, (n, out_ch), dtype=np.float64)
    y = C(X, torch.from_numpy(c["edge_index"]))
    (y * torch.from_numpy(Rc)).sum().backward()
    rows = np.unique(synth.randint(93, 1, n, 256))
    save("grads_conv10k", rows=rows, gX_rows=X.grad.numpy()[rows], gX_norm=np.array(float(X.grad.norm())),
         gV=C.fsw_embed.projVecs.grad.numpy(), gfreqs=C.fsw_embed.freqs.grad.numpy(), gW=C.mlp[0].weight.grad.numpy(),
         gb=C.mlp[0].bias.grad.numpy())


def case_grads_homog(emb, conv):
    """Backward goldens for total_mass_encoding_method 'homog' / 'homog_alt' (reference fsw_embedding.py:874-882, 1136-1144):
    same tiny graph and cotangent as grads_tiny, weighted adjacency (masses on both sides of 1 after the 0.7 scale)."""
    g = np.load(os.path.join(GOLD, "tiny_graph.npz"))
    gt = np.load(os.path.join(GOLD, "grads_tiny.npz"))
    dt = torch.float64
    n, d, S = 64, 8, 16
    X0, V, fr, bias, ei, R = g["X"], g["V"], gt["freqs"], g["bias"], g["edge_index"], gt["R"]
    adj, _, _ = conv.FSW_conv.edge_index_to_adj(torch.from_numpy(ei), None, n, 0, dt)
    adj3 = torch.sparse_coo_tensor(adj.indices(), torch.from_numpy(g["adj3_values"]), adj.shape).coalesce()
    arrays = {}
    for method in ("homog", "homog_alt"):
        for tag, A in (("unit", adj), ("weighted", adj3)):
            Em = emb.FSW_embedding(d_in=d, d_out=S + 1, encode_total_mass=True, total_mass_encoding_scale=0.7,
                                   total_mass_encoding_method=method, learnable_slices=True, learnable_freqs=True,
                                   learnable_total_mass_encoding_scale=True, device="cpu", dtype=dt, load_custom_cuda_lib=False)
            set_params(Em, V, fr, bias=bias, scale=0.7)
            X = T(X0, dt).requires_grad_(True)
            out = Em(X, A, graph_mode=True)
            (out * torch.from_numpy(R)).sum().backward()
            k = method + "_" + tag
            arrays.update({"out_" + k: out.detach().numpy(), "gX_" + k: X.grad.numpy(), "gV_" + k: Em.projVecs.grad.numpy(),
                           "gfreqs_" + k: Em.freqs.grad.numpy(), "gbias_" + k: Em.bias.grad.numpy(),
                           "gscale_" + k: np.array(float(Em.total_mass_encoding_scale.grad))})
    save("grads_homog", **arrays)


def case_grads_w(emb, conv):
    """Gradients with respect to the WEIGHTS from the reference's autograd (ag.div_sparse_dense.backward fsw_embedding.py:1656,
    ag.cumsum_sparse.backward :2160, ag.permute_sparse.backward :1286, custom_lowclamp :1735): tiny graph with sparse weights
    (rows on both sides of the pad threshold), tau = 1 and tau = 3, and a dense batch of weighted point clouds."""
    g = np.load(os.path.join(GOLD, "tiny_graph.npz"))
    gt = np.load(os.path.join(GOLD, "grads_tiny.npz"))
    dt = torch.float64
    n, d, S = 64, 8, 16
    X0, V, fr, bias, ei, R = g["X"], g["V"], gt["freqs"], g["bias"], g["edge_index"], gt["R"]
    adj, _, _ = conv.FSW_conv.edge_index_to_adj(torch.from_numpy(ei), None, n, 0, dt)
    arrays = {}
    for tag, tau, fn in (("tau1", 1.0, "identity"), ("tau3", 3.0, "log")):
        Em = emb.FSW_embedding(d_in=d, d_out=S + 1, encode_total_mass=True, total_mass_encoding_scale=0.7, total_mass_encoding_function=fn,
                               total_mass_pad_thresh=tau, learnable_slices=True, learnable_freqs=True, device="cpu", dtype=dt,
                               load_custom_cuda_lib=False)
        set_params(Em, V, fr, bias=bias, scale=0.7)
        vals = torch.from_numpy(g["adj3_values"].copy()).requires_grad_(True)
        A = torch.sparse_coo_tensor(adj.indices(), vals, adj.shape, is_coalesced=True)
        X = T(X0, dt).requires_grad_(True)
        out = Em(X, A, graph_mode=True)
        (out * torch.from_numpy(R)).sum().backward()
        arrays.update({"out_" + tag: out.detach().numpy(), "gW_" + tag: vals.grad.numpy(), "gX_" + tag: X.grad.numpy(),
                       "gV_" + tag: Em.projVecs.grad.numpy(), "gfreqs_" + tag: Em.freqs.grad.numpy()})
    # dense weights, two point clouds of 40 points (non-graph mode), masses 0.6 and 1.7
    B, npts = 2, 40
    Xc = synth.normal(95, 1, (B, npts, d), dtype=np.float64)
    Wc = synth.uniform01(96, 1, 0, B * npts).reshape(B, npts)
    Wc = Wc / Wc.sum(axis=1, keepdims=True) * np.array([[0.6], [1.7]])
    Rc = synth.normal(97, 1, (B, S), dtype=np.float64)
    Em = emb.FSW_embedding(d_in=d, d_out=S, device="cpu", dtype=dt, load_custom_cuda_lib=False, enable_bias=False)
    set_params(Em, V, fr)
    Wt = torch.from_numpy(Wc).requires_grad_(True)
    out = Em(torch.from_numpy(Xc), Wt)
    (out * torch.from_numpy(Rc)).sum().backward()
    arrays.update({"cloud_X": Xc, "cloud_W": Wc, "cloud_R": Rc, "cloud_out": out.detach().numpy(), "cloud_gW": Wt.grad.numpy()})
    save("grads_w", **arrays)


def case_testconv64(emb, conv):
    """The configuration of the reference's own test_conv.py (:9-57) on deterministic inputs: float64, 100 vertices, 50 vertex /
    11 edge features, 35 outputs, three MLP layers, homogeneous degree encoding with the 'log' function and a learnable scale,
    self_loop_weight 0.2, final BatchNorm, eval mode; out, out(16 x) and the gradients of out.norm()."""
    dt = torch.float64
    n, dv, de, dout = 100, 50, 11, 35
    ei = synth.er_multigraph(n, 1000, seed=101)
    ei = ei[:, ei[0] < ei[1]]                                            # networkx G.edges: every undirected edge once
    X0 = synth.normal(102, 1, (n, dv), dtype=np.float64)
    E0 = synth.normal(103, 1, (ei.shape[1], de), dtype=np.float64)
    torch.manual_seed(104)
    C = conv.FSW_conv(dv, dout, edgefeat_dim=de, mlp_layers=3, bias=False, vertex_degree_encoding_function='log',
                      vertex_degree_encoding_scale=1, learnable_vertex_degree_encoding_scale=True, homog_degree_encoding=True,
                      learnable_embedding=True, concat_self=True, batchNorm_final=True, device="cpu", dtype=dt, self_loop_weight=0.2)
    C.eval()
    X = torch.from_numpy(X0).requires_grad_(True)
    Ef = torch.from_numpy(E0).requires_grad_(True)
    eit = torch.from_numpy(ei)
    out = C(X, edge_index=eit, edge_features=Ef)
    with torch.no_grad():
        out2 = C(16 * X, edge_index=eit, edge_features=16 * Ef)
    out.norm().backward()
    arrays = {"edge_index": ei, "X": X0, "Ef": E0, "out": out.detach().numpy(), "out16": out2.numpy(),
              "gX": X.grad.numpy(), "gEf": Ef.grad.numpy()}
    for k, v in C.state_dict().items():
        arrays["param." + k] = v.detach().numpy()
    for k, p in C.named_parameters():
        if p.grad is not None:
            arrays["grad." + k] = p.grad.numpy()
    save("testconv64", **arrays)


def case_edgefeat(emb, conv):
    """Edge features (reference fsw_embedding.py:934-968, fsw_conv.py:419-439): FSW_conv with edgefeat_dim = 3 on
    the tiny multigraph (duplicate edges: the reference sums their features and weights in coalesce()), forward and
    autograd gradients in float64; plus a self-loop variant (loops carry zero edge features, fsw_conv.py:428-433)."""
    g = np.load(os.path.join(GOLD, "tiny_graph.npz"))
    dt = torch.float64
    n, d, de, out_ch, embed_dim = 64, 8, 3, 5, 17
    ei, X0 = g["edge_index"], g["X"]
    EF = synth.normal(95, 1, (ei.shape[1], de), dtype=np.float64)
    V = synth.unit_slices(embed_dim - 1, d + de, seed=96)
    fr = cases.random_freqs(embed_dim - 1, seed=97)
    Wl = (synth.normal(98, 1, (out_ch, embed_dim + d), dtype=np.float64) / 5).astype(np.float32)
    bl = (0.1 * synth.normal(99, 1, (out_ch,), dtype=np.float64)).astype(np.float32)
    R = synth.normal(94, 1, (n, out_ch), dtype=np.float64)
    arrays = dict(edge_features=EF, V=V, freqs=fr, lin_w=Wl, lin_b=bl, R=R)
    for tag, kw in (("plain", {}), ("selfloop", {"self_loop_weight": 0.5})):
        C = conv.FSW_conv(d, out_ch, edgefeat_dim=de, embed_dim=embed_dim, device="cpu", dtype=dt, **kw)
        set_params(C.fsw_embed, V, fr)
        with torch.no_grad():
            C.mlp[0].weight.copy_(T(Wl, dt))
            C.mlp[0].bias.copy_(T(bl, dt))
        X = T(X0, dt).requires_grad_(True)
        Ef = torch.from_numpy(EF).requires_grad_(True)
        y = C(X, torch.from_numpy(ei), edge_features=Ef)
        (y * torch.from_numpy(R)).sum().backward()
        adj, X_edge, _ = conv.FSW_conv.edge_index_to_adj(torch.from_numpy(ei), Ef.detach(), n, de, dt, **kw)
        with torch.no_grad():
            e = C.fsw_embed(X.detach(), adj, X_edge, graph_mode=True)
        arrays.update({"y_" + tag: y.detach().numpy(), "emb_" + tag: e.numpy(), "gX_" + tag: X.grad.numpy(),
                       "gEF_" + tag: Ef.grad.numpy(), "gV_" + tag: C.fsw_embed.projVecs.grad.numpy(),
                       "gfreqs_" + tag: C.fsw_embed.freqs.grad.numpy(), "gW_" + tag: C.mlp[0].weight.grad.numpy()})
    save("edgefeat_tiny", **arrays)


def case_coherence(emb):
    """minimize_mutual_coherence (fsw_embedding.py:3045-3248) on three fixed starting points, float64."""
    out = {}
    for name, (n, d, seed) in {"a": (24, 6, 71), "b": (64, 16, 72), "c": (128, 64, 73)}.items():
        X0 = torch.from_numpy(synth.normal(seed, 1, (n, d), dtype=np.float64))
        Xr = emb.minimize_mutual_coherence(X0.clone(), report=False)
        G = Xr @ Xr.t()
        G.fill_diagonal_(0)
        out["X0_" + name], out["Xref_" + name], out["mu_" + name] = X0.numpy(), Xr.numpy(), np.array(float(G.abs().max()))
    save("coherence", **out)


def case_er1m(emb, conv, nslices=256, serialize=4):
    # BASELINE config 3: ER-style multigraph 1M nodes / 10M edges, 128 feat, 256 slices (+ degree column)
    t0 = time.time()
    c = cases.er1m(nslices)
    n, d, ei, X, V, fr = c["n"], c["d"], c["edge_index"], c["X"], c["V"], c["freqs"]
    print("inputs generated in %.1fs" % (time.time() - t0), flush=True)
    dt = torch.float64
    adj, _, indeg = conv.FSW_conv.edge_index_to_adj(torch.from_numpy(ei), None, n, 0, dt)
    deg = indeg.numpy().reshape(-1)
    Em = emb.FSW_embedding(d_in=d, d_out=nslices + 1, encode_total_mass=True, enable_bias=False, freqs_init="spread",
                           device="cpu", dtype=dt, load_custom_cuda_lib=False)
    set_params(Em, V, fr)
    t0 = time.time()
    with torch.no_grad():
        e = Em(T(X, dt), adj, graph_mode=True, serialize_num_slices=serialize)
    secs = time.time() - t0
    print("reference fp64 forward: %.1fs" % secs, flush=True)
    zero = np.nonzero(deg == 0)[0]
    top = np.argsort(-deg)[:32]
    rows = np.unique(np.concatenate([zero, top, synth.randint(74, 1, n, 1000)]))
    save("er1m", rows=rows, emb_rows_f64=e.numpy()[rows], emb_colnorm_f64=e.norm(dim=0).numpy(),
         emb_norm_f64=np.array(float(e.norm())), nnz_coalesced=np.array(adj.values().numel()),
         num_zero_degree=np.array(zero.shape[0]), max_degree=np.array(deg.max()),
         degree_hist=np.bincount(deg.astype(np.int64)))
    return {"er1m_fp64_forward_seconds": secs, "er1m_serialize_num_slices": serialize,
            "er1m_edges_coalesced": int(adj.values().numel()), "er1m_slices": nslices}


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "small"
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(os.cpu_count())
    emb, conv = ref_harness.load()
    timings_path = os.path.join(GOLD, "ref_timings.json")
    timings = json.load(open(timings_path)) if os.path.exists(timings_path) else {}
    timings["cpu_threads"] = os.cpu_count()
    timings["torch"] = torch.__version__
    if what == "small":
        case_pointcloud(emb)
        case_tiny_graph(emb, conv)
        case_segcumsum(emb)
        timings.update(case_conv10k(emb, conv))
        case_rmat(emb, conv)
    elif what == "grads":
        case_grads(emb, conv)
    elif what == "grads_homog":
        case_grads_homog(emb, conv)
    elif what == "coherence":
        case_coherence(emb)
    elif what == "edgefeat":
        case_edgefeat(emb, conv)
    elif what == "grads_w":
        case_grads_w(emb, conv)
    elif what == "testconv64":
        case_testconv64(emb, conv)
    elif what == "er1m":
        timings.update(case_er1m(emb, conv))
    else:
        raise SystemExit("usage: python -m oracle.make_goldens [small|grads|grads_homog|coherence|edgefeat|grads_w|testconv64|er1m]")
    json.dump(timings, open(timings_path, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
