"""GPU tests of the float64 build of the path and of the gradients with respect to the weights (csrc/embed_generic.hip,
fsw_project_f64), against goldens computed by the reference itself in float64 (oracle/make_goldens.py).

Tolerance: 1e-12 norm-wise relative for float64 forward results and 1e-10 for gradients (sums of ~1e3 terms accumulated
in a different order than the reference's sparse ops); float32 weight gradients 2e-5 like the other float32 gradients.
"""
import numpy as np
import pytest
import torch

from tests import cases
from tests.conftest import golden, relerr

pytestmark = pytest.mark.gpu
F64, G64 = 1e-12, 1e-10


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def t(a, dev, dtype=torch.float64):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device=dev, dtype=dtype)


def make_embedding(dev, V, freqs, bias=None, scale=None, dtype=torch.float64, **kw):
    from fsw_gnn_amd import FSW_embedding
    S, d = V.shape
    encode = kw.get("encode_total_mass", False)
    E = FSW_embedding(d_in=d, d_out=S + (1 if encode else 0), device=dev, dtype=dtype, **kw)
    with torch.no_grad():
        E.projVecs.copy_(t(V, dev, dtype))
        E.freqs.copy_(t(freqs, dev, dtype))
        if bias is not None and E.enable_bias:
            E.bias.copy_(t(bias, dev, dtype))
        if scale is not None:
            E.total_mass_encoding_scale.fill_(scale)
    return E


def sparse_adj(idx, vals, shape, dev, dtype=torch.float64):
    return torch.sparse_coo_tensor(torch.from_numpy(idx).to(dev), t(vals, dev, dtype), shape).coalesce()


def test_float64_projection_on_the_matrix_cores(dev):
    from fsw_gnn_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(1)
    for n, d, S in ((1000, 64, 32), (777, 13, 70), (50, 3, 129), (4096, 128, 256)):
        X, V = rng.standard_normal((n, d)), rng.standard_normal((S, d))
        Xd, Vd = t(X, dev), t(V, dev)
        Xp = torch.full((n, S + 5), float("nan"), device=dev, dtype=torch.float64)
        stats = torch.zeros(8, dtype=torch.int32, device=dev)
        assert L.fsw_project_f64(Xd.data_ptr(), n, d, d, Vd.data_ptr(), S, d, Xp.data_ptr(), S + 5, stats.data_ptr(),
                                 torch.cuda.current_stream().cuda_stream) == 0
        assert relerr(Xp[:, :S].cpu().numpy(), X @ V.T) < 1e-14 and int(stats[0]) == 0


def test_float64_tiny_graph_every_variant(dev):
    """Every forward variant of the tiny-graph golden (weights, pad element, mass encodings, 'gcn' + self loops, dense W) in
    float64 against the reference's float64 output."""
    g = golden("tiny_graph")
    X = t(g["X"], dev)
    adj = sparse_adj(g["adj_indices"], g["adj_values"], (64, 64), dev)
    with torch.no_grad():
        E = make_embedding(dev, g["V"], g["freqs"], enable_bias=False)
        out = E(X, adj, graph_mode=True).cpu().numpy()
        assert out.dtype == np.float64 and relerr(out, g["out_plain_nomass_nobias"]) < F64
        assert np.abs(out[56:]).max() == 0.0                                  # isolated recipients
        for fn in ("identity", "sqrt", "log"):
            for method in ("plain", "homog", "homog_alt"):
                E = make_embedding(dev, g["V"], g["freqs"], bias=g["bias"], scale=0.7, encode_total_mass=True,
                                   total_mass_encoding_function=fn, total_mass_encoding_method=method, total_mass_encoding_scale=0.7)
                assert relerr(E(X, adj, graph_mode=True).cpu().numpy(), g["out_%s_%s" % (fn, method)]) < F64, (fn, method)
        E = make_embedding(dev, g["V"], g["freqs"], enable_bias=False, total_mass_pad_thresh=3.0)
        assert relerr(E(X, adj, graph_mode=True).cpu().numpy(), g["out_tau3"]) < F64
        E = make_embedding(dev, g["V"], g["freqs"], enable_bias=False)
        adj2 = sparse_adj(g["adj2_indices"], g["adj2_values"], (64, 64), dev)
        assert relerr(E(X, adj2, graph_mode=True).cpu().numpy(), g["out_gcn_selfloop"]) < F64
        adj3 = sparse_adj(g["adj_indices"], g["adj3_values"], (64, 64), dev)
        assert relerr(E(X, adj3, graph_mode=True).cpu().numpy(), g["out_weighted"]) < F64
        assert relerr(E(X, adj3.to_dense(), graph_mode=True).cpu().numpy(), g["out_weighted"]) < F64
    # the point cloud of BASELINE config 1 (1000 points: the global-scratch sort of the generic kernel)
    gp, c = golden("pointcloud_1k"), cases.pointcloud_1k()
    with torch.no_grad():
        E = make_embedding(dev, c["V"], c["freqs"])
        assert relerr(E(t(c["X"], dev)).cpu().numpy(), gp["out_f64"]) < F64


@pytest.mark.parametrize("kind", ["plain", "homog", "homog_alt"])
def test_float64_backward_vs_reference_autograd(dev, kind):
    g, gt = golden("tiny_graph"), golden("grads_tiny")
    gg = gt if kind == "plain" else golden("grads_homog")
    for tag, vals in (("unit", g["adj_values"]), ("weighted", g["adj3_values"])):
        key = tag if kind == "plain" else kind + "_" + tag
        E = make_embedding(dev, g["V"], gt["freqs"], bias=g["bias"], scale=0.7, encode_total_mass=True, total_mass_encoding_scale=0.7,
                           total_mass_encoding_method=kind, learnable_slices=True, learnable_freqs=True,
                           learnable_total_mass_encoding_scale=True)
        X = t(g["X"], dev).requires_grad_(True)
        out = E(X, sparse_adj(g["adj_indices"], vals, (64, 64), dev), graph_mode=True)
        assert relerr(out.detach().cpu().numpy(), gg["out_" + key]) < F64
        (out * t(gt["R"], dev)).sum().backward()
        assert relerr(X.grad.cpu().numpy(), gg["gX_" + key]) < G64
        assert relerr(E.projVecs.grad.cpu().numpy(), gg["gV_" + key]) < G64
        assert relerr(E.freqs.grad.cpu().numpy(), gg["gfreqs_" + key]) < G64
        assert relerr(E.bias.grad.cpu().numpy(), gg["gbias_" + key]) < G64
        assert abs(float(E.total_mass_encoding_scale.grad) - float(gg["gscale_" + key])) < G64 * abs(float(gg["gscale_" + key]))


@pytest.mark.parametrize("dtype,tol", [(torch.float64, G64), (torch.float32, 2e-5)])
def test_weight_gradients_vs_reference_autograd(dev, dtype, tol):
    """d loss / d W (sparse graph weights with rows on both sides of the pad threshold, tau = 1 and 3, mass column through
    identity / log; dense point-cloud weights) against the reference's autograd; float64 and float32 modules."""
    g, gt, gw = golden("tiny_graph"), golden("grads_tiny"), golden("grads_w")
    idx = torch.from_numpy(g["adj_indices"]).to(dev)
    for tag, tau, fn in (("tau1", 1.0, "identity"), ("tau3", 3.0, "log")):
        E = make_embedding(dev, g["V"], gt["freqs"], bias=g["bias"], scale=0.7, dtype=dtype, encode_total_mass=True,
                           total_mass_encoding_scale=0.7, total_mass_encoding_function=fn, total_mass_pad_thresh=tau,
                           learnable_slices=True, learnable_freqs=True)
        vals = t(g["adj3_values"], dev, dtype).requires_grad_(True)
        A = torch.sparse_coo_tensor(idx, vals, (64, 64), is_coalesced=True)
        X = t(g["X"], dev, dtype).requires_grad_(True)
        out = E(X, A, graph_mode=True)
        assert relerr(out.detach().cpu().numpy(), gw["out_" + tag]) < (F64 if dtype == torch.float64 else 1e-5)
        (out * t(gt["R"], dev, dtype)).sum().backward()
        assert vals.grad is not None and relerr(vals.grad.cpu().numpy(), gw["gW_" + tag]) < tol, tag
        assert relerr(X.grad.cpu().numpy(), gw["gX_" + tag]) < tol
        assert relerr(E.projVecs.grad.cpu().numpy(), gw["gV_" + tag]) < tol
        assert relerr(E.freqs.grad.cpu().numpy(), gw["gfreqs_" + tag]) < tol
    E = make_embedding(dev, g["V"], gt["freqs"], dtype=dtype, enable_bias=False)
    W = t(gw["cloud_W"], dev, dtype).requires_grad_(True)
    out = E(t(gw["cloud_X"], dev, dtype), W)
    assert relerr(out.detach().cpu().numpy(), gw["cloud_out"]) < (F64 if dtype == torch.float64 else 1e-5)
    (out * t(gw["cloud_R"], dev, dtype)).sum().backward()
    assert relerr(W.grad.cpu().numpy(), gw["cloud_gW"]) < tol


def test_float64_conv_replays_the_reference_test_script(dev):
    """The reference's own test_conv.py (:9-57): float64 FSW_conv with edge features, three MLP layers, homogeneous 'log'
    degree encoding with a learnable scale, self_loop_weight 0.2, final BatchNorm, eval mode.  Same state_dict, same inputs:
    out, the homogeneity deviation the script prints, and every gradient of out.norm() against the reference's."""
    from fsw_gnn_amd import FSW_conv
    g = golden("testconv64")
    torch.manual_seed(0)
    C = FSW_conv(50, 35, edgefeat_dim=11, mlp_layers=3, bias=False, vertex_degree_encoding_function='log',
                 vertex_degree_encoding_scale=1, learnable_vertex_degree_encoding_scale=True, homog_degree_encoding=True,
                 learnable_embedding=True, concat_self=True, batchNorm_final=True, device=dev, dtype=torch.float64, self_loop_weight=0.2)
    sd = {k[len("param."):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("param.")}
    assert set(sd) == set(C.state_dict())                       # the reference's state_dict loads as is
    C.load_state_dict(sd)
    C.eval()
    X = t(g["X"], dev).requires_grad_(True)
    Ef = t(g["Ef"], dev).requires_grad_(True)
    ei = torch.from_numpy(g["edge_index"]).to(dev)
    out = C(X, edge_index=ei, edge_features=Ef)
    with torch.no_grad():
        out2 = C(16 * X, edge_index=ei, edge_features=16 * Ef)
    assert relerr(out.detach().cpu().numpy(), g["out"]) < 1e-11
    assert float((torch.norm(out2 - 16 * out) / torch.norm(out)).detach()) < 1e-10     # "Relative deviation from homogeneity"
    out.norm().backward()
    assert relerr(X.grad.cpu().numpy(), g["gX"]) < G64
    assert relerr(Ef.grad.cpu().numpy(), g["gEf"]) < G64
    for k, p in C.named_parameters():
        if "grad." + k in g.files:
            assert p.grad is not None and relerr(p.grad.cpu().numpy(), g["grad." + k]) < G64, k
    # ten SGD steps like the script: runs, the learnable degree scale moves
    s0 = float(C.fsw_embed.total_mass_encoding_scale)
    opt = torch.optim.SGD(C.parameters(), lr=0.01)
    for _ in range(3):
        opt.step()
        opt.zero_grad()
        C(X, edge_index=ei, edge_features=Ef).norm().backward()
    assert float(C.fsw_embed.total_mass_encoding_scale) != s0 and torch.isfinite(C.fsw_embed.total_mass_encoding_scale.grad)


def test_float64_readout_layer(dev):
    from fsw_gnn_amd import FSW_readout
    from oracle import fsw_oracle as O
    rng = np.random.default_rng(4)
    sizes = [40, 1, 300, 0, 77]
    n, d = sum(sizes), 8
    gi = np.repeat(np.arange(5), sizes).astype(np.int64)
    X = rng.standard_normal((n, d))
    ro = FSW_readout(d, 20, concat_self=False, mlp_layers=0, bias=False, device=dev, dtype=torch.float64)
    with torch.no_grad():
        out = ro(t(X, dev), torch.from_numpy(gi).to(dev), 5).cpu().numpy()
    V, fr = ro.fsw_embed.projVecs.detach().cpu().numpy(), ro.fsw_embed.freqs.detach().cpu().numpy()
    ref = O.fsw_embedding_forward(X, np.concatenate([[0], np.cumsum(sizes)]), np.arange(n), np.ones(n), V, fr, encode_total_mass=True)
    assert relerr(out, ref) < F64 and np.abs(out[3]).max() == 0.0
