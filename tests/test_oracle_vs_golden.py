"""Pins the CPU oracle (oracle/fsw_oracle.py, oracle/fsw_oracle.c) to golden vectors captured from the
reference's own CPU path (oracle/make_goldens.py).  Runs without a GPU.

Tolerances: the oracle computes in float64 like the float64 goldens, with a different (CSR, per-row)
evaluation order, so agreement is expected near 1e-13; 1e-10 norm-wise is asserted.
"""
import numpy as np
import pytest

from oracle import fsw_oracle as O
from tests import cases
from tests.conftest import golden, relerr

TOL64 = 1e-10


def test_pointcloud_1k_config1():
    g = golden("pointcloud_1k")
    c = cases.pointcloud_1k()
    out = O.point_cloud_forward(c["X"][None], None, c["V"], c["freqs"])[0]
    assert relerr(out, g["out_f64"]) < TOL64
    # the reference's own float32 output sits within its float32 accuracy floor of the float64 one
    assert relerr(g["out_f32"], g["out_f64"]) < 1e-5


def test_pointcloud_batch_weighted_padding_zero_weight():
    g = golden("pointcloud_batch")
    out = O.point_cloud_forward(g["X"], g["W"], g["V"], g["freqs"]) + g["bias"].astype(np.float64)
    assert relerr(out, g["out_f64"]) < TOL64
    n = g["X"].shape[1]
    outu = O.point_cloud_forward(g["X"], np.full(g["W"].shape, 1.0 / n), g["V"], g["freqs"]) + g["bias"].astype(np.float64)
    assert relerr(outu, g["out_f64_uniform"]) < TOL64


def _tiny_csr(g, which="adj"):
    idx = g["adj_indices"] if which != "adj2" else g["adj2_indices"]
    vals = {"adj": g["adj_values"], "adj2": g["adj2_values"], "adj3": g["adj3_values"]}[which]
    return O.csr_from_coo(idx[0], idx[1], vals, 64)


def test_tiny_graph_adjacency_matches_reference_coalesce():
    g = golden("tiny_graph")
    rowptr, col, w, indeg = O.coalesce_edge_index(g["edge_index"], 64)
    rp, cl, vv = _tiny_csr(g)
    assert np.array_equal(rowptr, rp) and np.array_equal(col, cl) and np.array_equal(w, vv)
    assert np.array_equal(indeg, g["in_degrees"])
    assert (np.diff(rowptr)[56:] == 0).all() and w.max() >= 2.0      # isolated rows and duplicate edges exist
    rowptr2, col2, w2, _ = O.coalesce_edge_index(g["edge_index"], 64, self_loop_weight=0.5, edge_weighting="gcn")
    rp2, cl2, vv2 = _tiny_csr(g, "adj2")
    assert np.array_equal(rowptr2, rp2) and np.array_equal(col2, cl2)
    assert relerr(w2, vv2) < 1e-14


@pytest.mark.parametrize("fn", ["identity", "sqrt", "log"])
@pytest.mark.parametrize("method", ["plain", "homog", "homog_alt"])
def test_tiny_graph_total_mass_encodings(fn, method):
    g = golden("tiny_graph")
    rp, cl, vv = _tiny_csr(g)
    out = O.fsw_embedding_forward(g["X"], rp, cl, vv, g["V"], g["freqs"], bias=g["bias"], encode_total_mass=True,
                                  total_mass_encoding_function=fn, total_mass_encoding_method=method,
                                  total_mass_encoding_scale=0.7)
    assert relerr(out, g["out_%s_%s" % (fn, method)]) < TOL64


def test_tiny_graph_variants():
    g = golden("tiny_graph")
    rp, cl, vv = _tiny_csr(g)
    base = O.fsw_embedding_forward(g["X"], rp, cl, vv, g["V"], g["freqs"])
    assert relerr(base, g["out_plain_nomass_nobias"]) < TOL64
    assert np.abs(base[56:]).max() == 0.0                            # zero in-degree -> zero embedding row
    tau3 = O.fsw_embedding_forward(g["X"], rp, cl, vv, g["V"], g["freqs"], total_mass_pad_thresh=3.0)
    assert relerr(tau3, g["out_tau3"]) < TOL64
    rp2, cl2, vv2 = _tiny_csr(g, "adj2")
    assert relerr(O.fsw_embedding_forward(g["X"], rp2, cl2, vv2, g["V"], g["freqs"]), g["out_gcn_selfloop"]) < TOL64
    rp3, cl3, vv3 = _tiny_csr(g, "adj3")
    assert relerr(O.fsw_embedding_forward(g["X"], rp3, cl3, vv3, g["V"], g["freqs"]), g["out_weighted"]) < TOL64
    wm = O.fsw_embedding_forward(g["X"], rp3, cl3, vv3, g["V"], g["freqs"], bias=g["bias"], encode_total_mass=True,
                                 total_mass_encoding_scale=0.7)
    assert relerr(wm, g["out_weighted_mass"]) < TOL64


def test_duplicates_equal_separate_unit_elements():
    """Property used by the HIP CSR build: k parallel edges == one edge of weight k."""
    g = golden("tiny_graph")
    ei = g["edge_index"]
    order = np.argsort(ei[1], kind="stable")
    rowptr = np.concatenate([[0], np.cumsum(np.bincount(ei[1], minlength=64))])
    out = O.fsw_embedding_forward(g["X"], rowptr, ei[0][order], np.ones(ei.shape[1]), g["V"], g["freqs"])
    assert relerr(out, g["out_plain_nomass_nobias"]) < TOL64


def test_conv10k_config2():
    g = golden("conv10k")
    c = cases.conv10k()
    rowptr, col, w, indeg = O.coalesce_edge_index(c["edge_index"], c["n"])
    assert w.shape[0] == int(g["nnz_coalesced"])
    emb = O.fsw_embedding_forward(c["X"], rowptr, col, w, c["V"], c["freqs"], encode_total_mass=True)
    rows = g["rows"]
    assert relerr(emb[rows], g["emb_rows_f64"]) < TOL64
    assert relerr(np.linalg.norm(emb, axis=0), g["emb_colnorm_f64"]) < TOL64
    y = O.conv_tail(emb, c["X"].astype(np.float64), linear_weight=c["lin_w"], linear_bias=c["lin_b"])
    assert relerr(y[rows], g["conv_rows_f64"]) < TOL64
    assert abs(np.linalg.norm(y) - float(g["conv_norm_f64"])) / float(g["conv_norm_f64"]) < TOL64
    # accuracy floor of the reference's float32 path against its own float64 path (documented in DESIGN.md)
    assert relerr(g["emb_rows_f32"], g["emb_rows_f64"]) < 1e-5


def test_rmat14_skewed_degrees():
    g = golden("rmat14")
    c = cases.rmat(14)
    rowptr, col, w, indeg = O.coalesce_edge_index(c["edge_index"], c["n"])
    assert w.shape[0] == int(g["nnz_coalesced"])
    assert np.diff(rowptr).max() > 1000                               # hubs present
    rows = g["rows"]
    # evaluate only the sampled rows (plus nothing else): build a sub-CSR
    deg = np.diff(rowptr)[rows]
    sub_rp = np.concatenate([[0], np.cumsum(deg)])
    idx = np.concatenate([np.arange(rowptr[r], rowptr[r + 1]) for r in rows])
    emb = O.fsw_embedding_forward(c["X"], sub_rp, col[idx], w[idx], c["V"], c["freqs"], encode_total_mass=True)
    assert relerr(emb, g["emb_rows_f64"]) < TOL64


def test_segcumsum_vectors():
    g = golden("segcumsum")
    for tag, tol in (("f64", 1e-13), ("f32", 1e-5)):
        got = O.segcumsum(g["values_" + tag], g["ids"])
        assert relerr(got, g["slow_" + tag]) == 0.0                   # same left-to-right order as segcumsum_slow
        assert relerr(got, g["out_" + tag]) < tol                     # reference log-step scan, different rounding


@pytest.mark.parametrize("tag,which", [("unit", "adj"), ("weighted", "adj3")])
def test_backward_oracle_matches_reference_autograd(tag, which):
    """The analytic gradients of the oracle against the reference's autograd (float64 goldens, tiny graph)."""
    g = golden("tiny_graph")
    gg = golden("grads_tiny")
    rp, cl, vv = _tiny_csr(g, which)
    R = gg["R"]
    gX, gV, gxi = O.fsw_embed_csr_backward(g["X"], rp, cl, vv, g["V"], gg["freqs"], R[:, 1:])
    assert relerr(gX, gg["gX_" + tag]) < 1e-10
    assert relerr(gV, gg["gV_" + tag]) < 1e-10
    assert relerr(gxi, gg["gfreqs_" + tag]) < 1e-10
    assert relerr(R.sum(axis=0), gg["gbias_" + tag]) < 1e-12
    mass = np.zeros(64)
    np.add.at(mass, np.repeat(np.arange(64), np.diff(rp)), vv)
    assert abs((R[:, 0] * mass).sum() - float(gg["gscale_" + tag])) < 1e-10 * abs(float(gg["gscale_" + tag]))


def test_backward_oracle_conv10k():
    gg = golden("grads_conv10k")
    c = cases.conv10k()
    rowptr, col, w, _ = O.coalesce_edge_index(c["edge_index"], c["n"])
    X = c["X"].astype(np.float64)
    emb = O.fsw_embedding_forward(X, rowptr, col, w, c["V"], c["freqs"], encode_total_mass=True)
    Rc = cases.synth.normal(92, 1, (c["n"], c["out_ch"]), dtype=np.float64)
    h = np.concatenate([emb, X], axis=1)
    Wl, bl = c["lin_w"].astype(np.float64), c["lin_b"].astype(np.float64)
    pre = h @ Wl.T + bl
    gpre = Rc * np.where(pre >= 0, 1.0, 0.2)                                   # LeakyReLU(0.2) backward
    gh = gpre @ Wl
    E = c["embed_dim"]
    gX_e, gV, gxi = O.fsw_embed_csr_backward(X, rowptr, col, w, c["V"], c["freqs"], gh[:, 1:E])
    gX = gX_e + gh[:, E:]
    assert relerr(gX[gg["rows"]], gg["gX_rows"]) < 1e-9
    assert relerr(gV, gg["gV"]) < 1e-9 and relerr(gxi, gg["gfreqs"]) < 1e-9
    assert relerr(gpre.T @ h, gg["gW"]) < 1e-9


@pytest.mark.parametrize("tag,slw", [("plain", 0.0), ("selfloop", 0.5)])
def test_edge_features_oracle_forward_and_backward(tag, slw):
    """Edge features (reference fsw_embedding.py:934-968, fsw_conv.py:419-439) incl. duplicate edges and self loops."""
    g, ge = golden("tiny_graph"), golden("edgefeat_tiny")
    n, d, de = 64, 8, 3
    rowptr, col, w, _, ef, slot = O.coalesce_edge_index(g["edge_index"], n, self_loop_weight=slw, edge_features=ge["edge_features"])
    X = g["X"].astype(np.float64)
    emb = O.fsw_embedding_forward(X, rowptr, col, w, ge["V"], ge["freqs"], encode_total_mass=True, edge_feat=ef)
    assert relerr(emb, ge["emb_" + tag]) < TOL64
    Wl, bl = ge["lin_w"].astype(np.float64), ge["lin_b"].astype(np.float64)
    pre = np.concatenate([emb, X], axis=1) @ Wl.T + bl
    assert relerr(np.where(pre >= 0, pre, 0.2 * pre), ge["y_" + tag]) < TOL64
    gpre = ge["R"] * np.where(pre >= 0, 1.0, 0.2)
    gh = gpre @ Wl
    E = emb.shape[1]
    gX, gV, gxi, gef = O.fsw_embed_csr_backward(X, rowptr, col, w, ge["V"], ge["freqs"], gh[:, 1:E], edge_feat=ef)
    assert relerr(gX + gh[:, E:], ge["gX_" + tag]) < 1e-10
    assert relerr(gV, ge["gV_" + tag]) < 1e-10 and relerr(gxi, ge["gfreqs_" + tag]) < 1e-10
    assert relerr(gef[slot], ge["gEF_" + tag]) < 1e-10                       # every duplicate receives its slot's gradient
