"""CPU oracle for the FSW_conv / FSW_embedding forward hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a plain-numpy restatement of the reference's algorithm.  It is imported only by
tests/, by __graft_entry__.smoke() and by bench.py's cpu_baseline leg, and only as the checker.  The
product path (fsw_gnn_amd/) never imports it and has no CPU fallback.

Parity pin: every function here is checked in tests/test_oracle_vs_golden.py against golden vectors
captured from the reference's own CPU path (/root/reference/fsw_embedding.py, fsw_conv.py run
unmodified in the build container by oracle/make_goldens.py; fixtures under tests/golden/).

Each function cites the reference lines (paths relative to /root/reference/) it restates.
"""
import numpy as np


# --------------------------------------------------------------------------------------------------
# adjacency construction                                                     fsw_conv.py:384-447
# --------------------------------------------------------------------------------------------------
def coalesce_edge_index(edge_index, num_vertices, self_loop_weight=0.0, edge_weighting="unit",
                        edge_values=None, dtype=np.float64, edge_features=None):
    """edge_index [2,E] (row 0 = sender, row 1 = recipient) -> CSR of adj[recipient, sender].

    fsw_conv.py:387      inds = edge_index.flip(0)  (row = recipient, col = sender)
    fsw_conv.py:388      unit values
    fsw_conv.py:390-395  optional self loops of weight self_loop_weight
    fsw_conv.py:397-398  coalesce(): duplicates summed, entries sorted by (recipient, sender)
    fsw_conv.py:400-409  in-degrees; 'gcn' weighting divides by sqrt(deg_i) * sqrt(deg_j)
    fsw_conv.py:419-439  edge features: same indices, coalesce() sums the feature vectors of duplicate edges; self
                         loops carry zero features.  With edge_features the return value gains ef[nnz, d_edge] and
                         slot[E] (CSR position of every input edge).
    Returns rowptr[int64 n+1], col[int64 nnz], w[dtype nnz], in_degrees[dtype n].
    """
    edge_index = np.asarray(edge_index)
    src = edge_index[0].astype(np.int64)
    dst = edge_index[1].astype(np.int64)
    vals = np.ones(src.shape[0], dtype=dtype) if edge_values is None else np.asarray(edge_values, dtype=dtype)
    ef_in = None if edge_features is None else np.asarray(edge_features, dtype=dtype).reshape(src.shape[0], -1)
    num_input_edges = src.shape[0]
    if self_loop_weight > 0:
        loops = np.arange(num_vertices, dtype=np.int64)
        src = np.concatenate([src, loops])
        dst = np.concatenate([dst, loops])
        vals = np.concatenate([vals, np.full(num_vertices, self_loop_weight, dtype=dtype)])
        if ef_in is not None:
            ef_in = np.concatenate([ef_in, np.zeros((num_vertices, ef_in.shape[1]), dtype=dtype)])
    key = dst * np.int64(num_vertices) + src
    order = np.argsort(key, kind="stable")
    key = key[order]
    vals = vals[order]
    if key.size:
        first = np.concatenate([[True], key[1:] != key[:-1]])
    else:
        first = np.zeros(0, dtype=bool)
    seg = np.cumsum(first) - 1
    ukey = key[first]
    w = np.zeros(ukey.shape[0], dtype=dtype)
    np.add.at(w, seg, vals)
    row = ukey // num_vertices
    col = ukey % num_vertices
    rowptr = np.zeros(num_vertices + 1, dtype=np.int64)
    np.add.at(rowptr, row + 1, 1)
    rowptr = np.cumsum(rowptr)
    in_deg = np.zeros(num_vertices, dtype=dtype)
    np.add.at(in_deg, row, w)
    if edge_weighting == "gcn":
        ds = np.sqrt(in_deg)
        w = w / ds[row] / ds[col]
    elif edge_weighting != "unit":
        raise RuntimeError("Invalid weighting method passed in argument <edge_weighting>")
    if ef_in is not None:
        ef = np.zeros((ukey.shape[0], ef_in.shape[1]), dtype=dtype)
        np.add.at(ef, seg, ef_in[order])
        slot = np.empty(order.shape[0], dtype=np.int64)
        slot[order] = seg
        return rowptr, col, w, in_deg, ef, slot[:num_input_edges]
    return rowptr, col, w, in_deg


def csr_from_coo(rows, cols, vals, num_rows):
    """Coalesced COO (sorted by (row, col)) -> CSR.  Used for FSW_embedding.forward(W=sparse)."""
    rows = np.asarray(rows, dtype=np.int64)
    rowptr = np.zeros(num_rows + 1, dtype=np.int64)
    np.add.at(rowptr, rows + 1, 1)
    return np.cumsum(rowptr), np.asarray(cols, dtype=np.int64), np.asarray(vals)


# --------------------------------------------------------------------------------------------------
# embedding core                                                       fsw_embedding.py:778-1112
# --------------------------------------------------------------------------------------------------
def fsw_embed_csr(X, rowptr, col, w, projVecs, freqs, total_mass_pad_thresh=1.0, dtype=np.float64,
                  chunk_elems=1 << 23, return_mass=False, edge_feat=None):
    """out[r, k] = (1+xi_k) * sum_t Delta_t * p_(t)  for every CSR row r and slice k.

    fsw_embedding.py:778-784    mass m_r = sum of the row's weights
    fsw_embedding.py:787        pad deficit max(tau - m_r, 0)
    fsw_embedding.py:790-821    if ANY row is deficient every row gets one extra element x = 0 carrying
                                its deficit (zero for non-deficient rows)
    fsw_embedding.py:817,823-829 normalise by max(m_r, tau)
    fsw_embedding.py:909-913    projection Xp = X . projVecs^T
    fsw_embedding.py:917-932,1016-1025  per-slice ascending sort of the neighbourhood, weights follow
    fsw_embedding.py:1031-1032  inclusive cumulative weights c_t (the segmented cumsum)
    fsw_embedding.py:1047-1075  Delta_t = 2 w_t sinc(xi w_t) cos(pi xi (2 c_t - w_t)),
                                sinc(z) = sin(pi z)/(pi z)
    fsw_embedding.py:1084-1109  out = (1+xi) * sum_t Delta_t p_(t)
    fsw_embedding.py:934-968    edge features (edge_feat [nnz, d_edge], projVecs [S, d_in + d_edge]): the key of
                                neighbour j of recipient i is <x_j, v[:d_in]> + <e_ij, v[d_in:]>; the pad element stays 0
    """
    X = np.asarray(X, dtype=dtype)
    V = np.asarray(projVecs, dtype=dtype)
    Ve = None
    if edge_feat is not None:
        edge_feat = np.asarray(edge_feat, dtype=dtype)
        Ve = V[:, X.shape[1]:]
        V = V[:, :X.shape[1]]
    xi = np.asarray(freqs, dtype=dtype)
    rowptr = np.asarray(rowptr, dtype=np.int64)
    col = np.asarray(col, dtype=np.int64)
    w = np.asarray(w, dtype=dtype)
    tau = np.dtype(dtype).type(total_mass_pad_thresh)
    nrows = rowptr.shape[0] - 1
    S = V.shape[0]
    deg = np.diff(rowptr)

    mass = np.zeros(nrows, dtype=dtype)
    np.add.at(mass, np.repeat(np.arange(nrows), deg), w)
    deficit = np.maximum(tau - mass, 0)
    any_deficit = bool((deficit > 0).any())
    denom = np.maximum(mass, tau) if any_deficit else mass

    Xp = X @ V.T                                                   # [n, S]
    out = np.zeros((nrows, S), dtype=dtype)

    for D in np.unique(deg):
        rows = np.nonzero(deg == D)[0]
        Dp = int(D) + (1 if any_deficit else 0)
        if Dp == 0:
            continue
        step = max(1, chunk_elems // max(1, Dp * S))
        for a in range(0, rows.shape[0], step):
            rr = rows[a:a + step]
            R = rr.shape[0]
            idx = rowptr[rr][:, None] + np.arange(D)[None, :]      # [R, D]
            keys = np.zeros((R, Dp, S), dtype=dtype)
            wts = np.zeros((R, Dp), dtype=dtype)
            if D > 0:
                keys[:, :D, :] = Xp[col[idx]]
                if Ve is not None:
                    keys[:, :D, :] += edge_feat[idx] @ Ve.T
                wts[:, :D] = w[idx]
            if any_deficit:
                wts[:, D] = deficit[rr]                            # pad element at x = 0
            wts = wts / denom[rr][:, None]
            order = np.argsort(keys, axis=1, kind="stable")
            ks = np.take_along_axis(keys, order, axis=1)
            ws = np.take_along_axis(np.broadcast_to(wts[:, :, None], keys.shape), order, axis=1)
            c = np.cumsum(ws, axis=1)
            delta = 2 * ws * np.sinc(xi[None, None, :] * ws) * np.cos(np.pi * xi[None, None, :] * (2 * c - ws))
            out[rr] = (1 + xi)[None, :] * np.sum(delta * ks, axis=1)
    if return_mass:
        return out, mass
    return out


def total_mass_encode(out, mass, function="identity", method="plain", scale=1.0):
    """Prepend the total-mass column.                               fsw_embedding.py:853-884

    'identity' f(m)=m; 'sqrt' f(m)=2m/(sqrt(m+1)+1); 'log' f(m)=log1p(m)        (:857-865)
    'plain'  -> [f*scale, emb]; 'homog' -> [f*scale*mean|emb|, emb];
    'homog_alt' -> [part1(f*scale)*mean|emb|, part2(f*scale)*emb]                (:874-882, 1136-1144)
    """
    m = mass[:, None]
    if function == "identity":
        tm = m.copy()
    elif function == "sqrt":
        tm = 2 * (m / (np.sqrt(m + 1) + 1))
    elif function == "log":
        tm = np.log1p(m)
    else:
        raise RuntimeError("bad total_mass_encoding_function")
    tm = tm * scale
    if method == "plain":
        return np.concatenate([tm, out], axis=-1)
    norm = np.mean(np.abs(out), axis=-1, keepdims=True)
    if method == "homog":
        return np.concatenate([tm * norm, out], axis=-1)
    if method == "homog_alt":
        p1 = np.where(tm <= 1, tm * (2 - tm), 1)
        p2 = np.where(tm <= 1, tm * tm, 2 * tm - 1)
        return np.concatenate([p1 * norm, p2 * out], axis=-1)
    raise RuntimeError("bad total_mass_encoding_method")


def fsw_embedding_forward(X, rowptr, col, w, projVecs, freqs, bias=None, encode_total_mass=False,
                          total_mass_encoding_function="identity", total_mass_encoding_method="plain",
                          total_mass_encoding_scale=1.0, total_mass_pad_thresh=1.0, dtype=np.float64, edge_feat=None):
    """FSW_embedding.forward in graph mode on a CSR adjacency.    fsw_embedding.py:587-890"""
    emb, mass = fsw_embed_csr(X, rowptr, col, w, projVecs, freqs, total_mass_pad_thresh, dtype, return_mass=True,
                              edge_feat=edge_feat)
    if encode_total_mass:
        emb = total_mass_encode(emb, mass, total_mass_encoding_function, total_mass_encoding_method,
                                np.dtype(dtype).type(total_mass_encoding_scale))
    if bias is not None:
        emb = emb + np.asarray(bias, dtype=dtype)                  # fsw_embedding.py:886-888
    return emb


def point_cloud_forward(X, W, projVecs, freqs, total_mass_pad_thresh=1.0, dtype=np.float64):
    """Non-graph mode: X [b, n, d], W [b, n] (or None = unit) -> [b, S].  fsw_embedding.py:704-728, 983-1004

    The dense branch of the reference computes diff(2 c sinc(2 xi c)); by the sum-to-product identity
    this equals the sparse branch's 2 w sinc(xi w) cos(pi xi (2c - w)) used here.
    """
    X = np.asarray(X, dtype=dtype)
    b, n, d = X.shape
    Wd = np.ones((b, n), dtype=dtype) if W is None else np.asarray(W, dtype=dtype)
    rowptr = np.arange(b + 1, dtype=np.int64) * n
    col = np.arange(b * n, dtype=np.int64)
    return fsw_embed_csr(X.reshape(b * n, d), rowptr, col, Wd.reshape(-1), projVecs, freqs,
                         total_mass_pad_thresh, dtype)


# --------------------------------------------------------------------------------------------------
# FSW_conv tail                                                             fsw_conv.py:357-369
# --------------------------------------------------------------------------------------------------
def conv_tail(emb, vertex_features, message_weight_vs_self=1.0, concat_self=True, linear_weight=None,
              linear_bias=None, negative_slope=0.2):
    """cat(mw * emb, x) -> one Linear + LeakyReLU layer (the default mlp_layers=1 MLP)."""
    h = np.concatenate([message_weight_vs_self * emb, vertex_features], axis=-1) if concat_self else emb
    if linear_weight is None:
        return h
    y = h @ np.asarray(linear_weight, dtype=h.dtype).T
    if linear_bias is not None:
        y = y + np.asarray(linear_bias, dtype=h.dtype)
    return np.where(y >= 0, y, negative_slope * y)


# --------------------------------------------------------------------------------------------------
# segmented cumulative sum                                     fsw_embedding.py:3016-3027
# --------------------------------------------------------------------------------------------------
def segcumsum(values, segment_ids):
    """Inclusive scan of values restarted wherever consecutive segment_ids differ (segcumsum_slow)."""
    values = np.asarray(values)
    ids = np.asarray(segment_ids)
    n = values.shape[0]
    if n == 0:
        return values.copy()
    start = np.concatenate([[True], ids[1:] != ids[:-1]])
    out = np.empty_like(values)
    # sequential definition, vectorised per segment so rounding matches a left-to-right sum
    bounds = np.nonzero(start)[0].tolist() + [n]
    for a, b in zip(bounds[:-1], bounds[1:]):
        out[a:b] = np.cumsum(values[a:b])
    return out


# --------------------------------------------------------------------------------------------------
# backward of the embedding core                      fsw_embedding.py:1232-2257 (the ag.*.backward chain)
# --------------------------------------------------------------------------------------------------
def _dsinc(z):
    """d/dz sinc(z), sinc(z) = sin(pi z)/(pi z)  (reference sp.dsinc, fsw_embedding.py:2760-2774)."""
    z = np.asarray(z, dtype=np.float64)
    safe = np.where(z == 0, 1.0, z)
    return np.where(z == 0, 0.0, (np.cos(np.pi * z) - np.sinc(z)) / safe)


def fsw_embed_csr_backward(X, rowptr, col, w, projVecs, freqs, G, total_mass_pad_thresh=1.0, chunk_elems=1 << 22,
                           return_gXp=False, Xp_override=None, edge_feat=None, keys_override=None):
    """Gradients of sum(out * G) for out = fsw_embed_csr(...) with respect to X, projVecs and freqs (float64).

    The reference obtains them by reverse-mode autograd through its sparse ops: sum_sparseToDense.backward,
    mul/sinc_cos backward (fsw_embedding.py:1796-1817), cumsum_sparse.backward = reverse segcumsum (:2158-2172),
    permute_sparse.backward (:1284-1325), sort.backward = scatter (:2055-2070), tensordot.  With the sort order
    held fixed (it is piecewise constant) this is
        d out[r,k] / d p_(t)  = (1 + xi_k) Delta_t
        d out[r,k] / d xi_k   = sum_t Delta_t p_(t) + (1 + xi_k) sum_t (d Delta_t / d xi) p_(t)
        d Delta / d xi        = 2 w [ w sinc'(xi w) cos(B) - sinc(xi w) pi (2c - w) sin(B) ],  B = pi xi (2c - w)
    followed by gX = gXp . V, gV = gXp^T . X.  The weights are treated as constants.
    Xp_override: use these projections (e.g. the float32 ones of the path under test) to decide the sort order;
    two neighbours whose projections agree to float32 rounding may otherwise swap ranks between float32 and
    float64, which moves g (C[s] - C[s+1]) between two entries of gXp (the gradient is discontinuous there).
    """
    X = np.asarray(X, dtype=np.float64)
    V = np.asarray(projVecs, dtype=np.float64)
    Ve = gkey_all = None
    if edge_feat is not None:      # edge features: returns (gX, gV [S, d_in + d_edge], gxi, g_edge_feat [nnz, d_edge])
        edge_feat = np.asarray(edge_feat, dtype=np.float64)
        Ve, V = V[:, X.shape[1]:], V[:, :X.shape[1]]
        gkey_all = np.zeros((edge_feat.shape[0], V.shape[0]))
    xi = np.asarray(freqs, dtype=np.float64)
    G = np.asarray(G, dtype=np.float64)
    rowptr = np.asarray(rowptr, dtype=np.int64)
    col = np.asarray(col, dtype=np.int64)
    w = np.asarray(w, dtype=np.float64)
    tau = float(total_mass_pad_thresh)
    nrows = rowptr.shape[0] - 1
    S = V.shape[0]
    deg = np.diff(rowptr)
    mass = np.zeros(nrows)
    np.add.at(mass, np.repeat(np.arange(nrows), deg), w)
    deficit = np.maximum(tau - mass, 0)
    any_deficit = bool((deficit > 0).any())
    denom = np.maximum(mass, tau) if any_deficit else mass
    Xp = X @ V.T if Xp_override is None else np.asarray(Xp_override, dtype=np.float64)
    gXp = np.zeros_like(Xp)
    gxi = np.zeros(S)
    for D in np.unique(deg):
        rows = np.nonzero(deg == D)[0]
        Dp = int(D) + (1 if any_deficit else 0)
        if D == 0:
            continue
        step = max(1, chunk_elems // max(1, Dp * S))
        for a in range(0, rows.shape[0], step):
            rr = rows[a:a + step]
            R = rr.shape[0]
            idx = rowptr[rr][:, None] + np.arange(D)[None, :]
            keys = np.zeros((R, Dp, S))
            wts = np.zeros((R, Dp))
            keys[:, :D, :] = Xp[col[idx]]
            if Ve is not None:
                keys[:, :D, :] += edge_feat[idx] @ Ve.T
            if keys_override is not None:                                       # float32 keys of the path under test
                keys[:, :D, :] = np.asarray(keys_override, dtype=np.float64)[idx]
            wts[:, :D] = w[idx]
            if any_deficit:
                wts[:, D] = deficit[rr]
            wts = wts / denom[rr][:, None]
            order = np.argsort(keys, axis=1, kind="stable")
            ks = np.take_along_axis(keys, order, axis=1)
            ws = np.take_along_axis(np.broadcast_to(wts[:, :, None], keys.shape), order, axis=1)
            c = np.cumsum(ws, axis=1)
            x3 = xi[None, None, :]
            B = np.pi * x3 * (2 * c - ws)
            sc = np.sinc(x3 * ws)
            delta = 2 * ws * sc * np.cos(B)
            ddelta = 2 * ws * (ws * _dsinc(x3 * ws) * np.cos(B) - sc * np.pi * (2 * c - ws) * np.sin(B))
            g = G[rr][:, None, :]                                              # [R, 1, S]
            gxi += np.sum(g * (delta * ks + (1 + x3) * ddelta * ks), axis=(0, 1))
            gk_sorted = g * (1 + x3) * delta                                   # gradient w.r.t. the sorted keys
            gk = np.zeros_like(gk_sorted)
            np.put_along_axis(gk, order, gk_sorted, axis=1)                    # back to neighbour order
            np.add.at(gXp, col[idx], gk[:, :D, :])                             # the pad element has no source row
            if gkey_all is not None:
                gkey_all[idx] = gk[:, :D, :]
    if gkey_all is not None:
        return gXp @ V, np.concatenate([gXp.T @ X, gkey_all.T @ edge_feat], axis=1), gxi, gkey_all @ Ve
    if return_gXp:
        return gXp @ V, gXp.T @ X, gxi, gXp
    return gXp @ V, gXp.T @ X, gxi


def mass_value(mass, function="identity"):
    if function == "identity":
        return mass
    if function == "sqrt":
        return 2 * (mass / (np.sqrt(mass + 1) + 1))
    return np.log1p(mass)
