"""FSW_embedding: host-side mirror of the reference's FSW_embedding module (reference fsw_embedding.py:169-1144)
over the hand-written HIP kernels of libfsw_hip.so.

Same constructor arguments, parameter names (projVecs, freqs, bias, total_mass_encoding_scale -- state_dict
compatible, reference fsw_embedding.py:397-409), forward signature and assertion messages as the reference.
The computation is NOT the reference's chain of torch.sparse_coo operations: forward() builds a CSR adjacency
(graph.py), projects the points with fp32 MFMA (fsw_project_f32) and runs the fused per-neighbourhood
sort + cumulative-sum + Fourier readout kernels (fsw_embed_f32).  There is no CPU or pure-PyTorch fallback:
tensors must live on a HIP device and the native library must be present.

Autograd: forward under grad mode goes through _EmbedGraphFn (HIP backward kernels, csrc/embed_bwd.hip and
csrc/embed_wsort_bwd.hip) for every weight mode and degree class; gradients flow to X, projVecs, freqs, bias and the
total-mass scale (the weights W are constants).
Edge features (d_edge > 0) go through the coalescing CSR build and the general-weight kernels.
dtype=torch.float64 modules and gradients w.r.t. the weights W / sparse edge features run on the generic kernels
(csrc/embed_generic.hip: any degree, float64 arithmetic).  Not implemented (raise NotImplementedError): Cartesian mode.
"""
import ctypes
import numbers
import struct

import numpy as np
import os

import torch
import torch.nn as nn
from torch.autograd.function import once_differentiable

from . import _lib
from .coherence import minimize_mutual_coherence
from .graph import CSRGraph, build_csr, build_csr_coalesced

version = "0.1-mi355x"

# Same role as the reference's module flag (fsw_embedding.py:109): validate user input on every forward.
# The checks are fused into the HIP kernels (flags word read back once), not separate isnan/isinf scans.
fsw_embedding_basic_safety_checks = True

_MASS_FN = {"identity": 0, "sqrt": 1, "log": 2}


def _round_up(v, m):
    return (v + m - 1) // m * m


def edge_feat_rows(ctx):
    return ctx.num_edge_rows


def _grad_x(L, gXp, S, ldp, V, stream):
    """gX = gXp[:, :S] . V  ([n, S] x [S, d_in]).  The projection kernel (bf16x3 on the matrix cores, fp32-accurate) takes it as
    a projection of the rows of gXp onto the d_in "slices" V^T when the shapes suit it (S <= 256, 16-byte rows); hipBLASLt's
    fp32 GEMM otherwise (2.8 ms against 0.6 ms at 1M rows, S = 256, d_in = 128)."""
    n, d_in = gXp.shape[0], V.shape[1]
    if S % 4 == 0 and S <= 256 and ldp % 4 == 0 and d_in >= 1 and n >= 1:
        Vt = V.t().contiguous()                                   # [d_in, S]
        ldo = _round_up(d_in, 32)
        out = torch.empty((n, ldo), dtype=torch.float32, device=gXp.device)
        _lib.check(L.fsw_project_f32(_lib.ptr(gXp), n, S, ldp, _lib.ptr(Vt), d_in, S, _lib.ptr(out), ldo, None, 0, None, stream),
                   "fsw_project_f32")
        return out if ldo == d_in else out[:, :d_in].contiguous()
    return gXp[:, :S] @ V


GEMM_TN_MIN_ROWS = 1 << 16   # below this the BLAS GEMM is as good


def gemm_tn(A, B):
    """A^T . B for tall row-major A [K, M], B [K, N] (float32, unit inner stride): csrc/gemm_tn.hip when the result fits its 16 blocks
    of 64 x 64 and K is large (the weight-gradient shape: 1M rows x 256 x 128), the BLAS GEMM otherwise."""
    K, M = A.shape
    N = B.shape[1]
    blocks = -(-M // 64) * -(-N // 64)
    if (A.is_cuda and K >= GEMM_TN_MIN_ROWS and blocks <= 16 and not os.environ.get("FSW_GEMM_TN_OFF") and A.dtype == torch.float32 and B.dtype == torch.float32
            and A.stride(1) == 1 and B.stride(1) == 1 and B.shape[0] == K):
        L = _lib.lib()
        C = torch.empty((M, N), dtype=torch.float32, device=A.device)
        wsb = int(L.fsw_gemm_tn_workspace_bytes(M, N))
        ws = torch.empty(wsb, dtype=torch.uint8, device=A.device)
        _lib.check(L.fsw_gemm_tn_f32(_lib.ptr(A), A.stride(0), _lib.ptr(B), B.stride(0), K, M, N, _lib.ptr(C), N, 0.0, _lib.ptr(ws), wsb,
                                     torch.cuda.current_stream(A.device).cuda_stream), "fsw_gemm_tn_f32")
        return C
    return A.t() @ B


class LinearTallFn(torch.autograd.Function):
    """y = h . W^T + b like torch.nn.functional.linear, with the weight gradient gW = gy^T . h -- a reduction over the ROWS, a million
    of them in a training step at BASELINE config 3 -- on gemm_tn instead of the BLAS library's split-K (reference fsw_conv.py:361:
    the first Linear layer of the MLP applied to cat((emb, vertex_features)))."""

    @staticmethod
    def forward(ctx, h, W, b):
        ctx.save_for_backward(h, W)
        ctx.has_bias = b is not None
        return torch.addmm(b, h, W.t()) if b is not None else h @ W.t()

    @staticmethod
    def backward(ctx, gy):
        h, W = ctx.saved_tensors
        gy = gy.contiguous()
        gh = gy @ W if ctx.needs_input_grad[0] else None
        gW = gemm_tn(gy, h) if ctx.needs_input_grad[1] else None
        gb = gy.sum(dim=0) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        return gh, gW, gb


class LinearSplitTallFn(torch.autograd.Function):
    """y = [scale * emb | x] . W^T + b without building the concatenation (reference fsw_conv.py:357-361: torch.cat((mw * emb,
    vertex_features)) followed by mlp[0]): two GEMMs on the operands where they lie, weight gradient blocks on gemm_tn.  Saves the
    [n, embed_dim + in_channels] buffer, its copy kernel and the slicing copies of its backward (0.8 + 0.5 ms at config 3)."""

    @staticmethod
    def forward(ctx, emb, x, W, b, scale):
        E = emb.shape[1]
        ctx.save_for_backward(emb, x, W)
        ctx.scale, ctx.has_bias = float(scale), b is not None
        y = torch.addmm(b, x, W[:, E:].t()) if b is not None else x @ W[:, E:].t()
        return y.addmm_(emb, W[:, :E].t(), alpha=float(scale))

    @staticmethod
    def backward(ctx, gy):
        emb, x, W = ctx.saved_tensors
        E = emb.shape[1]
        gy = gy.contiguous()
        gemb = (gy @ W[:, :E]).mul_(ctx.scale) if ctx.needs_input_grad[0] else None
        gx = gy @ W[:, E:] if ctx.needs_input_grad[1] else None
        gW = None
        if ctx.needs_input_grad[2]:
            gW = torch.empty_like(W)
            gW[:, :E] = gemm_tn(gy, emb).mul_(ctx.scale)
            gW[:, E:] = gemm_tn(gy, x)
        gb = gy.sum(dim=0) if (ctx.has_bias and ctx.needs_input_grad[3]) else None
        return gemb, gx, gW, gb, None


class _EmbedGraphFn(torch.autograd.Function):
    """out = out_scale * E(X, graph) with gradients for X, projVecs, freqs, bias and the total-mass scale.

    Forward runs the same HIP kernels as inference (prepare + embed_into); backward runs csrc/embed_bwd.hip
    (neighbourhood ranks recomputed in registers, one wave-wide float atomic per neighbour into gXp) and two plain
    GEMMs.  Replaces the reference's chain of sparse autograd Functions (fsw_embedding.py:1232-2257).
    Every weight mode and degree class is supported (unit-weight register rows use the float64 coefficient tables, weighted
    rows and rows above 32 neighbours evaluate the coefficients in float64 on the fly); the weights themselves are
    constants (no gradient w.r.t. W); the 'homog' mass encodings are differentiated by embed_autograd on top of this.

    slice_range = (ka, kb): only slices ka..kb-1 (multi-GPU slice sharding, dist.py): the output is
    [mass column | slices ka..kb-1] and the parameter gradients are full-size tensors that are zero outside the block.
    With reduce_grads the block gradients of all ranks of `group` are summed by all_reduce inside backward, so that every rank ends with
    the gradients one GPU would compute (the mass column, replicated on every rank, is counted once).
    """

    @staticmethod
    def forward(ctx, X, projVecs, freqs, bias, mass_scale, edge_feat, module, graph, out_scale, slice_range, group, reduce_grads):
        ka, kb = (0, module.nSlices) if slice_range is None else slice_range
        has_mass = 1 if module.encode_total_mass else 0
        with torch.no_grad():
            # 'homog' / 'homog_alt': embed_autograd asks for the 'plain' embedding (module._force_plain) and applies the
            # epilogue with differentiable torch ops on top of it
            assert (not module.encode_total_mass) or module.total_mass_encoding_method == 'plain' or module._force_plain
            prepared = module.prepare(X, graph, slice_range=slice_range)
            out = torch.empty((graph.num_rows, has_mass + kb - ka), dtype=X.dtype, device=X.device)
            module.embed_into(X, graph, out, out_scale=out_scale, prepared=prepared)
        ctx.module, ctx.graph, ctx.prepared, ctx.out_scale = module, graph, prepared, float(out_scale)
        ctx.slice_range, ctx.group, ctx.reduce_grads = (ka, kb), group, bool(reduce_grads)
        ctx.num_edge_rows = 0 if edge_feat is None else edge_feat.shape[0]
        ctx.save_for_backward(X, projVecs, freqs)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        module, graph, prepared, out_scale = ctx.module, ctx.graph, ctx.prepared, ctx.out_scale
        X, Vfull, freqs_full = ctx.saved_tensors
        ka, kb = ctx.slice_range
        group = ctx.group
        sharded = ctx.reduce_grads        # group may be None = the default process group
        L = _lib.lib()
        S, has_mass = kb - ka, (1 if module.encode_total_mass else 0)
        V = Vfull.detach()[ka:kb]
        fr = freqs_full.detach()[ka:kb].contiguous()
        g = g.contiguous()
        stream = torch.cuda.current_stream(X.device).cuda_stream
        ldp, Xp, table, st = prepared["ldp"], prepared["Xp"], prepared["table"], prepared["stats"]
        gX = gV = gfreqs = gbias = gscale = None
        need_xp = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        gEf = None
        gVb = gfb = None          # block gradients (rows ka..kb-1 of projVecs, entries ka..kb-1 of freqs)
        if S > 0 and graph.ef is not None and (need_xp or ctx.needs_input_grad[2] or ctx.needs_input_grad[5]):
            # edge features: the kernels store the gradient of every key; everything else is index_add + three GEMMs
            nnz = st[_lib.STAT_NNZ]
            scratch = None
            if st[_lib.STAT_NUM_GLOBAL] > 0:
                scratch = torch.empty(int(L.fsw_embed_scratch_bytes(st[_lib.STAT_MAX_DEGREE])), dtype=torch.uint8, device=X.device)
            gkey = torch.zeros((max(nnz, 1), S), dtype=torch.float32, device=X.device)
            gf = torch.zeros(S, dtype=torch.float32, device=X.device)
            a = module.make_args(graph, st, Xp, ldp, fr, S, table, None, 0, None, out_scale, has_mass, scratch, slice_offset=ka)
            _lib.check(L.fsw_embed_backward_keys_f32(ctypes.byref(a), None, _lib.ptr(g), g.stride(0), _lib.ptr(gkey), S, _lib.ptr(gf), stream),
                       "fsw_embed_backward_keys_f32")
            gkey, ef = gkey[:nnz], graph.ef[:nnz]
            if need_xp:
                gXp = torch.zeros((X.shape[0], S), dtype=torch.float32, device=X.device).index_add_(0, graph.col[:nnz].long(), gkey)
                if ctx.needs_input_grad[0]:
                    gX = gXp @ V[:, :module.d_in]
                if ctx.needs_input_grad[1]:
                    gVb = torch.cat([gXp.t() @ X.detach(), gkey.t() @ ef], dim=1)
            if ctx.needs_input_grad[2]:
                gfb = gf
            if ctx.needs_input_grad[5]:
                slot = graph.slot_of_edge[:edge_feat_rows(ctx)].long()
                gslots = gkey @ V[:, module.d_in:]
                gEf = torch.where((slot >= 0)[:, None], gslots[slot.clamp(min=0)], torch.zeros((), device=X.device))
        elif S > 0 and (need_xp or ctx.needs_input_grad[2]):
            gXp = torch.zeros((X.shape[0], ldp), dtype=torch.float32, device=X.device)
            dtable = None
            if prepared["unit_fast"]:
                dtable = torch.empty_like(table)
                _lib.check(L.fsw_unit_dcoeff_table(_lib.ptr(fr), S, _lib.REG_MAX_DEG, _lib.ptr(dtable), ldp, stream), "fsw_unit_dcoeff_table")
            scratch = None
            if st[_lib.STAT_NUM_GLOBAL] > 0:
                scratch = torch.empty(int(L.fsw_embed_scratch_bytes(st[_lib.STAT_MAX_DEGREE])), dtype=torch.uint8, device=X.device)
            gf = torch.zeros(S, dtype=torch.float32, device=X.device)
            a = module.make_args(graph, st, Xp, ldp, fr, S, table, None, 0, None, out_scale, has_mass, scratch, slice_offset=ka)
            nnz = st[_lib.STAT_NNZ]
            if prepared["unit_fast"] and need_xp and 0 < nnz * S * 4 <= module.store_sum_budget(X.device):
                # store-and-sum: every neighbour's key gradient is stored once (plain stores), then summed sender by sender over
                # the sender-major entry list -- no float atomics, reproducible gradients
                gkey = torch.empty((nnz, S), dtype=torch.float32, device=X.device)
                cptr, order = graph.sender_major()
                _lib.check(L.fsw_embed_backward_keys_f32(ctypes.byref(a), _lib.ptr(dtable), _lib.ptr(g), g.stride(0), _lib.ptr(gkey), S,
                                                         _lib.ptr(gf), stream), "fsw_embed_backward_keys_f32")
                _lib.check(L.fsw_segment_sum_rows_f32(_lib.ptr(gkey), S, _lib.ptr(cptr), _lib.ptr(order), graph.num_cols, nnz, S,
                                                      _lib.ptr(gXp), ldp, stream), "fsw_segment_sum_rows_f32")
                del gkey
            else:
                _lib.check(L.fsw_embed_backward_f32(ctypes.byref(a), _lib.ptr(dtable), _lib.ptr(g), g.stride(0), _lib.ptr(gXp), ldp,
                                                    _lib.ptr(gf), stream), "fsw_embed_backward_f32")
            if ctx.needs_input_grad[0]:
                gX = _grad_x(L, gXp, S, ldp, V[:, :module.d_in], stream)
            if ctx.needs_input_grad[1]:
                gVb = gemm_tn(gXp[:, :S], X.detach())
            if ctx.needs_input_grad[2]:
                gfb = gf
        # block gradients -> full-size parameter gradients (zero outside the block), summed over the ranks when sharded
        if ctx.needs_input_grad[0] and gX is None:
            gX = torch.zeros_like(X)
        if ctx.needs_input_grad[5] and gEf is None:
            gEf = torch.zeros((edge_feat_rows(ctx), module.d_edge), dtype=X.dtype, device=X.device)
        if ctx.needs_input_grad[1]:
            gV = torch.zeros_like(Vfull)
            if gVb is not None:
                gV[ka:kb] = gVb
        if ctx.needs_input_grad[2]:
            gfreqs = torch.zeros_like(freqs_full)
            if gfb is not None:
                gfreqs[ka:kb] = gfb
        if ctx.needs_input_grad[3]:
            gbias = torch.zeros(has_mass + module.nSlices, dtype=g.dtype, device=g.device)
            gb = out_scale * g.sum(dim=0)
            if has_mass and ((not sharded) or dist_rank(group) == 0):
                gbias[0] = gb[0]                       # replicated column: counted once in the sum over the ranks
            gbias[has_mass + ka:has_mass + kb] = gb[has_mass:]
        if ctx.needs_input_grad[4]:
            deg = (graph.rowptr[1:] - graph.rowptr[:-1]).long()
            if graph.w is None:
                m = deg.to(torch.float32)                                  # unit weights: total mass = in-degree
            else:
                rows = torch.repeat_interleave(torch.arange(graph.num_rows, device=X.device), deg)
                m = torch.zeros(graph.num_rows, dtype=torch.float32, device=X.device).index_add_(0, rows, graph.w[:rows.numel()])
            fn = module.total_mass_encoding_function
            fm = m if fn == 'identity' else (2 * (m / (torch.sqrt(m + 1) + 1)) if fn == 'sqrt' else torch.log1p(m))
            gscale = (out_scale * (g[:, 0] * fm).sum()).reshape(())     # identical on every rank: no reduction
        if sharded:
            import torch.distributed as dist
            for t in (gX, gV, gfreqs, gbias, gEf):
                if t is not None:
                    dist.all_reduce(t, group=group)
        return gX, gV, gfreqs, gbias, gscale, gEf, None, None, None, None, None, None


def dist_rank(group):
    import torch.distributed as dist
    return dist.get_rank(group)


class SimpleCSR:
    """Plain CSR adjacency for the generic kernels (csrc/embed_generic.hip): rowptr int32 [num_rows + 1], col int32 [nnz] with
    the entries in the order of the coalesced COO tensor they came from (row-major), max_degree on the host."""

    def __init__(self, rec, snd, num_rows, num_cols):
        dev = rec.device
        counts = torch.bincount(rec, minlength=num_rows) if rec.numel() else torch.zeros(num_rows, dtype=torch.int64, device=dev)
        self.rowptr = torch.zeros(num_rows + 1, dtype=torch.int32, device=dev)
        self.rowptr[1:] = torch.cumsum(counts, 0).to(torch.int32)
        self.col = snd.to(torch.int32).contiguous()
        self.rec = rec
        self.num_rows, self.num_cols, self.nnz = num_rows, num_cols, int(rec.numel())
        self.max_degree = int(counts.max()) if num_rows > 0 and rec.numel() else 0
        if fsw_embedding_basic_safety_checks and rec.numel():
            assert bool((rec[1:] >= rec[:-1]).all()), 'adjacency entries must be sorted by recipient (coalesced COO order)'
            assert int(snd.min()) >= 0 and int(snd.max()) < num_cols and int(rec.min()) >= 0 and int(rec.max()) < num_rows, \
                "adjacency index out of range"


def _generic_args(module, csr, Xp, ldp, Ke, freqs, wvals, out_scale, has_mass, mass_scale, scratch):
    a = _lib.GenericArgs()
    a.value_dtype = 1 if Xp.dtype == torch.float64 else 0
    a.S = freqs.numel()
    a.rowptr, a.col = csr.rowptr.data_ptr(), csr.col.data_ptr() if csr.nnz else None
    a.w = wvals.data_ptr() if wvals is not None else None
    a.num_rows, a.max_degree = csr.num_rows, csr.max_degree
    a.Xp, a.ldp = Xp.data_ptr(), ldp
    a.Ke, a.ldke = (Ke.data_ptr(), Ke.stride(0)) if Ke is not None else (None, 0)
    a.freqs, a.tau = freqs.data_ptr(), float(module.total_mass_pad_thresh)
    a.out_scale, a.has_mass = float(out_scale), has_mass
    a.mass_fn, a.mass_scale = _MASS_FN[module.total_mass_encoding_function], float(mass_scale)
    a.scratch, a.scratch_bytes = scratch.data_ptr(), scratch.numel()
    return a


class _GenericEmbedFn(torch.autograd.Function):
    """out = out_scale * E(X, W) through the generic kernels (csrc/embed_generic.hip: any degree, float32 or float64 storage,
    float64 arithmetic) with gradients for X, projVecs, freqs, bias, the total-mass scale, the WEIGHTS and the edge features.

    Two callers: every float64 module (the float64 build of the path, reference test_conv.py:24), and float32 modules whose
    weights require a gradient (the tuned float32 backward kernels treat W as a constant).  Replaces the reference's sparse
    autograd chain incl. ag.div_sparse_dense.backward (fsw_embedding.py:1656), ag.cumsum_sparse.backward (:2160, the reverse
    segmented cumsum) and ag.permute_sparse.backward (:1286)."""

    @staticmethod
    def forward(ctx, X, V, freqs, bias, mass_scale, wvals, efvals, module, csr, out_scale):
        L = _lib.lib()
        dev, dt = X.device, X.dtype
        stream = torch.cuda.current_stream(dev).cuda_stream
        S, d_in = module.nSlices, module.d_in
        has_mass = 1 if module.encode_total_mass else 0
        with torch.no_grad():
            Xc, Vd, fr = X.detach().contiguous(), V.detach(), freqs.detach().contiguous()
            ldp = _round_up(S, 64)
            Xp = torch.empty((Xc.shape[0], ldp), dtype=dt, device=dev)
            if dt == torch.float64:
                _lib.check(L.fsw_project_f64(_lib.ptr(Xc), Xc.shape[0], d_in, Xc.stride(0), _lib.ptr(Vd), S, Vd.stride(0), _lib.ptr(Xp), ldp,
                                             None, stream), "fsw_project_f64")
            else:
                _lib.check(L.fsw_project_f32(_lib.ptr(Xc), Xc.shape[0], d_in, Xc.stride(0), _lib.ptr(Vd), S, Vd.stride(0), _lib.ptr(Xp), ldp,
                                             None, 0, None, stream), "fsw_project_f32")
            Ke = None
            if efvals is not None:
                Ke = (efvals.detach() @ Vd[:, d_in:].t()).contiguous()        # [nnz, S], the reference's E x S edge term (:934-968)
            wv = wvals.detach().contiguous() if wvals is not None else None
            scratch = torch.empty(int(L.fsw_embed_generic_scratch_bytes(csr.max_degree, max(csr.num_rows, 1))), dtype=torch.uint8, device=dev)
            out = torch.empty((csr.num_rows, has_mass + S), dtype=dt, device=dev)
            ms = float(mass_scale.detach()) if mass_scale is not None else 1.0
            a = _generic_args(module, csr, Xp, ldp, Ke, fr, wv, out_scale, has_mass, ms, scratch)
            a.out, a.ldo = out.data_ptr(), out.stride(0)
            b = bias.detach().contiguous() if bias is not None else None
            a.bias = b.data_ptr() if b is not None else None
            _lib.check(L.fsw_embed_generic(ctypes.byref(a), stream), "fsw_embed_generic")
        ctx.module, ctx.csr, ctx.out_scale, ctx.mass_scale_value = module, csr, float(out_scale), ms
        ctx.aux = (Xp, ldp, Ke, wv)
        ctx.has_ef, ctx.has_w = efvals is not None, wvals is not None
        ctx.save_for_backward(X, V, freqs, efvals if efvals is not None else X.new_zeros(0))
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        L = _lib.lib()
        module, csr, out_scale = ctx.module, ctx.csr, ctx.out_scale
        X, V, freqs, efvals = ctx.saved_tensors
        Xp, ldp, Ke, wv = ctx.aux
        dev, dt = X.device, X.dtype
        stream = torch.cuda.current_stream(dev).cuda_stream
        S, d_in = module.nSlices, module.d_in
        has_mass = 1 if module.encode_total_mass else 0
        g = g.contiguous()
        need = ctx.needs_input_grad
        gX = gV = gfreqs = gbias = gscale = gW = gEf = None
        nnz = csr.nnz
        want_key = need[0] or need[1] or (ctx.has_ef and need[6])
        want_w = ctx.has_w and need[5]
        if nnz and S and (want_key or need[2] or want_w):
            gkey = torch.zeros((nnz, S), dtype=dt, device=dev) if want_key else None
            gf = torch.zeros(S, dtype=dt, device=dev) if need[2] else None
            gw = torch.zeros(nnz, dtype=dt, device=dev) if want_w else None
            scratch = torch.empty(int(L.fsw_embed_generic_scratch_bytes(csr.max_degree, max(csr.num_rows, 1))), dtype=torch.uint8, device=dev)
            a = _generic_args(module, csr, Xp, ldp, Ke, freqs.detach().contiguous(), wv, out_scale, has_mass, ctx.mass_scale_value, scratch)
            a.g, a.ldg = g.data_ptr(), g.stride(0)
            a.gkey, a.ldk = (gkey.data_ptr(), S) if gkey is not None else (None, 0)
            a.gfreq = gf.data_ptr() if gf is not None else None
            a.gw = gw.data_ptr() if gw is not None else None
            _lib.check(L.fsw_embed_generic(ctypes.byref(a), stream), "fsw_embed_generic (backward)")
            Vd = V.detach()
            if need[0] or need[1]:
                gXp = torch.zeros((X.shape[0], S), dtype=dt, device=dev).index_add_(0, csr.col.long(), gkey)
                if need[0]:
                    gX = gXp @ Vd[:, :d_in]
                if need[1]:
                    gV = gXp.t() @ X.detach()
                    if ctx.has_ef:
                        gV = torch.cat([gV, gkey.t() @ efvals.detach()], dim=1)
            if ctx.has_ef and need[6]:
                gEf = gkey @ Vd[:, d_in:]
            gfreqs, gW = gf, gw
        if need[0] and gX is None:
            gX = torch.zeros_like(X)
        if need[1] and gV is None:
            gV = torch.zeros_like(V)
        if need[2] and gfreqs is None:
            gfreqs = torch.zeros_like(freqs)
        if ctx.has_ef and need[6] and gEf is None:
            gEf = torch.zeros_like(efvals)
        if need[3]:
            gbias = out_scale * g.sum(dim=0)
        if has_mass and (need[4] or want_w):
            m = torch.zeros(csr.num_rows, dtype=dt, device=dev)
            if nnz:
                m.index_add_(0, csr.rec, wv if wv is not None else torch.ones(nnz, dtype=dt, device=dev))
            fn = module.total_mass_encoding_function
            if need[4]:
                fm = m if fn == 'identity' else (2 * (m / (torch.sqrt(m + 1) + 1)) if fn == 'sqrt' else torch.log1p(m))
                gscale = (out_scale * (g[:, 0] * fm).sum()).reshape(())
            if want_w:      # the total-mass column depends on the weights through m
                dfm = torch.ones_like(m) if fn == 'identity' else (1 / torch.sqrt(m + 1) if fn == 'sqrt' else 1 / (1 + m))
                gW = (gW if gW is not None else torch.zeros(nnz, dtype=dt, device=dev)) + (out_scale * ctx.mass_scale_value * g[:, 0] * dfm)[csr.rec]
        if want_w and gW is None:
            gW = torch.zeros(nnz, dtype=dt, device=dev)
        return gX, gV, gfreqs, gbias, gscale, gW, gEf, None, None, None


class FSW_embedding(nn.Module):
    def __init__(self,
                 d_in, d_out=None, nSlices=None, nFreqs=None, collapse_freqs=False,
                 d_edge=0,
                 encode_total_mass=False,
                 total_mass_encoding_function='identity',
                 total_mass_encoding_scale=1.0,
                 total_mass_encoding_method='plain',
                 total_mass_pad_thresh=1.0,
                 learnable_slices=False, learnable_freqs=False, learnable_total_mass_encoding_scale=False,
                 freqs_init='random',
                 minimize_slice_coherence=False,
                 enable_bias=True,
                 device=None, dtype=torch.float32,
                 load_custom_cuda_lib=True,
                 report=False, user_warnings=True,
                 report_on_coherence_minimization=False):
        super().__init__()
        self.user_warnings = user_warnings
        # reference fsw_embedding.py:195-206 loads its CUDA library here; this build cannot run without its
        # native library, so load_custom_cuda_lib=False is rejected instead of selecting a slow path.
        if not load_custom_cuda_lib:
            raise RuntimeError("fsw_gnn_amd has no pure-PyTorch path: load_custom_cuda_lib=False is not supported")
        _lib.lib()

        assert d_in >= 0, 'd_in must be nonnegative'
        assert d_edge >= 0, 'd_edge must be nonnegative'
        assert (d_out is None) or (d_out >= 0), 'd_out must be nonnegative or None'
        if d_out == 0:
            encode_total_mass = False
        self.d_in, self.d_edge = d_in, d_edge
        self.encode_total_mass = bool(encode_total_mass)
        self.total_mass_encoding_dim = 1 if self.encode_total_mass else 0
        self.total_mass_encoding_scale_init = total_mass_encoding_scale

        total_mass_pad_thresh = float(total_mass_pad_thresh)
        assert not np.isinf(total_mass_pad_thresh), 'total_mass_pad_thresh cannot be inf'
        assert not np.isnan(total_mass_pad_thresh), 'total_mass_pad_thresh cannot be NaN'
        assert total_mass_pad_thresh > 0, 'total_mass_pad_thresh must be positive'
        self.total_mass_pad_thresh = total_mass_pad_thresh
        assert total_mass_encoding_method in {'plain', 'homog', 'homog_alt'}, \
            "<total_mass_encoding_method> must be one of 'plain', 'homog', 'homog_alg'"
        self.total_mass_encoding_method = total_mass_encoding_method
        assert total_mass_encoding_function in {'identity', 'sqrt', 'log'}, \
            "<total_mass_encoding_function> must be one of 'identity', 'sqrt', 'log'"
        self.total_mass_encoding_function = total_mass_encoding_function

        # size logic: reference fsw_embedding.py:242-261
        if (d_out is not None) and (nSlices is None) and (nFreqs is None):
            self.cartesian_mode = False
            self.collapse_freqs = False
            self.d_out = d_out
            self.nSlices = d_out - self.total_mass_encoding_dim
            self.nFreqs = d_out - self.total_mass_encoding_dim
        elif (d_out is None) and (nSlices is not None) and (nFreqs is not None):
            raise NotImplementedError("fsw_gnn_amd: Cartesian mode (nSlices x nFreqs) is out of scope (SURVEY.md 2, #14)")
        else:
            assert False, "Expected exactly one of (d_out != None) or (nSlices != None and nFreqs != None)"
        assert self.d_out >= 0, 'd_out must be nonnegative'
        self.minimize_slice_coherence = minimize_slice_coherence
        self.learnable_slices = learnable_slices
        self.learnable_freqs = learnable_freqs
        self.learnable_total_mass_encoding_scale = learnable_total_mass_encoding_scale
        self.freqs_init = freqs_init
        self.enable_bias = enable_bias
        if device is None:
            device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.device_new = torch.device(device)
        assert dtype.is_floating_point and (not dtype.is_complex), \
            'dtype must be real floating-point; instead got dtype=%s' % (dtype)
        if dtype not in (torch.float32, torch.float64):
            raise NotImplementedError("fsw_gnn_amd: the HIP kernels are built for float32 and float64 (got dtype=%s)" % dtype)
        self.dtype_new = dtype
        self.report = report
        self.report_on_coherence_minimization = report_on_coherence_minimization
        if report:
            print('Fourier Sliced-Wasserstein Embedding, MI355X-native build %s' % version)
            print('Using %d (slice, frequency) pairs; device: %s    dtype: %s' % (self.nSlices, self.device_new, dtype))
        self.reset_parameters()

    # ------------------------------------------------------------------------------------------------
    def reset_parameters(self, freqs_init=None, minimize_slice_coherence=None, report=None,
                         report_on_coherence_minimization=None):
        """Reference fsw_embedding.py:343-414 / 445-559 (parameter semantics; RNG values differ)."""
        self.freqs_init = self.freqs_init if freqs_init is None else freqs_init
        if minimize_slice_coherence is not None:
            self.minimize_slice_coherence = minimize_slice_coherence
        if report is not None:
            self.report = report
        if hasattr(self, 'device_new'):
            device = self.device_new
            delattr(self, 'device_new')
        else:
            device = self.get_device()
        if hasattr(self, 'dtype_new'):
            dtype = self.dtype_new
            delattr(self, 'dtype_new')
        else:
            dtype = self.get_dtype()
        projVecs, freqs, bias, scale = FSW_embedding.generate_embedding_parameters(
            d_in=self.d_in + self.d_edge, nSlices=self.nSlices, nFreqs=self.nFreqs,
            total_mass_encoding_dim=self.total_mass_encoding_dim,
            total_mass_encoding_scale_init=self.total_mass_encoding_scale_init, freqs_init=self.freqs_init, device=device,
            minimize_slice_coherence=self.minimize_slice_coherence, report=self.report_on_coherence_minimization)
        self.projVecs = nn.Parameter(projVecs.to(dtype=dtype, device=device), requires_grad=self.learnable_slices)
        self.freqs = nn.Parameter(freqs.to(dtype=dtype, device=device), requires_grad=self.learnable_freqs)
        if self.enable_bias:
            self.bias = nn.Parameter(bias.to(dtype=dtype, device=device), requires_grad=self.learnable_slices)
        if self.encode_total_mass:
            self.total_mass_encoding_scale = nn.Parameter(scale.to(dtype=dtype), requires_grad=self.learnable_total_mass_encoding_scale)
        return self

    @staticmethod
    def generate_embedding_parameters(d_in, nSlices, nFreqs, total_mass_encoding_dim, total_mass_encoding_scale_init,
                                      freqs_init, device, minimize_slice_coherence=False, report=False):
        dt = torch.float64
        # A. random unit projection vectors (reference :448-456), optionally spread out by mutual-coherence
        #    minimisation (:464, :3045-3248 -> coherence.py)
        projVecs = torch.randn(size=(nSlices, d_in), dtype=dt, device=device)
        projVecs = nn.functional.normalize(projVecs, p=2.0, dim=1, eps=0)
        if minimize_slice_coherence and nSlices > 1 and d_in > 0:
            projVecs = minimize_mutual_coherence(projVecs, report=report)
        if nSlices > 0 and d_in > 0:
            assert not torch.isinf(projVecs).any(), "Found infs in projVecs"
            assert not torch.isnan(projVecs).any(), "Found nans in projVecs"
        # B. frequencies (reference :489-535)
        shape = (nFreqs,)
        if nFreqs == 0:
            freqs = torch.zeros(size=shape, dtype=dt, device=device)
        elif isinstance(freqs_init, numbers.Real):
            assert not np.isinf(freqs_init), 'freqs_init cannot be infinite'
            assert not np.isnan(freqs_init), 'freqs_init cannot be NaN'
            freqs = freqs_init * torch.ones(size=shape, dtype=dt, device=device)
        elif isinstance(freqs_init, tuple):
            assert (len(freqs_init) == 2), 'When freqs_init is a tuple, it must be of length 2'
            a, b = freqs_init
            assert not np.isinf(a) and not np.isinf(b), 'Received infinite value in freqs_init tuple'
            assert not np.isnan(a) and not np.isnan(b), 'Received NaN value in freqs_init tuple'
            assert a <= b, 'When freqs_init is a tuple, it is required to satisfy freqs_init[0] <= freqs_init[1]'
            if nFreqs == 1:
                freqs = a + (b - a) / 2 * torch.ones(size=shape, dtype=dt, device=device)
            else:
                freqs = a + (b - a) * (torch.arange(nFreqs, dtype=dt, device=device) / (nFreqs - 1))
        elif freqs_init == 'random':
            freqs = torch.rand(size=shape, dtype=dt, device=device)
            freqs, _ = torch.sort(freqs, dim=0)
            freqs = freqs / (1 - freqs)
        elif freqs_init == 'spread':
            freqs = (0.5 + torch.arange(nFreqs, dtype=dt, device=device)) / nFreqs
            freqs = freqs / (1 - freqs)
        else:
            raise RuntimeError("Invalid value for argument freqs_init; expected number, tuple (a,b) of numbers "
                               "denoting an interval, 'random' or 'spread'")
        if nFreqs > 0:
            assert not torch.isinf(freqs).any(), "Found infs in freqs"
            assert not torch.isnan(freqs).any(), "Found nans in freqs"
        # C. zero bias (reference :542-550) and the total-mass scale
        bias = torch.zeros(size=(nSlices + total_mass_encoding_dim,), dtype=dt, device=device)
        scale = torch.tensor(total_mass_encoding_scale_init, device=device, dtype=dt) if total_mass_encoding_dim > 0 else None
        return projVecs, freqs, bias, scale

    def spread_freqs_at_interval(self, center, radius):
        """Reference fsw_embedding.py:567-582."""
        assert radius >= 0
        if (self.nFreqs == 1) or (radius == 0):
            freqs_new = center * torch.ones_like(self.freqs)
        else:
            spread = 2 * (0.5 + torch.arange(self.nFreqs, dtype=self.get_dtype(), device=self.get_device())) / self.nFreqs - 1
            spread = spread * 1 / (1 - 1 / self.nFreqs)
            freqs_new = center + radius * spread
        sd = self.state_dict()
        sd['freqs'] = freqs_new
        self.load_state_dict(sd)
        return self

    def get_device(self):
        return self.projVecs.device

    def get_dtype(self):
        return self.projVecs.dtype

    # ------------------------------------------------------------------------------------------------
    def forward(self, X, W='unit', X_edge=None, graph_mode=False, serialize_num_slices=None):
        """Same contract as the reference forward (fsw_embedding.py:587-890).

        X (<batch>, n, d_in); W (<batch>, n) | 'unit' | 'uniform'; graph_mode: W (<batch>, nRecipients, n) dense or
        coalesced torch.sparse_coo, X shared by the recipients.  Returns (<batch>, [nRecipients,] d_out).
        """
        assert self.total_mass_pad_thresh > 0, 'total_mass_pad_thresh must be positive'
        if self.d_edge > 0:
            assert graph_mode, 'd_edge > 0 (given at initialization) necessitates graph_mode=True on forward call'
            assert X_edge is not None, 'X_edge must be provided since d_edge > 0'
            assert torch.is_tensor(W), 'When X_edge is provided, W must be provided explicitly'
            assert X_edge.device == self.get_device() and X_edge.dtype == self.get_dtype(), 'X_edge has the wrong device or dtype'
            assert X_edge.is_sparse == W.is_sparse, 'X_edge and W must either both or neither be sparse'
        else:
            assert (X_edge is None) or (X_edge.numel() == 0), 'X_edge should be None or empty since d_edge == 0'
        assert torch.is_tensor(X), 'X must be a pytorch tensor. Instead got type %s' % (type(X))
        assert torch.is_tensor(W) or W in {'unit', 'uniform'}, "W must be a pytorch tensor, 'unit' or 'uniform'"
        assert X.dtype == self.get_dtype(), ("X has the wrong dtype. Expected %s, got %s" % (self.get_dtype(), X.dtype))
        assert X.device == self.get_device(), ("X is on the wrong device. Expected %s, got %s" % (self.get_device(), X.device))
        if X.device.type != 'cuda':
            raise RuntimeError("fsw_gnn_amd: forward needs tensors on a HIP device ('cuda'); there is no CPU path")
        needs_grad = torch.is_grad_enabled() and (X.requires_grad or any(p.requires_grad for p in self.parameters()))
        # float64 modules, and float32 calls that ask for gradients of the weights or of sparse edge features, run on the
        # generic kernels (csrc/embed_generic.hip); everything else on the tuned float32 kernels
        generic = self.get_dtype() == torch.float64 or (torch.is_grad_enabled() and (
            (torch.is_tensor(W) and W.requires_grad) or (X_edge is not None and torch.is_tensor(X_edge) and X_edge.requires_grad)))
        if torch.is_tensor(W):
            assert W.dtype == self.get_dtype(), ("W has the wrong dtype. Expected %s, got %s" % (self.get_dtype(), W.dtype))
            assert W.device == self.get_device(), ("W is on the wrong device. Expected %s, got %s" % (self.get_device(), W.device))
            if W.is_sparse or W.layout != torch.strided:
                assert W.layout == torch.sparse_coo, ("Sparse W has an unsupported sparsity layout '%s'. Only the COO "
                                                      "layout (torch.sparse_coo) is currently supported." % (W.layout))
                assert W.is_coalesced(), 'Sparse W must be coalesced'
                assert W.dense_dim() == 0, 'W.dense_dim() must be zero'
        assert len(X.shape) >= 2, "X must be a tensor of order at least 2"
        assert X.shape[-1] == self.d_in, "The last dimension of X must equal d_in=%d. Instead got %d" % (self.d_in, X.shape[-1])

        d = self.d_in
        efvals = None
        if not graph_mode:
            batch_dims = tuple(X.shape[0:-2])
            n = X.shape[-2]
            B = int(np.prod(batch_dims)) if batch_dims else 1
            if torch.is_tensor(W):
                assert (len(W.shape) == len(X.shape) - 1) and (tuple(W.shape) == tuple(X.shape[0:-1])), \
                    "Shape mismatch between X and W: If X.shape = (b1,b2,...,bk,n,d_in) then W.shape should be (b1,b2,...,bk,n) (unless graph_mode=True)"
                wvals = None if W.is_sparse else W.reshape(-1)
            elif W == 'unit':
                wvals = None
            else:  # 'uniform'
                wvals = torch.full((B * n,), 1.0 / n, dtype=self.get_dtype(), device=X.device)
            if torch.is_tensor(W) and W.is_sparse:
                # coalesced COO weights (reference fsw_embedding.py:664-668 accepts them in both modes): entry (b.., j)
                # is element j of multiset b; absent entries are absent elements (weight 0)
                idx, wvals = W.indices(), W.values()
                if batch_dims:
                    strides = torch.tensor(list(np.cumprod((batch_dims + (1,))[::-1])[::-1][1:]), device=X.device, dtype=torch.int64)
                    rec = (idx[:len(batch_dims)] * strides[:, None]).sum(0)
                else:
                    rec = torch.zeros(idx.shape[1], device=X.device, dtype=torch.int64)
                snd = (rec * n + idx[-1]).contiguous()
                rec = rec.contiguous()
            else:
                rec = torch.arange(B, device=X.device, dtype=torch.int64).repeat_interleave(n)
                snd = torch.arange(B * n, device=X.device, dtype=torch.int64)
            Xf = X.reshape(B * n, d)
            num_rows, out_shape = B, batch_dims
        else:
            assert torch.is_tensor(W), 'W must be explicitly provided when graph_mode=True'
            batch_dims = tuple(W.shape[0:-2])
            nR, n = W.shape[-2], W.shape[-1]
            assert (len(W.shape) == len(X.shape)) and (W.shape[-1] == X.shape[-2]) and (tuple(W.shape[0:-2]) == tuple(X.shape[0:-2])), \
                "Shape mismatch between X and W: When graph_mode=True, if W.shape = (b1,b2,...,bk,nRecipients,n) then X.shape should be (b1,b2,...,bk,n,d_in)"
            B = int(np.prod(batch_dims)) if batch_dims else 1
            efvals = None
            if W.is_sparse:
                idx, wvals = W.indices(), W.values()
                if self.d_edge > 0:
                    assert X_edge.is_coalesced() and X_edge.values().shape[0] == wvals.shape[0], \
                        'Sparse X_edge must have the same number of values() as W'
                    if fsw_embedding_basic_safety_checks:
                        assert (X_edge.indices() == idx).all(), 'Sparse X_edge must have the same nonzero pattern as W'
                    efvals = X_edge.values().reshape(wvals.shape[0], -1)
            else:
                idx = W.nonzero(as_tuple=False).t().contiguous()
                wvals = W[tuple(idx)]
                if self.d_edge > 0:
                    efvals = X_edge[tuple(idx)].reshape(wvals.shape[0], -1)
            if self.d_edge > 0:
                assert efvals.shape[1] == self.d_edge, 'X_edge must carry d_edge features per edge'
            if batch_dims:
                strides = torch.tensor(list(np.cumprod((batch_dims + (1,))[::-1])[::-1][1:]), device=X.device, dtype=torch.int64)
                b = (idx[:len(batch_dims)] * strides[:, None]).sum(0)
            else:
                b = 0
            rec = (b * nR + idx[-2]).contiguous()
            snd = (b * n + idx[-1]).contiguous()
            Xf = X.reshape(B * n, d)
            num_rows, out_shape = B * nR, batch_dims + (nR,)

        if generic and num_rows > 0 and Xf.shape[0] > 0:
            if self.d_out == 0:
                return torch.zeros(out_shape + (0,), dtype=X.dtype, device=X.device)
            return self._forward_generic(Xf.contiguous(), rec, snd, wvals, efvals, num_rows).reshape(out_shape + (self.d_out,))
        if num_rows == 0 or Xf.shape[0] == 0:
            # nothing to embed / empty multisets only: the reference returns an empty tensor, resp. the embedding of the
            # pad element alone; the kernels need at least one row and one point
            if num_rows == 0:
                return torch.zeros(out_shape + (self.d_out,), dtype=X.dtype, device=X.device)
            Xf = torch.zeros((1, d), dtype=X.dtype, device=X.device)
        if self.d_edge > 0:
            graph = build_csr_coalesced(rec, snd, wvals.contiguous(), efvals.contiguous(), num_rows, Xf.shape[0])
        else:
            graph = build_csr(rec, snd, wvals, num_rows, Xf.shape[0])
        if needs_grad:
            out = self.embed_autograd(Xf.contiguous(), graph)
        else:
            out = torch.empty((num_rows, self.d_out), dtype=X.dtype, device=X.device)
            self.embed_into(Xf, graph, out, out_scale=1.0, serialize_num_slices=serialize_num_slices)
        return out.reshape(out_shape + (self.d_out,))

    def _forward_generic(self, Xf, rec, snd, wvals, efvals, num_rows):
        """The generic-kernel path (float64 modules; float32 with gradients of W / X_edge): differentiable in everything."""
        if self.nSlices == 0:
            raise NotImplementedError("fsw_gnn_amd: nSlices == 0 with encode_total_mass is not supported")
        if fsw_embedding_basic_safety_checks:      # reference fsw_embedding.py:652-703
            assert bool(torch.isfinite(Xf).all()), "The entries of X cannot contain NaNs or infs"
            if wvals is not None:
                assert bool(torch.isfinite(wvals).all()), "All entries of W must be finite"
                assert bool((wvals >= 0).all()), "All entries of W must be nonnegative"
        csr = SimpleCSR(rec, snd, num_rows, Xf.shape[0])
        bias = self.bias if self.enable_bias else None
        scale = self.total_mass_encoding_scale if self.encode_total_mass else None
        plain = (not self.encode_total_mass) or self.total_mass_encoding_method == 'plain'
        P = _GenericEmbedFn.apply(Xf, self.projVecs, self.freqs, bias if plain else None, scale, wvals, efvals, self, csr, 1.0)
        return P if plain else self._homog_epilogue(P, 1.0, bias)

    def _homog_epilogue(self, P, out_scale, bias):
        """'homog' / 'homog_alt' (reference fsw_embedding.py:874-882, 1136-1144) on the 'plain' embedding P = [f(m) scale | emb]
        without bias, out of place so that autograd differentiates it."""
        tm = P[:, 0:1] / out_scale
        emb = P[:, 1:]
        norm = emb.abs().mean(dim=-1, keepdim=True) / out_scale
        if self.total_mass_encoding_method == 'homog':
            col0 = out_scale * tm * norm
        else:
            col0 = out_scale * torch.where(tm <= 1, tm * (2 - tm), torch.ones_like(tm)) * norm
            emb = emb * torch.where(tm <= 1, tm.square(), 2 * tm - 1)
        out = torch.cat([col0, emb], dim=1)
        return out + out_scale * bias if bias is not None else out

    def embed_autograd(self, X, graph, out_scale=1.0, edge_feat=None, slice_range=None, group=None, reduce_grads=False):
        """Differentiable embedding of a CSR graph (training path): see _EmbedGraphFn.  edge_feat: the per-input-edge
        feature tensor the graph was coalesced from (its gradient is routed back through graph.slot_of_edge).
        slice_range / group: this rank's block of slices under slice sharding (dist.py)."""
        bias = self.bias if self.enable_bias else None
        scale = self.total_mass_encoding_scale if self.encode_total_mass else None
        if (not self.encode_total_mass) or self.total_mass_encoding_method == 'plain':
            return _EmbedGraphFn.apply(X, self.projVecs, self.freqs, bias, scale, edge_feat, self, graph, out_scale, slice_range, group,
                                       reduce_grads)
        if slice_range is not None:
            raise NotImplementedError("slice sharding supports total_mass_encoding_method='plain' only")
        # 'homog' / 'homog_alt' (reference fsw_embedding.py:874-882, 1136-1144): the 'plain' embedding without bias from the
        # kernels, then the same epilogue as embed_into(), out of place so that autograd differentiates it
        self._force_plain = True
        try:
            P = _EmbedGraphFn.apply(X, self.projVecs, self.freqs, None, scale, edge_feat, self, graph, out_scale, None, None, False)
        finally:
            self._force_plain = False
        return self._homog_epilogue(P, out_scale, bias)

    # ------------------------------------------------------------------------------------------------
    # backward of unit-weight graphs: store the key gradients ([nnz, nSlices] float32) and sum them sender by sender instead of
    # float atomics, as long as that buffer stays below this size (10.2 GB at 10M edges x 256 slices); 0 = always atomics
    store_sum_backward_max_bytes = 32 << 30
    store_sum_free_memory_fraction = 0.25   # ... and below this share of the device memory that is free right now

    def store_sum_budget(self, device):
        """Bytes the transient [nnz, nSlices] key-gradient buffer of the store-and-sum backward may take: the fixed ceiling,
        capped by a quarter of the currently free device memory (free in the driver's sense + what torch's caching allocator
        holds unused), so that a training run that fits with the atomic backward never runs out of memory inside backward."""
        cap = int(self.store_sum_backward_max_bytes)
        if cap <= 0:
            return 0
        try:
            free, _ = torch.cuda.mem_get_info(device)
            cached = torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device)
            cap = min(cap, int((free + max(cached, 0)) * self.store_sum_free_memory_fraction))
        except Exception:   # noqa: BLE001 -- no query, keep the fixed ceiling
            pass
        return cap
    _force_plain = False   # embed_autograd: 'plain' mass column and no bias from the kernels, epilogue in torch

    def prepare(self, X, graph: CSRGraph, x_copy=None, linear2=None, slice_range=None):
        """Projection of a block of slices (default: all) + (unit weights) coefficient table + the one device->host stats
        read of the forward.

        linear2 = (W2 [H2, d_in] contiguous, b2 [H2] or None, Y2 [n, H2]): the projection GEMM also writes
        Y2[graph.invperm[i]] = X[i] . W2^T + b2, the vertex-feature half of FSW_conv's first Linear layer in the
        row order of the degree bins (csrc/conv_fused.hip reads it back as contiguous runs).
        x_copy (optional [num_cols, d_in] view with unit inner stride): the projection kernel also stores X there
        (FSW_conv's concat buffer, reference fsw_conv.py:357-358).

        Returns a dict that embed_into(prepared=...) or FSW_conv's fused Linear path consume.  Input validation
        (reference fsw_embedding.py:652-703) happens here: the kernels set flag bits, the host reads them once.
        """
        L = _lib.lib()
        dev = X.device
        ka, kb = (0, self.nSlices) if slice_range is None else slice_range
        assert 0 <= ka <= kb <= self.nSlices, 'bad slice_range'
        S = kb - ka
        assert X.is_contiguous()
        stream = torch.cuda.current_stream(dev).cuda_stream
        unit_fast = graph.w is None and self.total_mass_pad_thresh <= 1.0
        if S == 0:      # a rank without slices (more ranks than slices): only the stats
            if x_copy is not None:
                x_copy.copy_(X)
            return {"Xp": None, "ldp": 0, "table": None, "stats": self._checked_stats(graph), "unit_fast": unit_fast,
                    "slice_range": (ka, kb)}
        ldp = _round_up(S, 64)
        Xp = torch.empty((X.shape[0], ldp), dtype=torch.float32, device=dev)
        V = self.projVecs.detach()[ka:kb]
        if linear2 is not None:
            W2, b2, Y2 = linear2
            assert x_copy is None and W2.is_contiguous() and W2.shape[1] == self.d_in and Y2.stride(1) == 1
            rc = L.fsw_project_linear_f32(_lib.ptr(X), X.shape[0], self.d_in, X.stride(0), _lib.ptr(V), S, V.stride(0),
                                          _lib.ptr(Xp), ldp, _lib.ptr(W2), W2.shape[0], W2.stride(0), _lib.ptr(b2),
                                          _lib.ptr(Y2), Y2.stride(0), _lib.ptr(graph.invperm), _lib.ptr(graph.stats_dev), stream)
        else:
            rc = L.fsw_project_f32(_lib.ptr(X), X.shape[0], self.d_in, X.stride(0), _lib.ptr(V), S, V.stride(0), _lib.ptr(Xp),
                                   ldp, _lib.ptr(x_copy), x_copy.stride(0) if x_copy is not None else 0,
                                   _lib.ptr(graph.stats_dev), stream)
        _lib.check(rc, "fsw_project_f32")
        table = None
        if unit_fast:
            table = torch.empty((int(L.fsw_unit_table_rows(_lib.REG_MAX_DEG)), ldp), dtype=torch.float32, device=dev)
            fr = self.freqs.detach()[ka:kb]
            rc = L.fsw_unit_coeff_table(_lib.ptr(fr), S, _lib.REG_MAX_DEG, _lib.ptr(table), ldp, stream)
            _lib.check(rc, "fsw_unit_coeff_table")
        st = self._checked_stats(graph)
        return {"Xp": Xp, "ldp": ldp, "table": table, "stats": st, "unit_fast": unit_fast, "slice_range": (ka, kb)}

    def _checked_stats(self, graph):
        """The one device->host copy of a forward: validation flags, degree-class counts and -- parked in the spare stats
        word right before the copy -- the current value of the total-mass scale (read from the parameter every time: no
        host-side cache that an in-place edit of .data could outdate)."""
        if self.encode_total_mass:
            graph.stats_dev[_lib.STAT_USER:_lib.STAT_USER + 1].view(torch.float32).copy_(
                self.total_mass_encoding_scale.detach().reshape(1).to(torch.float32))
        st = graph.read_stats()
        if fsw_embedding_basic_safety_checks:
            fl = st[_lib.STAT_FLAGS]
            assert not (fl & _lib.FLAG_INDEX_RANGE), "adjacency index out of range"
            assert not (fl & _lib.FLAG_X_NONFINITE), "The entries of X cannot contain NaNs or infs"
            assert not (fl & _lib.FLAG_W_NONFINITE), "All entries of W must be finite"
            assert not (fl & _lib.FLAG_W_NEGATIVE), "All entries of W must be nonnegative"
        return st

    def make_args(self, graph, st, Xp, ldp, freqs, S, table, out_ptr, ldo, bias_ptr, out_scale, has_mass, scratch=None,
                  slice_offset=0, chunk=None):
        """struct fsw_embed_args for one call.  chunk = c restricts the call to the recipients of row chunk c of a graph
        built with chunk_rows > 0 (graph.py)."""
        a = _lib.EmbedArgs()
        a.rowptr, a.col = graph.rowptr.data_ptr(), graph.col.data_ptr()
        a.w = graph.w.data_ptr() if graph.w is not None else None
        a.perm = graph.perm.data_ptr()
        if chunk is None:
            assert graph.num_chunks == 1, 'a graph with row chunks is consumed chunk by chunk'
            a.bin_start, rows = graph.bin_start.data_ptr(), graph.num_rows
        else:
            a.bin_start = graph.bin_start.view(-1, _lib.NUM_BINS + 1)[chunk].data_ptr()
            rows = min(graph.chunk_rows, graph.num_rows - chunk * graph.chunk_rows) if graph.chunk_rows else graph.num_rows
        a.num_rows = rows
        bsh = getattr(graph, "bin_start_host", None)     # host copy of the bin table (read with the stats): exact grids
        if bsh is not None:
            a.bin_start_host = bsh[0 if chunk is None else chunk].ctypes.data
        a.Xp, a.ldp, a.freqs, a.S, a.tau = Xp.data_ptr(), ldp, freqs.data_ptr(), S, float(self.total_mass_pad_thresh)
        a.unit_table, a.ldt = (table.data_ptr() if table is not None else None), ldp
        a.out, a.ldo, a.bias = out_ptr, ldo, bias_ptr
        a.out_scale, a.has_mass = float(out_scale), has_mass
        a.mass_fn = _MASS_FN[self.total_mass_encoding_function]
        a.mass_scale = struct.unpack('f', struct.pack('i', st[_lib.STAT_USER]))[0] if self.encode_total_mass else 1.0
        a.num_reg_rows, a.num_lds_rows = min(st[_lib.STAT_NUM_REG], rows), min(st[_lib.STAT_NUM_LDS], rows)
        a.num_global_rows, a.num_zero_rows = min(st[_lib.STAT_NUM_GLOBAL], rows), min(st[_lib.STAT_NUM_ZERO], rows)
        a.max_degree = st[_lib.STAT_MAX_DEGREE]
        a.scratch = scratch.data_ptr() if scratch is not None else None
        a.scratch_bytes = scratch.numel() if scratch is not None else 0
        if graph.ef is not None:            # edge features: Ve = projVecs[:, d_in:], read in place (row stride d_in + d_edge)
            V = self.projVecs.detach()
            a.efeat, a.d_edge = graph.ef.data_ptr(), self.d_edge
            a.Ve, a.ldve = V.data_ptr() + 4 * (slice_offset * V.stride(0) + self.d_in), V.stride(0)
        return a

    def embed_into(self, X, graph: CSRGraph, out, out_scale=1.0, serialize_num_slices=None, slice_range=None, x_copy=None,
                   prepared=None, chunks=None, long_rows_only=False):
        """Writes out_scale * E(X, graph) into the left columns of `out` (row stride out.stride(0)).

        X [num_cols, d_in] float32 contiguous; out [num_rows, >= width] float32 with unit inner stride, where
        width = d_out, or total_mass_dim + (kb - ka) when slice_range = (ka, kb) restricts the call to a block of
        slices (multi-GPU slice sharding, dist.py: column 0 is still the total-mass column, then slices ka..kb-1).
        x_copy (optional [num_cols, d_in] view with unit inner stride): the projection kernel also stores X there
        (FSW_conv's concat buffer, reference fsw_conv.py:357-358).
        prepared: the result of prepare() (its slice_range is used); chunks: row chunks to process (default all).
        long_rows_only: write only the rows of more than REG_MAX_DEG neighbours (the other rows of `out` are not touched).
        This is the hot path: projection (MFMA) -> coefficient table -> fused neighbourhood kernels.
        """
        L = _lib.lib()
        dev = X.device
        has_mass = 1 if self.encode_total_mass else 0
        if x_copy is not None:
            assert x_copy.shape == X.shape and x_copy.stride(1) == 1 and x_copy.dtype == X.dtype
        if prepared is not None:
            assert slice_range is None or tuple(slice_range) == tuple(prepared["slice_range"])
            slice_range = prepared["slice_range"]
        ka, kb = (0, self.nSlices) if slice_range is None else slice_range
        assert 0 <= ka < kb <= self.nSlices or self.d_out == 0, 'bad slice_range'
        S = kb - ka
        partial = (ka, kb) != (0, self.nSlices)
        assert X.is_contiguous() and out.stride(1) == 1 and out.shape[0] == graph.num_rows and out.shape[1] >= has_mass + S
        stream = torch.cuda.current_stream(dev).cuda_stream
        if self.d_out == 0:
            return out
        if self.nSlices == 0:
            raise NotImplementedError("fsw_gnn_amd: nSlices == 0 with encode_total_mass is not supported")
        method = self.total_mass_encoding_method
        plain = (not self.encode_total_mass) or method == 'plain' or self._force_plain
        if partial and not plain:
            raise NotImplementedError("slice sharding supports total_mass_encoding_method='plain' only")
        bias = self.bias.detach() if (self.enable_bias and plain and not self._force_plain) else None
        if bias is not None and partial:
            bias = torch.cat([bias[:has_mass], bias[has_mass + ka:has_mass + kb]])
        unit_fast = graph.w is None and self.total_mass_pad_thresh <= 1.0

        step = S if (serialize_num_slices is None or serialize_num_slices >= S) else int(serialize_num_slices)
        assert step >= 1, 'serialize_num_slices must be None or a positive integer'
        if prepared is not None:
            assert step == S, 'a prepared projection covers its slices in one chunk'
            ldp, Xp, table, st = prepared["ldp"], prepared["Xp"], prepared["table"], prepared["stats"]
        else:
            ldp = _round_up(step, 64)
            Xp = torch.empty((X.shape[0], ldp), dtype=torch.float32, device=dev)
            table = None
            if unit_fast:
                table = torch.empty((int(L.fsw_unit_table_rows(_lib.REG_MAX_DEG)), ldp), dtype=torch.float32, device=dev)
            st = None
        V = self.projVecs.detach()[ka:kb]
        freqs = self.freqs.detach()[ka:kb]
        scratch = None
        assert (graph.ef is None) == (self.d_edge == 0), 'edge features must be given exactly when d_edge > 0'
        if chunks is None:
            chunks = [None] if graph.num_chunks == 1 and not graph.chunk_rows else range(graph.num_chunks)
        for k0 in range(0, S, step):
            k1 = min(S, k0 + step)
            Sc = k1 - k0
            fc = freqs[k0:k1]
            if prepared is None:
                Vc = V[k0:k1]
                rc = L.fsw_project_f32(_lib.ptr(X), X.shape[0], self.d_in, X.stride(0), _lib.ptr(Vc), Sc, Vc.stride(0),
                                       _lib.ptr(Xp), ldp, _lib.ptr(x_copy) if k0 == 0 else None,
                                       x_copy.stride(0) if x_copy is not None else 0,
                                       _lib.ptr(graph.stats_dev) if k0 == 0 else None, stream)
                _lib.check(rc, "fsw_project_f32")
                if unit_fast:
                    rc = L.fsw_unit_coeff_table(_lib.ptr(fc), Sc, _lib.REG_MAX_DEG, _lib.ptr(table), ldp, stream)
                    _lib.check(rc, "fsw_unit_coeff_table")
            if st is None:
                st = self._checked_stats(graph)   # one device->host read per forward (flags + degree classes)
            if scratch is None and st[_lib.STAT_NUM_GLOBAL] > 0:
                scratch = torch.empty(int(L.fsw_embed_scratch_bytes(st[_lib.STAT_MAX_DEGREE])), dtype=torch.uint8, device=dev)
            first = (k0 == 0)
            hm = has_mass if first else 0          # the first chunk also writes the total-mass column
            col0 = 0 if first else has_mass + k0   # first destination column of this chunk
            for c in chunks:
                a = self.make_args(graph, st, Xp, ldp, fc, Sc, table, out.data_ptr() + 4 * col0, out.stride(0),
                                   (bias.data_ptr() + 4 * col0) if bias is not None else None, out_scale, hm, scratch,
                                   slice_offset=ka + k0, chunk=c)
                if long_rows_only:      # rows above REG_MAX_DEG only (FSW_conv: the fused kernel did the others)
                    a.num_reg_rows, a.num_zero_rows = 0, 0
                rc = L.fsw_embed_f32(ctypes.byref(a), stream)
                _lib.check(rc, "fsw_embed_f32")

        if not plain:
            # 'homog' / 'homog_alt' (reference fsw_embedding.py:874-882, 1136-1144): rarely used epilogues, done with
            # three elementwise torch ops on the kernel's 'plain' output (the kernel wrote f(m)*scale in column 0)
            emb = out[:, 1:self.d_out]
            tm = out[:, 0:1] / out_scale
            norm = emb.abs().mean(dim=-1, keepdim=True) / out_scale
            if method == 'homog':
                out[:, 0:1] = out_scale * tm * norm
            else:
                out[:, 0:1] = out_scale * torch.where(tm <= 1, tm * (2 - tm), torch.ones_like(tm)) * norm
                emb.mul_(torch.where(tm <= 1, tm.square(), 2 * tm - 1))
            if self.enable_bias:
                out[:, :self.d_out] += out_scale * self.bias.detach()
        return out


# ----------------------------------------------------------------------------------------------------
# segmented cumulative sum: public signature of the reference (fsw_embedding.py:2795)
# ----------------------------------------------------------------------------------------------------
def segcumsum(values, segment_ids, max_seg_size=None, in_place=False, thorough_verify_input=False,
              always_use_pure_torch=False, reverse=False):
    """Inclusive scan of `values` restarted wherever consecutive `segment_ids` differ (HIP, stream-ordered).

    Keeps the reference signature; max_seg_size is accepted and unused (the scan is single-level), and
    always_use_pure_torch=True is rejected because this build has no PyTorch path.
    """
    assert values.dim() == 1, 'values must be a 1-dimensional tensor'
    assert segment_ids.dim() == 1, 'segment_ids must be a 1-dimensional tensor'
    assert segment_ids.numel() == values.numel(), 'values and segment_ids must contain the same number of elements'
    assert segment_ids.dtype in (torch.int32, torch.int64), 'segment_ids must have int32 or int64 dtype'
    assert values.device == segment_ids.device, 'values and segment_ids must be on the same device'
    assert not segment_ids.is_sparse, 'segment_ids cannot be sparse'
    assert segment_ids.is_contiguous(), 'segment_ids must be in contiguous format'
    assert not values.is_sparse, 'values cannot be sparse'
    assert (not in_place) or values.is_contiguous(), 'when in_place==True, values must be in contiguous format'
    if max_seg_size is not None:
        assert isinstance(max_seg_size, numbers.Number)
        assert max_seg_size >= 1
    if always_use_pure_torch:
        raise RuntimeError("fsw_gnn_amd has no pure-PyTorch path: always_use_pure_torch=True is not supported")
    if values.device.type != 'cuda':
        raise RuntimeError("fsw_gnn_amd.segcumsum: tensors must live on a HIP device (no CPU path)")
    if values.dtype == torch.float32:
        dtype_num = 0
    elif values.dtype == torch.float64:
        dtype_num = 1
    else:
        raise RuntimeError("Unsupported input_tensor dtype ''%s''" % (str(values.dtype)))
    if thorough_verify_input:
        _, cc = torch.unique_consecutive(segment_ids, return_counts=True)
        _, ct = torch.unique(segment_ids, return_counts=True)
        assert cc.numel() == ct.numel(), 'repeated segment IDs detected'
        assert not torch.isinf(values).any(), "Found infs in ''values''"
        assert not torch.isnan(values).any(), "Found nans in ''values''"
    L = _lib.lib()
    src = values.contiguous()
    out = src if in_place else torch.empty_like(src)
    n = src.numel()
    if n == 0:
        return out
    ws_bytes = int(L.fsw_segcumsum_workspace_bytes(n))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=values.device)
    stream = torch.cuda.current_stream(values.device).cuda_stream
    rc = L.fsw_segcumsum(dtype_num, _lib.ptr(src), _lib.ptr(out), _lib.ptr(segment_ids), segment_ids.element_size(), n,
                         1 if reverse else 0, _lib.ptr(ws), ws_bytes, stream)
    _lib.check(rc, "fsw_segcumsum")
    return out
