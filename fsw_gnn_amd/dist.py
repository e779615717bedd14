"""Multi-GPU: shard the SLICE axis over the ranks (RCCL over xGMI; gloo in the CPU tests and rehearsals).

The reference has no distributed code at all; BASELINE north_star / SURVEY.md 8(e) prescribe this
partition.  Every (recipient, slice) pair is independent and slice k is paired with frequency k, so rank r of G owns a
contiguous block of slices -- rows [ka, kb) of projVecs and entries [ka, kb) of freqs.  X, the CSR adjacency and the
total-mass column are replicated (each rank builds the CSR itself).  Three ways to put the layer back together:

  gather    (the contracted form)  every rank embeds its block, ONE all-gather per node-range chunk collects
            [G, rows, 1 + width] and a strided copy lays the blocks side by side: every rank ends with the full
            embedding, bit-identical to one GPU (no reduction anywhere).  The chunks pipeline: the collective of chunk
            c runs on the communicator's stream while the kernels of chunk c + 1 run on the compute stream.
  consumer  (FSW_conv whose first MLP module is a Linear layer, unit weights, rows <= 32 neighbours)  the layer
            output is  act(E . W1^T + x . W2^T + b)  and  E . W1^T = sum_r E_r . W1_r^T  over the slice blocks, so rank
            r multiplies ITS block by ITS columns of W1 inside the fused neighbourhood kernel (csrc/conv_fused.hip, the
            embedding never reaches HBM) and the ranks reduce-scatter the n x H partial sums instead of gathering the
            n x S embedding; the owner of a row block adds x . W2^T + b, applies the activation and one all-gather
            of the H-wide rows rebuilds the output.  No replicated GEMM, half the bytes of `gather` when H = S / 2, and
            output='sharded' stops after the reduce-scatter (the natural hand-over to a row-sharded next layer).
  exchange  (any FSW_conv configuration the slice shard supports, any degree)  every rank embeds its slice block for ALL rows and
            ONE all-to-all hands each row block to its owner: rank j receives, for the rows it owns, the slice blocks of all
            ranks -- (G-1)/G of n x S/G floats per rank instead of the n x S of `gather` or the n x H partial sums of `consumer`
            (S/G < H from 4 ranks on at BASELINE config 3: 112 MB against 448 MB at 8 ranks).  The owner lays the blocks side by
            side (bit-identical to the single-GPU embedding rows), runs the whole tail (concat, MLP) on its rows, and one
            all-gather of the out_channels-wide rows rebuilds the output; output='sharded' stops before it.
  training  `gather` through torch.autograd: the slice blocks are differentiated by this rank's backward kernels,
            block gradients are summed over the ranks inside backward (fsw_embedding._EmbedGraphFn), the replicated
            tail (MLP) computes identical gradients on every rank.

Node-range chunks come from the CSR build (graph.py: chunk_rows), which bins the rows by degree separately inside
every chunk, so that a kernel launch can be restricted to one chunk's recipients.
"""
import ctypes

import torch
import torch.distributed as dist

from . import _lib
from .graph import round_chunk_rows


def slice_partition(num_slices, world_size):
    """Contiguous, balanced blocks: returns [(ka, kb)] * world_size (first `rem` ranks get one extra slice)."""
    base, rem = divmod(num_slices, world_size)
    out, a = [], 0
    for r in range(world_size):
        b = a + base + (1 if r < rem else 0)
        out.append((a, b))
        a = b
    return out


def node_block(num_rows, world_size, rank):
    """Recipient-row sharding (FSW_conv.enable_node_parallel): rank r owns rows [r0, r0 + nl) of equal-size blocks of
    per = ceil(num_rows / world_size) rows (the last blocks may be short or empty).  Returns (per, r0, nl)."""
    per = -(-num_rows // world_size)
    r0 = min(rank * per, num_rows)
    return per, r0, min(r0 + per, num_rows) - r0


def chunk_plan(num_rows, world_size, num_chunks):
    """(chunk_rows, num_chunks) for pipelining a forward over node ranges: chunk_rows is a legal CSR chunk size
    (multiple of the binning granularity) divisible by world_size (every chunk reduce-scatters into equal blocks)."""
    num_chunks = max(1, int(num_chunks))
    cs = round_chunk_rows(-(-num_rows // num_chunks), multiple_of=world_size)
    return cs, -(-num_rows // cs)


def default_chunks(num_rows):
    """Pipeline depth: 4 node-range chunks for graphs large enough that a chunk still fills the chip."""
    return 4 if num_rows >= 400_000 else (2 if num_rows >= 100_000 else 1)


# ---------------------------------------------------------------------------------------------------------------------
# collectives (thin wrappers: every backend the tests use must take the same path shape)
# ---------------------------------------------------------------------------------------------------------------------
COLLECTIVES_ENABLED = True   # bench.py clears it to time a rank's compute alone: every collective becomes a local copy


class _Done:
    def wait(self):
        return True


def _reduce_scatter(out, inp, group, async_op):
    """out [m, H] <- sum over ranks of inp[rank * m:(rank + 1) * m].  RCCL and gloo-on-CPU reduce-scatter natively; gloo
    on device tensors (rehearsals with several ranks on one GPU) has no reduce_scatter: all_reduce + slice there."""
    if not COLLECTIVES_ENABLED:
        m, r = out.shape[0], dist.get_rank(group)
        out.copy_(inp[r * m:(r + 1) * m])
        return _Done()
    if inp.is_cuda and dist.get_backend(group) == "gloo":
        tmp = inp.clone()
        dist.all_reduce(tmp, group=group)
        m = out.shape[0]
        r = dist.get_rank(group)
        out.copy_(tmp[r * m:(r + 1) * m])
        return _Done()
    w = dist.reduce_scatter_tensor(out, inp, group=group, async_op=async_op)
    return w if async_op else _Done()


def _all_gather(out, inp, group, async_op):
    if not COLLECTIVES_ENABLED:
        m, r = inp.shape[0], dist.get_rank(group)
        out[r * m:(r + 1) * m].copy_(inp)
        return _Done()
    w = dist.all_gather_into_tensor(out, inp, group=group, async_op=async_op)
    return w if async_op else _Done()


def _all_to_all(out, inp, group, async_op):
    """out[j] <- block `rank` of rank j's inp, for every j (inp, out: [G, ...] contiguous, equal blocks)."""
    if not COLLECTIVES_ENABLED:
        out.copy_(inp)                            # the same bytes, moved locally
        return _Done()
    if inp.is_cuda and dist.get_backend(group) == "gloo":       # rehearsals with several gloo ranks on one GPU
        world, r = dist.get_world_size(group), dist.get_rank(group)
        big = torch.empty((world,) + tuple(inp.shape), dtype=inp.dtype, device=inp.device)
        dist.all_gather_into_tensor(big.view(world * inp.shape[0], *inp.shape[1:]), inp, group=group)
        out.copy_(big[:, r])
        return _Done()
    w = dist.all_to_all_single(out, inp, group=group, async_op=async_op)
    return w if async_op else _Done()


# ---------------------------------------------------------------------------------------------------------------------
# gather: all-gather of the embedding blocks
# ---------------------------------------------------------------------------------------------------------------------
def interleave_blocks(gathered, parts, has_mass, out):
    """gathered [G, rows, has_mass + wmax] -> out[:, :has_mass + S] (rows of `out` = rows of the blocks)."""
    if has_mass:
        out[:, 0] = gathered[0, :, 0]            # every rank computed the same total-mass column
    for r, (a, b) in enumerate(parts):
        if b > a:
            out[:, has_mass + a:has_mass + b] = gathered[r, :, has_mass:has_mass + (b - a)]
    return out


def all_gather_slice_blocks(local, parts, has_mass, out, group=None):
    """local [n, has_mass + max_width] of this rank -> out[:, :has_mass + S] on every rank (one collective).

    `parts` is slice_partition(S, world).  Blocks are padded to the widest block so that one
    all_gather_into_tensor (a single collective, equal message sizes) suffices.
    """
    world = dist.get_world_size(group)
    n = local.shape[0]
    wmax = max(b - a for a, b in parts)
    assert local.shape[1] == has_mass + wmax and local.is_contiguous()
    flat = torch.empty((world * n, has_mass + wmax), dtype=local.dtype, device=local.device)
    _all_gather(flat, local, group, async_op=False)            # concatenation along dim 0 (valid for RCCL and gloo)
    return interleave_blocks(flat.view(world, n, has_mass + wmax), parts, has_mass, out)


def pipelined_gather(compute_chunk, num_chunks, chunk_rows, num_rows, parts, has_mass, out, group=None, stats=None):
    """The chunk pipeline of the `gather` form.  compute_chunk(c, local_c) fills local_c [chunk_rows, has_mass + wmax]
    (rows of chunk c; rows past num_rows may stay unwritten) on the current stream; the all-gather of chunk c is issued
    asynchronously right behind it, the interleave copies follow once all kernels are queued."""
    world = dist.get_world_size(group)
    wmax = max(b - a for a, b in parts)
    w = has_mass + wmax
    local = torch.empty((num_chunks * chunk_rows, w), dtype=out.dtype, device=out.device)
    flat = torch.empty((num_chunks, world * chunk_rows, w), dtype=out.dtype, device=out.device)
    if num_chunks * chunk_rows > num_rows:
        local[num_rows:].zero_()
    works = []
    for c in range(num_chunks):
        lc = local[c * chunk_rows:(c + 1) * chunk_rows]
        compute_chunk(c, lc)
        works.append(_all_gather(flat[c], lc, group, async_op=num_chunks > 1))
    for c in range(num_chunks):
        works[c].wait()
        r0, r1 = c * chunk_rows, min((c + 1) * chunk_rows, num_rows)
        interleave_blocks(flat[c].view(world, chunk_rows, w)[:, :r1 - r0], parts, has_mass, out[r0:r1])
    if stats is not None:
        stats["collective"] = "all_gather"
        stats["bytes_sent_per_rank"] = local.numel() * local.element_size() * (world - 1)
        stats["bytes_received_per_rank"] = stats["bytes_sent_per_rank"]
    return out


def sharded_embed_into(emb_mod, X, graph, out, out_scale=1.0, group=None, x_copy=None, prepared=None, stats=None):
    """Slice-sharded version of FSW_embedding.embed_into: every rank ends with the full embedding in `out`."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1:
        return emb_mod.embed_into(X, graph, out, out_scale=out_scale, x_copy=x_copy, prepared=prepared)
    has_mass = 1 if emb_mod.encode_total_mass else 0
    parts = slice_partition(emb_mod.nSlices, world)
    ka, kb = parts[rank]
    if prepared is None:
        prepared = emb_mod.prepare(X, graph, x_copy=x_copy, slice_range=(ka, kb))
    elif x_copy is not None:
        x_copy.copy_(X)
    cs = graph.chunk_rows if graph.chunk_rows else graph.num_rows
    nchunks = graph.num_chunks

    def compute(c, lc):
        if kb > ka:   # columns past this rank's block (S not divisible by the world size) are never read back
            emb_mod.embed_into(X, graph, _RowWindow(lc, c * cs, graph.num_rows), out_scale=out_scale, prepared=prepared,
                               chunks=[c if graph.chunk_rows else None])

    return pipelined_gather(compute, nchunks, cs, graph.num_rows, parts, has_mass, out, group, stats)


class _RowWindow:
    """What embed_into needs of an output tensor, for a buffer that holds rows [row0, row0 + buf.shape[0]) of the full
    output: kernels address out[node * ldo], so the base pointer is shifted back by row0 rows (only rows of the window
    are ever written: the launch is restricted to the chunk's recipients)."""

    def __init__(self, buf, row0, num_rows):
        self.buf, self.row0 = buf, row0
        self.shape = (num_rows, buf.shape[1])
        self.dtype, self.device = buf.dtype, buf.device

    def stride(self, d):
        return self.buf.stride(d)

    def data_ptr(self):
        return self.buf.data_ptr() - self.row0 * self.buf.stride(0) * self.buf.element_size()


class _AllGatherSlices(torch.autograd.Function):
    """Differentiable all_gather_slice_blocks: backward hands every rank the columns of its own block (the tail after the
    embedding is replicated, so every rank holds the same full gradient; nothing is communicated here -- the sum of the
    block gradients over the ranks happens inside _EmbedGraphFn.backward)."""

    @staticmethod
    def forward(ctx, local, parts, has_mass, group):
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        wmax = max(b - a for a, b in parts)
        n = local.shape[0]
        S = parts[-1][1]
        padded = local
        if local.shape[1] != has_mass + wmax:
            padded = torch.zeros((n, has_mass + wmax), dtype=local.dtype, device=local.device)
            padded[:, :local.shape[1]] = local
        out = torch.empty((n, has_mass + S), dtype=local.dtype, device=local.device)
        all_gather_slice_blocks(padded.contiguous(), parts, has_mass, out, group)
        ctx.parts, ctx.has_mass, ctx.rank = parts, has_mass, rank
        return out

    @staticmethod
    def backward(ctx, g):
        ka, kb = ctx.parts[ctx.rank]
        hm = ctx.has_mass
        return torch.cat([g[:, :hm], g[:, hm + ka:hm + kb]], dim=1).contiguous(), None, None, None


def sharded_embed_autograd(emb_mod, X, graph, out_scale=1.0, group=None, edge_feat=None):
    """Training form: full embedding [n, d_out] on every rank with gradients flowing to X / projVecs / freqs / bias /
    the total-mass scale exactly as on one GPU (block gradients all-reduced inside backward)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if world == 1:
        return emb_mod.embed_autograd(X, graph, out_scale=out_scale, edge_feat=edge_feat)
    if emb_mod.nSlices < world:
        raise NotImplementedError("fsw_gnn_amd: slice-parallel training needs at least one slice per rank")
    has_mass = 1 if emb_mod.encode_total_mass else 0
    parts = slice_partition(emb_mod.nSlices, world)
    local = emb_mod.embed_autograd(X, graph, out_scale=out_scale, edge_feat=edge_feat, slice_range=parts[rank], group=group,
                                   reduce_grads=True)
    return _AllGatherSlices.apply(local, parts, has_mass, group)


# ---------------------------------------------------------------------------------------------------------------------
# consumer: the first Linear layer stays sharded, partial sums are reduce-scattered
# ---------------------------------------------------------------------------------------------------------------------
def reduce_scatter_pipeline(compute_partial, finish_rows, num_chunks, chunk_rows, num_rows, H, dtype, device, group=None,
                            output="replicated", stats=None):
    """Chunk pipeline of the `consumer` form (pure choreography: the kernels come in as callables, so the CPU tests drive
    it with the oracle).

    compute_partial(c, P_c)   fill P_c [chunk_rows, H] with this rank's partial sums for the rows of chunk c
    finish_rows(r0, r1, R)    R [r1 - r0, H] holds the summed partials of rows r0..r1-1 (all < num_rows): finish them in
                              place (+ x . W2^T + b, activation)
    Chunk c is reduce-scattered in equal blocks: rank r owns rows [c * cs + r * cs / G, c * cs + (r + 1) * cs / G).
    output = 'replicated': one all-gather per chunk rebuilds Y [num_rows, H] on every rank (returned);
    output = 'sharded'   : returns (R [num_chunks, cs / G, H], row0 [num_chunks]) -- this rank's finished rows.
    """
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    assert chunk_rows % world == 0
    m = chunk_rows // world
    P = torch.empty((num_chunks * chunk_rows, H), dtype=dtype, device=device)
    if num_chunks * chunk_rows > num_rows:
        P[num_rows:].zero_()
    R = torch.empty((num_chunks, m, H), dtype=dtype, device=device)
    works = []
    for c in range(num_chunks):
        Pc = P[c * chunk_rows:(c + 1) * chunk_rows]
        compute_partial(c, Pc)
        works.append(_reduce_scatter(R[c], Pc, group, async_op=num_chunks > 1))
    Y = torch.empty((num_chunks * chunk_rows, H), dtype=dtype, device=device) if output == "replicated" else None
    works2 = []
    for c in range(num_chunks):
        works[c].wait()
        r0 = c * chunk_rows + rank * m
        r1 = min(r0 + m, num_rows)
        if r1 > r0:
            finish_rows(r0, r1, R[c, :r1 - r0])
        if Y is not None:
            works2.append(_all_gather(Y[c * chunk_rows:(c + 1) * chunk_rows], R[c], group, async_op=num_chunks > 1))
    for w in works2:
        w.wait()
    if stats is not None:
        esz = P.element_size()
        stats["collective"] = "reduce_scatter" + ("+all_gather" if Y is not None else "")
        rs = num_chunks * chunk_rows * H * esz * (world - 1) // world
        stats["bytes_sent_per_rank"] = rs * (2 if Y is not None else 1)
        stats["bytes_received_per_rank"] = stats["bytes_sent_per_rank"]
    if Y is not None:
        return Y[:num_rows]
    row0 = torch.tensor([c * chunk_rows + rank * m for c in range(num_chunks)], dtype=torch.int64)
    return R, row0


def consumer_weight(conv, group=None):
    """This rank's columns of the first Linear layer packed for the fused kernel (one launch; callers issue it BEFORE the forward's
    device->host stats read so that it does not sit in the gap behind it)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    emb = conv.fsw_embed
    has_mass = 1 if emb.encode_total_mass else 0
    ka, kb = slice_partition(emb.nSlices, world)[rank]
    K = (has_mass if rank == 0 else 0) + (kb - ka)
    if K == 0:
        return None
    return conv._fused_weight(col0=0 if rank == 0 else has_mass + ka, K=K, want_w2=False)[0]


def consumer_forward(conv, x, graph, prepared, scale, group=None, output="replicated", stats=None, wq=None):
    """FSW_conv's slice-sharded fused layer (see the module docstring).  Preconditions checked by the caller:
    conv._fusable(), unit-weight graph without rows above 32 neighbours, prepared = projection of this rank's block."""
    L = _lib.lib()
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    emb = conv.fsw_embed
    lin = conv.mlp[0]
    H, E = lin.out_features, conv.embed_dim
    has_mass = 1 if emb.encode_total_mass else 0
    ka, kb = prepared["slice_range"]
    hm = has_mass if rank == 0 else 0            # rank 0 also carries the total-mass column (column 0 of W1)
    K = hm + (kb - ka)
    n = graph.num_rows
    cs, nchunks = (graph.chunk_rows, graph.num_chunks) if graph.chunk_rows else (round_chunk_rows(n, world), 1)
    assert graph.chunk_rows or cs >= n
    act, slope, next_module = 0, 0.0, 1
    if len(conv.mlp) > 1 and isinstance(conv.mlp[1], torch.nn.LeakyReLU):
        act, slope, next_module = 2, float(conv.mlp[1].negative_slope), 2
    elif len(conv.mlp) > 1 and isinstance(conv.mlp[1], torch.nn.ReLU):
        act, next_module = 1, 2
    st = prepared["stats"]
    if K > 0:
        if wq is None:
            wq = consumer_weight(conv, group)
        bias = None
        if emb.enable_bias:
            b = emb.bias.detach()
            bias = torch.cat([b[:hm], b[has_mass + ka:has_mass + kb]]).contiguous()
        fr = emb.freqs.detach()[ka:kb]
    stream = torch.cuda.current_stream(x.device).cuda_stream
    W2 = lin.weight.detach()[:, E:] if conv.concat_self else None
    lb = lin.bias.detach() if lin.bias is not None else None

    def compute_partial(c, Pc):
        if K == 0:
            Pc.zero_()
            return
        if kb == ka:       # rank 0 of a layer with more ranks than slices: only the mass column -- not worth a kernel variant
            raise NotImplementedError("fsw_gnn_amd: slice sharding needs at least one slice on rank 0")
        a = emb.make_args(graph, st, prepared["Xp"], prepared["ldp"], fr, kb - ka, prepared["table"], None, 0,
                          bias.data_ptr() if bias is not None else None, scale, hm, slice_offset=ka,
                          chunk=c if graph.chunk_rows else None)
        # Y rows are addressed by node id: shift the base so that node c * cs lands on row 0 of Pc
        ybase = Pc.data_ptr() - c * cs * Pc.stride(0) * 4
        rc = L.fsw_conv_fused_f32(ctypes.byref(a), wq.data_ptr(), wq.shape[1], None, H, None, 0, 0, 0, 0.0, ybase, Pc.stride(0), stream)
        _lib.check(rc, "fsw_conv_fused_f32")

    # x . W2^T + b of the rows this rank will own after the reduce-scatter (row block `rank` of every chunk): ONE gather of
    # those rows of x and ONE GEMM, queued before the neighbourhood kernels -- nothing of it waits for a collective.  What is
    # left behind the reduce-scatter is a single launch per chunk: R = act(R + own) (fsw_add_bias_act_f32).
    m = cs // world
    own = None
    if W2 is not None:
        pieces = []
        for c in range(nchunks):
            r0 = min(c * cs + rank * m, n)
            r1 = min(r0 + m, n)
            pieces.append(x[r0:r1])
            if r1 - r0 < m:                                         # pad rows past the last node: never read back
                pieces.append(x.new_zeros((m - (r1 - r0), x.shape[1])))
        xo = torch.cat(pieces, dim=0) if len(pieces) > 1 else pieces[0]
        own = (torch.addmm(lb, xo, W2.t()) if lb is not None else xo @ W2.t()).view(nchunks, m, H)

    def finish_rows(r0, r1, R):
        c = r0 // cs
        yin = own[c] if own is not None else None
        rc = L.fsw_add_bias_act_f32(_lib.ptr(R), R.stride(0), _lib.ptr(yin), yin.stride(0) if yin is not None else 0,
                                    _lib.ptr(lb) if (own is None and lb is not None) else None, r1 - r0, H, act, slope, stream)
        _lib.check(rc, "fsw_add_bias_act_f32")

    res = reduce_scatter_pipeline(compute_partial, finish_rows, nchunks, cs, n, H, x.dtype, x.device, group, output, stats)
    return res, next_module


# ---------------------------------------------------------------------------------------------------------------------
# exchange: all-to-all of the slice blocks, the tail on the rows a rank owns, all-gather of the output rows
# ---------------------------------------------------------------------------------------------------------------------
def all_to_all_pipeline(compute_block, finish_rows, num_chunks, chunk_rows, num_rows, w, H, dtype, device, group=None,
                        output="replicated", stats=None):
    """Chunk pipeline of the `exchange` form (pure choreography: the kernels come in as callables, so the CPU tests drive it
    with the oracle).

    compute_block(c, L_c)          fill L_c [chunk_rows, w] with this rank's slice block for the rows of chunk c
    finish_rows(c, r0, r1, B, Y_c) B [G, m, w]: block j = rank j's slice block for the rows r0 .. r0 + m - 1 this rank owns in
                                   chunk c (rows r1 .. are padding); write the finished rows into Y_c [m, H]
    Chunk c is split in equal row blocks: rank r owns rows [c * cs + r * cs / G, c * cs + (r + 1) * cs / G).
    output = 'replicated': one all-gather per chunk rebuilds Y [num_rows, H] on every rank (returned);
    output = 'sharded'   : returns (R [num_chunks, cs / G, H], row0 [num_chunks]) -- this rank's finished rows.
    """
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    assert chunk_rows % world == 0
    m = chunk_rows // world
    local = torch.empty((num_chunks, world, m, w), dtype=dtype, device=device)
    recv = torch.empty((num_chunks, world, m, w), dtype=dtype, device=device)
    if num_chunks * chunk_rows > num_rows:
        local.view(-1, w)[num_rows:].zero_()
    works = []
    for c in range(num_chunks):
        compute_block(c, local[c].view(chunk_rows, w))
        works.append(_all_to_all(recv[c], local[c], group, async_op=num_chunks > 1))
    R = torch.zeros((num_chunks, m, H), dtype=dtype, device=device)
    Y = torch.empty((num_chunks * chunk_rows, H), dtype=dtype, device=device) if output == "replicated" else None
    works2 = []
    for c in range(num_chunks):
        works[c].wait()
        r0 = c * chunk_rows + rank * m
        r1 = min(r0 + m, num_rows)
        if r1 > r0:
            finish_rows(c, r0, r1, recv[c], R[c])
        if Y is not None:
            works2.append(_all_gather(Y[c * chunk_rows:(c + 1) * chunk_rows], R[c], group, async_op=num_chunks > 1))
    for wk in works2:
        wk.wait()
    if stats is not None:
        esz = local.element_size()
        a2a = num_chunks * chunk_rows * w * esz * (world - 1) // world
        ag = num_chunks * chunk_rows * H * esz * (world - 1) // world if Y is not None else 0
        stats["collective"] = "all_to_all" + ("+all_gather" if Y is not None else "")
        stats["bytes_sent_per_rank"] = a2a + ag
        stats["bytes_received_per_rank"] = a2a + ag
    if Y is not None:
        return Y[:num_rows]
    row0 = torch.tensor([c * chunk_rows + rank * m for c in range(num_chunks)], dtype=torch.int64)
    return R, row0


def exchange_forward(conv, x, graph, prepared, scale, group=None, output="replicated", stats=None):
    """FSW_conv's slice-sharded layer in the `exchange` form (module docstring): works for every configuration embed_into()
    covers under slice sharding (unit or general weights, rows of any degree, any tail whose modules act row by row)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    emb = conv.fsw_embed
    has_mass = 1 if emb.encode_total_mass else 0
    parts = slice_partition(emb.nSlices, world)
    ka, kb = parts[rank]
    wmax = max(b - a for a, b in parts)
    w = has_mass + wmax
    n = graph.num_rows
    cs, nchunks = (graph.chunk_rows, graph.num_chunks) if graph.chunk_rows else (round_chunk_rows(n, world), 1)
    assert graph.chunk_rows or cs >= n
    E = conv.embed_dim
    width = E + conv.in_channels if conv.concat_self else E
    H = conv.out_channels

    def compute_block(c, lc):
        if kb > ka:
            emb.embed_into(x, graph, _RowWindow(lc, c * cs, n), out_scale=scale, prepared=prepared,
                           chunks=[c if graph.chunk_rows else None])

    def finish_rows(c, r0, r1, blocks, Yc):
        rows = r1 - r0
        buf = torch.empty((rows, width), dtype=x.dtype, device=x.device)
        interleave_blocks(blocks[:, :rows], parts, has_mass, buf)          # the embedding rows, as on one GPU
        if conv.concat_self:
            buf[:, E:] = x[r0:r1]
        Yc[:rows] = conv._tail_buffer(buf)

    return all_to_all_pipeline(compute_block, finish_rows, nchunks, cs, n, w, H, x.dtype, x.device, group, output, stats)
