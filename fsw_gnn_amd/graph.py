"""Adjacency handling: edge list -> CSR by recipient + degree bins, on the GPU (csrc/graph_build.hip).

Stands where the reference builds a coalesced torch.sparse_coo adjacency and repeatedly sorts its int64
COO keys (FSW_conv.edge_index_to_adj, reference fsw_conv.py:384-447; sp.get_slice_info, reference
fsw_embedding.py:2586-2678).
"""
import torch

from . import _lib


class CSRGraph:
    """Device-resident CSR adjacency adj[recipient, sender] with rows bucketed by in-degree.

    rowptr int32[num_rows+1], col int32[nnz], w float32[nnz] or None (unit weights),
    perm int32[num_rows] (rows ordered by degree bin), bin_start int32[NUM_BINS+1],
    stats_dev int32[NUM_STATS] (device) -- read back once per forward by `stats()`.
    """

    def __init__(self, num_rows, num_cols, num_edges, rowptr, col, w, perm, bin_start, stats_dev, invperm=None, ef=None,
                 slot_of_edge=None, chunk_rows=0, meta=None, hint=None):
        self.num_rows, self.num_cols, self.num_edges = num_rows, num_cols, num_edges
        # chunk_rows > 0: bin_start is [num_chunks, NUM_BINS + 1], rows binned per chunk of chunk_rows consecutive rows
        self.chunk_rows = chunk_rows
        self.num_chunks = -(-num_rows // chunk_rows) if chunk_rows else 1
        self.rowptr, self.col, self.w, self.perm, self.bin_start = rowptr, col, w, perm, bin_start
        self.stats_dev = stats_dev
        self.invperm = invperm
        self.ef = ef                        # [num_edges, d_edge] summed edge features of the coalesced entries, or None
        self.slot_of_edge = slot_of_edge    # int32[num_input_edges]: CSR entry of every input edge (coalesced build)
        self._stats = None
        self._sender_major = None
        self._meta = meta                   # int32 [NUM_STATS + num_chunks * (NUM_BINS + 1)]: stats_dev | bin_start, or None
        self.bin_start_host = None          # numpy int32 [num_chunks, NUM_BINS + 1] after read_stats()
        self._hint = hint                   # BuildHint of the layer that built the graph (told the maximal degree when it is read)

    def sender_major(self):
        """(cptr int32[num_cols + 1], order int32[nnz]): the CSR entries listed sender by sender (fsw_graph_transpose), built on
        first use and kept with the graph; the store-and-sum backward sums the stored key gradients over it."""
        if self._sender_major is None:
            L = _lib.lib()
            nnz = self.stats()[_lib.STAT_NNZ]
            dev = self.rowptr.device
            cptr = torch.empty(self.num_cols + 1, dtype=torch.int32, device=dev)
            order = torch.empty(max(nnz, 1), dtype=torch.int32, device=dev)
            ws_bytes = int(L.fsw_graph_workspace_bytes(self.num_cols, nnz))
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            stream = torch.cuda.current_stream(dev).cuda_stream
            _lib.check(L.fsw_graph_transpose(_lib.ptr(self.col), nnz, self.num_cols, _lib.ptr(cptr), _lib.ptr(order), _lib.ptr(ws), ws_bytes,
                                             stream), "fsw_graph_transpose")
            self._sender_major = (cptr, order)
        return self._sender_major

    def read_stats(self):
        """Fresh host copy of the stats words (one small device->host copy; synchronises the stream)."""
        if self._meta is not None:
            # stats and the bin table live in one buffer: ONE device->host copy brings both (the library sizes its grids by
            # the exact rows of every degree bin when it gets the host table: fsw_embed_args.bin_start_host)
            host = self._meta.cpu().numpy()
            self._stats = host[:_lib.NUM_STATS].tolist()
            self.bin_start_host = host[_lib.NUM_STATS:].reshape(self.num_chunks, _lib.NUM_BINS + 1)
        else:
            self._stats = self.stats_dev.cpu().tolist()
        if self._hint is not None:
            self._hint.note(self.num_rows, self.num_edges, self._stats[_lib.STAT_MAX_DEGREE])
        return self._stats

    def stats(self):
        """Host copy of the stats words as of the last read_stats() (read now if there was none)."""
        return self._stats if self._stats is not None else self.read_stats()

    def clear_input_flags(self):
        """A graph reused for another forward (FSW_conv.cache_graph): forget what earlier inputs set (X non-finite)."""
        self.stats_dev[_lib.STAT_FLAGS:_lib.STAT_FLAGS + 1].bitwise_and_(~_lib.FLAG_X_NONFINITE)
        self._stats = None

    @property
    def flags(self):
        return self.stats()[_lib.STAT_FLAGS]

    @property
    def max_degree(self):
        return self.stats()[_lib.STAT_MAX_DEGREE]

    def in_degrees(self):
        """Number of incoming edges per row (with multiplicity), float32 -- fsw_conv.py:400-401."""
        return (self.rowptr[1:] - self.rowptr[:-1]).to(torch.float32)


def round_chunk_rows(rows, multiple_of=1):
    """Smallest legal chunk_rows >= rows: a multiple of BIN_BLOCK_ROWS (graph_build.hip) and of `multiple_of`."""
    import math
    m = math.lcm(_lib.BIN_BLOCK_ROWS, int(multiple_of))
    return max(1, -(-int(rows) // m)) * m


TWO_LEVEL_MAX_DEGREE = 2048


class BuildHint:
    """What ONE owner (an FSW_conv / FSW_embedding instance) has learnt about the graphs it is fed, to pick between the two CSR
    builds -- which give the same CSR, entry for entry, so the choice only moves time.  The two-level build (one workgroup per
    bucket of 2048 rows, csrc/graph_build.hip) is the faster one on graphs without hub rows (0.25 against 0.31 ms at BASELINE
    config 3) but serialises on a hub-heavy bucket (6.4 ms once on the 64M-edge RMAT graph); the degrees are only known after
    the build.  So: the LSD build until this owner has seen a graph without rows above TWO_LEVEL_MAX_DEGREE neighbours, the
    two-level build while the last graph it saw -- and the last one of this shape -- had none.  No process-wide state: two layers
    fed different datasets do not influence each other, and a graph of unknown character never takes the risky build."""

    def __init__(self):
        self.last_skewed = None             # None: nothing seen yet
        self.skewed_shapes = set()

    def note(self, num_rows, num_edges, max_degree):
        self.last_skewed = max_degree > TWO_LEVEL_MAX_DEGREE
        if self.last_skewed:
            self.skewed_shapes.add((num_rows, num_edges))
        else:
            self.skewed_shapes.discard((num_rows, num_edges))

    def two_level(self, num_rows, num_edges):
        return self.last_skewed is False and (num_rows, num_edges) not in self.skewed_shapes


def build_csr(recipients, senders, edge_w, num_rows, num_cols, want_invperm=False, chunk_rows=0, algo="auto", hint=None):
    """recipients/senders: int64 CUDA tensors [E]; edge_w: float32 CUDA tensor [E] or None (unit weights).
    chunk_rows > 0: degree bins per chunk of chunk_rows consecutive rows (include/fsw_hip.h, fsw_graph_build).
    algo: 'lsd' (fsw_graph_build), 'two_level' (fsw_graph_build_two_level), or 'auto' = what `hint` (the caller's BuildHint)
    allows: two_level once the caller has seen graphs without hub rows, the LSD build otherwise and without a hint."""
    L = _lib.lib()
    dev = recipients.device
    if dev.type != "cuda":
        raise RuntimeError("fsw_gnn_amd.build_csr: tensors must live on a HIP device (no CPU path)")
    assert recipients.dtype == torch.int64 and senders.dtype == torch.int64, "edge endpoints must be int64"
    assert recipients.dim() == 1 and recipients.shape == senders.shape
    recipients = recipients.contiguous()
    senders = senders.contiguous()
    E = recipients.numel()
    if edge_w is not None:
        assert edge_w.dtype == torch.float32 and edge_w.shape == recipients.shape and edge_w.device == dev
        edge_w = edge_w.contiguous()
    rowptr = torch.empty(num_rows + 1, dtype=torch.int32, device=dev)
    col = torch.empty(max(E, 1), dtype=torch.int32, device=dev)
    w = torch.empty(max(E, 1), dtype=torch.float32, device=dev) if edge_w is not None else None
    perm = torch.empty(num_rows, dtype=torch.int32, device=dev)
    invperm = torch.empty(num_rows, dtype=torch.int32, device=dev) if want_invperm else None
    num_chunks = -(-num_rows // chunk_rows) if chunk_rows else 1
    assert num_chunks <= _lib.MAX_ROW_CHUNKS and chunk_rows % _lib.BIN_BLOCK_ROWS == 0, "bad chunk_rows"
    meta = torch.empty(_lib.NUM_STATS + num_chunks * (_lib.NUM_BINS + 1), dtype=torch.int32, device=dev)
    stats, bin_start = meta[:_lib.NUM_STATS], meta[_lib.NUM_STATS:].view(num_chunks, _lib.NUM_BINS + 1)
    ws_bytes = L.fsw_graph_workspace_bytes(num_rows, E)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    assert algo in ("auto", "lsd", "two_level")
    two_level = algo == "two_level" or (algo == "auto" and hint is not None and hint.two_level(num_rows, E))
    fn = L.fsw_graph_build_two_level if two_level else L.fsw_graph_build
    rc = fn(_lib.ptr(recipients), _lib.ptr(senders), _lib.ptr(edge_w), E, num_rows, num_cols, chunk_rows,
            _lib.ptr(rowptr), _lib.ptr(col), _lib.ptr(w), _lib.ptr(perm), _lib.ptr(invperm), _lib.ptr(bin_start),
            _lib.ptr(stats), _lib.ptr(ws), ws_bytes, stream)
    _lib.check(rc, "fsw_graph_build")
    return CSRGraph(num_rows, num_cols, E, rowptr, col, w, perm, bin_start if chunk_rows else bin_start.view(-1), stats, invperm,
                    chunk_rows=chunk_rows, meta=meta, hint=hint)


def build_csr_coalesced(recipients, senders, edge_w, edge_feat, num_rows, num_cols, want_slots=False):
    """Coalescing build: entries sorted by (recipient, sender), parallel edges merged (weights and edge features summed),
    like torch.sparse_coo_tensor(...).coalesce() in the reference (fsw_conv.py:397-398, 436-437).  Used with edge features.

    edge_w float32 [E] or None (1 per edge); edge_feat float32 [E, d_edge] or None.  The CSR arrays are sized for E
    entries, the first graph.stats()[STAT_NNZ] are valid."""
    L = _lib.lib()
    dev = recipients.device
    if dev.type != "cuda":
        raise RuntimeError("fsw_gnn_amd.build_csr_coalesced: tensors must live on a HIP device (no CPU path)")
    assert recipients.dtype == torch.int64 and senders.dtype == torch.int64 and recipients.shape == senders.shape
    recipients, senders = recipients.contiguous(), senders.contiguous()
    E = recipients.numel()
    d_edge = 0
    if edge_feat is not None:
        assert edge_feat.dtype == torch.float32 and edge_feat.dim() == 2 and edge_feat.shape[0] == E
        edge_feat = edge_feat.contiguous()
        d_edge = edge_feat.shape[1]
    if edge_w is not None:
        assert edge_w.dtype == torch.float32 and edge_w.shape == recipients.shape
        edge_w = edge_w.contiguous()
    rowptr = torch.empty(num_rows + 1, dtype=torch.int32, device=dev)
    col = torch.empty(max(E, 1), dtype=torch.int32, device=dev)
    w = torch.empty(max(E, 1), dtype=torch.float32, device=dev)
    ef = torch.empty((max(E, 1), d_edge), dtype=torch.float32, device=dev) if d_edge else None
    slot = torch.empty(max(E, 1), dtype=torch.int32, device=dev) if want_slots else None
    perm = torch.empty(num_rows, dtype=torch.int32, device=dev)
    meta = torch.empty(_lib.NUM_STATS + _lib.NUM_BINS + 1, dtype=torch.int32, device=dev)
    stats, bin_start = meta[:_lib.NUM_STATS], meta[_lib.NUM_STATS:]
    ws_bytes = L.fsw_graph_workspace_bytes(num_rows, E)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    rc = L.fsw_graph_build_coalesced(_lib.ptr(recipients), _lib.ptr(senders), _lib.ptr(edge_w), _lib.ptr(edge_feat), d_edge, E,
                                     num_rows, num_cols, _lib.ptr(rowptr), _lib.ptr(col), _lib.ptr(w), _lib.ptr(ef), _lib.ptr(slot),
                                     _lib.ptr(perm), None, _lib.ptr(bin_start), _lib.ptr(stats), _lib.ptr(ws), ws_bytes, stream)
    _lib.check(rc, "fsw_graph_build_coalesced")
    return CSRGraph(num_rows, num_cols, E, rowptr, col, w, perm, bin_start, stats, None, ef, slot, meta=meta)
