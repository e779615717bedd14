"""FSW_conv / FSW_readout: host-side mirror of the reference's PyG layers (reference fsw_conv.py:54-517).

Same constructor surface (including the `config` override dictionary, reference fsw_conv.py:185-205), same
parameter names (`fsw_embed.*`, `mlp.*`, `dim_reduct`, `bn_final.*`, `size_coeff`) and the same
forward(vertex_features, edge_index, edge_features=None) contract.  The adjacency is built as a CSR on the
GPU (graph.py) instead of a coalesced torch.sparse_coo tensor, the neighbourhood embedding is written by the
HIP kernels straight into the left columns of the concat buffer, and the MLP / BatchNorm tail stays stock
torch.nn as in the reference (fsw_conv.py:357-369).

When torch_geometric is importable the classes derive from MessagePassing and register themselves with
graphgym under 'fsw_conv' / 'fsw_readout' like the reference (fsw_conv.py:54, 451); otherwise they are plain
nn.Modules (the reference never calls propagate(): message/aggregate/update are empty stubs, :374-381).
"""
import ctypes
import inspect

import numpy as np
import torch

from . import _lib
from .coherence import minimize_mutual_coherence
from .fsw_embedding import FSW_embedding, LinearSplitTallFn, LinearTallFn, GEMM_TN_MIN_ROWS
from .graph import BuildHint, build_csr, build_csr_coalesced

try:  # optional dependency, exactly the names the reference imports (fsw_conv.py:4-9)
    from torch_geometric.nn import MessagePassing as _Base
    from torch_geometric.graphgym.register import register_layer, register_pooling
    _HAVE_PYG = True
except Exception:  # pragma: no cover - torch_geometric is absent in the build image
    _Base = torch.nn.Module
    _HAVE_PYG = False

    def register_layer(name):
        return lambda cls: cls

    def register_pooling(name):
        return lambda cls: cls


@register_layer('fsw_conv')
class FSW_conv(_Base):
    def __init__(self,
                 in_channels, out_channels, edgefeat_dim=0,
                 embed_dim=None, learnable_embedding=True,
                 encode_vertex_degrees=True, vertex_degree_encoding_function='identity',
                 vertex_degree_encoding_scale=1.0, learnable_vertex_degree_encoding_scale=False, homog_degree_encoding=False,
                 vertex_degree_pad_thresh=1.0,
                 concat_self=True, message_weight_vs_self=1.0,
                 bias=True,
                 mlp_layers=1, mlp_hidden_dim=None,
                 mlp_activation_final=torch.nn.LeakyReLU(negative_slope=0.2),
                 mlp_activation_hidden=torch.nn.LeakyReLU(negative_slope=0.2),
                 mlp_init=None,
                 batchNorm_final=False, batchNorm_hidden=False,
                 dropout_final=0, dropout_hidden=0,
                 self_loop_weight=0, edge_weighting='unit',
                 device=None, dtype=torch.float32,
                 config=None):
        if _HAVE_PYG:
            super().__init__(aggr=None)
        else:
            super().__init__()
        config = dict(config) if config is not None else {}
        arg_names = {p.name for p in inspect.signature(FSW_conv.__init__).parameters.values()} - {'config', 'self'}
        for key in config:
            if key not in arg_names:
                raise ValueError(f"Invalid argument '{key}' in config")
        given = locals()
        for name in arg_names:
            if name not in config:
                config[name] = given[name]
        self.init_helper(**config)

    def init_helper(self, in_channels, out_channels, edgefeat_dim, embed_dim, learnable_embedding,
                    encode_vertex_degrees, vertex_degree_encoding_function, vertex_degree_encoding_scale,
                    learnable_vertex_degree_encoding_scale, homog_degree_encoding, vertex_degree_pad_thresh,
                    concat_self, message_weight_vs_self, bias, mlp_layers, mlp_hidden_dim, mlp_activation_final,
                    mlp_activation_hidden, mlp_init, batchNorm_final, batchNorm_hidden, dropout_final, dropout_hidden,
                    self_loop_weight, edge_weighting, device, dtype):
        assert edge_weighting in {'unit', 'gcn'}, 'invalid value passed in argument <edge_weighting>'
        assert vertex_degree_encoding_function in {'identity', 'sqrt', 'log'}, \
            'invalid value passed in argument <vertex_degree_encoding_function>'
        if mlp_hidden_dim is None:
            mlp_hidden_dim = max(in_channels, out_channels)
        if (mlp_layers == 0) and (concat_self is False):
            embed_dim = out_channels
        elif embed_dim is None:
            embed_dim = 2 * max(in_channels, out_channels)
        embedding_bias = (bias and mlp_layers == 0)                         # fsw_conv.py:237
        method = 'homog' if homog_degree_encoding else 'plain'              # fsw_conv.py:240

        self.in_channels, self.out_channels, self.embed_dim = in_channels, out_channels, embed_dim
        self.edgefeat_dim = edgefeat_dim
        self.concat_self = concat_self
        self.edge_weighting = edge_weighting
        self.self_loop_weight = self_loop_weight
        self.message_weight_vs_self = message_weight_vs_self
        mlp_input_dim = in_channels + embed_dim if concat_self else embed_dim

        if mlp_layers == 0:                                                  # fsw_conv.py:255-267
            self.mlp = None
            if concat_self:
                with torch.no_grad():
                    dim_reduct = torch.randn(size=(out_channels, mlp_input_dim), device=device, dtype=dtype)
                    dim_reduct = minimize_mutual_coherence(dim_reduct, report=False)          # fsw_conv.py:260-262
                self.dim_reduct = torch.nn.Parameter(dim_reduct, requires_grad=learnable_embedding)
            self.bn_final = torch.nn.BatchNorm1d(num_features=out_channels, device=device, dtype=dtype) if batchNorm_final else None
        else:                                                                # fsw_conv.py:269-310
            self.bn_final = None
            mods = []
            for i in range(mlp_layers):
                last = (i == mlp_layers - 1)
                in_curr = mlp_input_dim if i == 0 else mlp_hidden_dim
                out_curr = out_channels if last else mlp_hidden_dim
                layer = torch.nn.Linear(in_curr, out_curr, bias=bias, device=device, dtype=dtype)
                if mlp_init is not None:
                    init = {'xavier_uniform': torch.nn.init.xavier_uniform_, 'xavier_normal': torch.nn.init.xavier_normal_,
                            'kaiming_uniform': torch.nn.init.kaiming_uniform_, 'kaiming_normal': torch.nn.init.kaiming_normal_}
                    if mlp_init not in init:
                        raise RuntimeError('Invalid value passed at argument mlp_init')
                    init[mlp_init](layer.weight)
                    if bias:
                        torch.nn.init.zeros_(layer.bias)
                mods.append(layer)
                if (batchNorm_final if last else batchNorm_hidden):
                    mods.append(torch.nn.BatchNorm1d(num_features=out_curr, device=device, dtype=dtype))
                act = mlp_activation_final if last else mlp_activation_hidden
                if act is not None:
                    mods.append(act)
                drop = dropout_final if last else dropout_hidden
                if drop > 0:
                    mods.append(torch.nn.Dropout(p=drop))
            self.mlp = torch.nn.Sequential(*mods)

        # defined by the reference and never used in its forward (fsw_conv.py:312); kept for state_dict parity
        self.size_coeff = torch.nn.Parameter(torch.ones(1, device=device, dtype=dtype) / np.sqrt(embed_dim),
                                             requires_grad=learnable_embedding)
        self.fsw_embed = FSW_embedding(d_in=in_channels, d_out=embed_dim, d_edge=edgefeat_dim,
                                       learnable_slices=learnable_embedding, learnable_freqs=learnable_embedding,
                                       learnable_total_mass_encoding_scale=learnable_vertex_degree_encoding_scale,
                                       encode_total_mass=encode_vertex_degrees,
                                       total_mass_encoding_function=vertex_degree_encoding_function,
                                       total_mass_encoding_scale=vertex_degree_encoding_scale,
                                       total_mass_encoding_method=method,
                                       total_mass_pad_thresh=vertex_degree_pad_thresh,
                                       minimize_slice_coherence=True, freqs_init='spread',
                                       enable_bias=embedding_bias, device=device, dtype=dtype)
        device = device if device is not None else self.fsw_embed.get_device()
        self.to(device=device, dtype=dtype)

    # ------------------------------------------------------------------------------------------------
    def build_graph(self, edge_index, num_vertices, edge_features=None, chunk_rows=0):
        """edge_index [2, E] int64 (row 0 = sender, row 1 = recipient) -> CSRGraph.

        With edge features (edgefeat_dim > 0, fsw_conv.py:419-439) the adjacency is coalesced like the reference's:
        parallel edges become one entry with summed weight and summed feature vector; self loops carry zero features.

        Unit weighting without self loops carries no weight array at all (the unit fast path).  Self loops
        (fsw_conv.py:390-395) append n weighted edges; 'gcn' weighting (fsw_conv.py:406-409) divides every
        edge by sqrt(deg_recipient) * sqrt(deg_sender) with deg = weighted in-degree.  Parallel edges stay
        separate elements (see DESIGN.md "duplicates"), which gives the same sums as the reference's coalesce.
        chunk_rows > 0: degree bins per chunk of consecutive rows (graph.py; the multi-GPU pipeline).
        """
        if edge_features is not None:
            src, dst = edge_index[0], edge_index[1]
            ef = edge_features.detach().reshape(src.numel(), -1)
            w = None
            if self.self_loop_weight > 0:
                loops = torch.arange(num_vertices, device=edge_index.device, dtype=torch.int64)
                w = torch.cat([torch.ones(src.numel(), device=src.device, dtype=torch.float32),
                               torch.full((num_vertices,), float(self.self_loop_weight), device=src.device, dtype=torch.float32)])
                src, dst = torch.cat([src, loops]), torch.cat([dst, loops])
                ef = torch.cat([ef, torch.zeros((num_vertices, ef.shape[1]), device=ef.device, dtype=ef.dtype)])
            if self.edge_weighting == 'gcn':
                if w is None:
                    w = torch.ones(src.numel(), device=src.device, dtype=torch.float32)
                deg = torch.zeros(num_vertices, device=src.device, dtype=torch.float32).scatter_add_(0, dst, w)
                ds = torch.sqrt(deg)
                w = w / ds[dst] / ds[src]
            return build_csr_coalesced(dst, src, w, ef.contiguous(), num_vertices, num_vertices, want_slots=True)
        if self.cache_graph:
            # optional CSR reuse across calls / layers.  Off by default: the reference rebuilds its
            # adjacency on every forward (fsw_conv.py:352) and bench.py times the rebuild.  The cache entry keeps the
            # edge_index tensor itself alive, so its address cannot be handed to another batch by the caching allocator;
            # an in-place edit bumps _version.  The flags earlier inputs left behind are cleared on a hit.
            key = (edge_index._version, tuple(edge_index.shape), int(num_vertices), float(self.self_loop_weight),
                   self.edge_weighting, int(chunk_rows))
            hit = getattr(self, '_graph_cache', None)
            if hit is not None and hit[0] is edge_index and hit[1] == key:
                hit[2].clear_input_flags()
                return hit[2]
            self.cache_graph = False
            try:
                graph = self.build_graph(edge_index, num_vertices, chunk_rows=chunk_rows)
            finally:
                self.cache_graph = True
            self._graph_cache = (edge_index, key, graph)
            return graph
        src, dst = edge_index[0], edge_index[1]
        w = None
        if self.self_loop_weight > 0:
            loops = torch.arange(num_vertices, device=edge_index.device, dtype=torch.int64)
            w = torch.cat([torch.ones(src.numel(), device=src.device, dtype=torch.float32),
                           torch.full((num_vertices,), float(self.self_loop_weight), device=src.device, dtype=torch.float32)])
            src = torch.cat([src, loops])
            dst = torch.cat([dst, loops])
        if self.edge_weighting == 'gcn':
            if w is None:
                w = torch.ones(src.numel(), device=src.device, dtype=torch.float32)
            deg = torch.zeros(num_vertices, device=src.device, dtype=torch.float32).scatter_add_(0, dst, w)
            ds = torch.sqrt(deg)
            w = w / ds[dst] / ds[src]
        return build_csr(dst, src, w, num_vertices, num_vertices, want_invperm=self._fusable(), chunk_rows=chunk_rows,
                         hint=self._build_hint())

    def forward(self, vertex_features, edge_index, edge_features=None):
        """vertex_features [n, in_channels], edge_index [2, E] long -> [n, out_channels] (fsw_conv.py:331-369)."""
        emb_mod = self.fsw_embed
        assert vertex_features.dtype == emb_mod.get_dtype(), 'vertex_features has incorrect dtype (expected %s, got %s)' % (emb_mod.get_dtype(), vertex_features.dtype)
        assert vertex_features.device == emb_mod.get_device(), 'vertex_features has incorrect device (expected %s, got %s)' % (emb_mod.get_device(), vertex_features.device)
        assert edge_index.device == emb_mod.get_device(), 'edge_index has incorrect device (expected %s, got %s)' % (emb_mod.get_device(), edge_index.device)
        if self.edgefeat_dim > 0:
            num_edges = edge_index.shape[1]
            assert edge_features is not None, 'Edge features must be provided since edgefeat_dim > 0'
            assert edge_features.dim() in (1, 2), 'edge_features should have the shape (num_edges, edegfeat_dim) (or optionally (num_edges,) in the case edgefeat_dim=1)'
            if self.edgefeat_dim == 1:
                assert tuple(edge_features.shape) in {(num_edges,), (num_edges, 1)}, 'edge_features should have the shape (num_edges, edegfeat_dim) (or optionally (num_edges,) in the case edgefeat_dim=1)'
            else:
                assert tuple(edge_features.shape) == (num_edges, self.edgefeat_dim), 'edge_features must have the shape (num_edges, edgefeat_dim)'
            assert edge_features.dtype == vertex_features.dtype and edge_features.device == vertex_features.device
        else:
            assert edge_features is None, 'Edge features should not be provided since edgefeat_dim = 0'
        if vertex_features.device.type != 'cuda':
            raise RuntimeError("fsw_gnn_amd: forward needs tensors on a HIP device ('cuda'); there is no CPU path")
        needs_grad = torch.is_grad_enabled() and (vertex_features.requires_grad or any(p.requires_grad for p in self.parameters())
                                                  or (edge_features is not None and edge_features.requires_grad))
        n = vertex_features.size(0)
        if n == 0:
            return torch.zeros((0, self.out_channels), dtype=vertex_features.dtype, device=vertex_features.device)
        if emb_mod.get_dtype() == torch.float64:
            if getattr(self, '_node_parallel', False) or getattr(self, '_slice_parallel', None) is not None:
                raise NotImplementedError("fsw_gnn_amd: slice / node sharding exists for float32 layers only (the float64 build runs "
                                          "the generic kernels on one GPU)")
            # float64 build (the reference's test_conv.py runs the layer this way): the reference's own structure -- coalesced
            # COO adjacency and edge features (fsw_conv.py:384-447), FSW_embedding.forward on the generic kernels, torch tail
            adj, x_edge = self.adjacency_coo(edge_index, edge_features if self.edgefeat_dim > 0 else None, n, vertex_features.dtype)
            emb = emb_mod(vertex_features, W=adj, X_edge=x_edge, graph_mode=True)
            return self._tail(emb, vertex_features)
        x = vertex_features.contiguous()
        if getattr(self, '_node_parallel', False):
            if needs_grad:
                raise NotImplementedError("fsw_gnn_amd: node-parallel training is not implemented")
            return self._forward_node_parallel(x, edge_index)
        sp = getattr(self, '_slice_parallel', None)
        if sp is not None:
            return self._forward_slice_parallel(sp, x, vertex_features, edge_index, edge_features, needs_grad)
        graph = self.build_graph(edge_index, n, edge_features if self.edgefeat_dim > 0 else None)
        E = self.embed_dim
        scale = float(self.message_weight_vs_self) if self.concat_self else 1.0      # fsw_conv.py:357-358

        if needs_grad:
            # training path: differentiable embedding (HIP forward + backward kernels), the tail through torch autograd
            ef_in = edge_features.reshape(edge_index.shape[1], -1) if self.edgefeat_dim > 0 else None
            emb = emb_mod.embed_autograd(x, graph, edge_feat=ef_in)
            return self._tail(emb, vertex_features)

        prepared = None
        # The fused kernel covers the rows of at most REG_MAX_DEG neighbours; the stats read inside prepare() tells whether the
        # graph has longer ones, which are then finished by the long-row kernels + one GEMM on those rows (_finish_long_rows).
        if self._fusable():
            # fast path: the projection GEMM also produces x . W2^T + b, then ONE kernel does the neighbourhood
            # embedding and E . W1^T (+ activation); the embedding never reaches HBM (csrc/conv_fused.hip)
            lin = self.mlp[0]
            wq, w2 = self._fused_weight()
            y = torch.empty((n, lin.out_features), dtype=x.dtype, device=x.device)
            # up to 128 features the projection kernel produces x . W2^T + b as four more 32-column slabs of the same pass, rows in
            # degree-bin order (contiguous runs for the fused kernel); above, its slabs would take a third pass over X (3.8 ms at
            # 4M x 256): a BLAS GEMM writes the block in NODE order and the fused kernel reads whole rows of it by node id (no row
            # permutation: 1.1 ms at 4M rows)
            in_kernel = self.concat_self and self.in_channels <= 128
            by_node = self.concat_self and not in_kernel
            yin = torch.empty_like(y) if in_kernel else None
            lin2 = (w2, lin.bias.detach() if lin.bias is not None else None, yin) if in_kernel else None
            prepared = emb_mod.prepare(x, graph, linear2=lin2)
            if by_node:
                yin = torch.addmm(lin.bias.detach(), x, w2.t()) if lin.bias is not None else x @ w2.t()
            st = prepared["stats"]
            if prepared["unit_fast"]:
                next_module = self._fused_linear(graph, prepared, scale, wq, yin, y, by_node)     # every row of at most 32 neighbours
                nlong = st[_lib.STAT_NUM_LDS] + st[_lib.STAT_NUM_GLOBAL]
                if nlong > 0:
                    self._finish_long_rows(x, graph, prepared, scale, yin, y, next_module, by_node)
                for m in self.mlp[next_module:]:
                    y = m(y)
                return y

        if self._split_first_linear():
            emb = torch.empty((n, E), dtype=x.dtype, device=x.device)
            emb_mod.embed_into(x, graph, emb, out_scale=scale, prepared=prepared)
            return self._tail_split(emb, x)
        width = E + self.in_channels if self.concat_self else E
        buf = torch.empty((n, width), dtype=x.dtype, device=x.device)
        xc = buf[:, E:] if self.concat_self else None   # right half of cat((emb, x))
        if prepared is not None:                       # fusable configuration, but the graph has long rows
            emb_mod.embed_into(x, graph, buf, out_scale=scale, prepared=prepared)
            if xc is not None:
                xc.copy_(x)
        else:
            emb_mod.embed_into(x, graph, buf, out_scale=scale, x_copy=xc)   # X stored by the projection kernel
        return self._tail_buffer(buf)

    def adjacency_coo(self, edge_index, edge_features, num_vertices, dtype):
        """Coalesced COO adjacency adj[recipient, sender] and edge-feature tensor exactly as the reference builds them
        (FSW_conv.edge_index_to_adj, fsw_conv.py:384-447): parallel edges summed, optional self loops, 'gcn' weighting by the
        weighted in-degrees; differentiable in edge_features (torch's coalesce)."""
        dev = edge_index.device
        inds = edge_index.flip(0)
        vals = torch.ones(edge_index.shape[1], device=dev, dtype=dtype)
        if self.self_loop_weight > 0:
            loops = torch.arange(num_vertices, device=dev).reshape(1, num_vertices).repeat(2, 1)
            inds = torch.cat((inds, loops), dim=1)
            vals = torch.cat((vals, self.self_loop_weight * torch.ones(num_vertices, device=dev, dtype=dtype)), dim=0)
        adj = torch.sparse_coo_tensor(indices=inds, values=vals, size=(num_vertices, num_vertices)).coalesce()
        if self.edge_weighting == 'gcn':
            ai, av = adj.indices(), adj.values()
            ds = torch.sqrt(torch.zeros(num_vertices, device=dev, dtype=dtype).index_add_(0, ai[0], av))
            adj = torch.sparse_coo_tensor(ai, av / ds[ai[0]] / ds[ai[1]], adj.shape, is_coalesced=True)
        x_edge = None
        if edge_features is not None:
            ef = edge_features
            if self.self_loop_weight > 0:
                shape = list(ef.shape)
                shape[0] = num_vertices
                ef = torch.cat((ef, torch.zeros(shape, device=dev, dtype=dtype)), dim=0)
            size = tuple(adj.shape) if ef.dim() == 1 else tuple(adj.shape) + (self.edgefeat_dim,)
            x_edge = torch.sparse_coo_tensor(indices=inds, values=ef, size=size).coalesce()
        return adj, x_edge

    def _tail(self, emb, vertex_features):
        """concat with the vertex features, MLP / dim_reduct, final BatchNorm through torch autograd (fsw_conv.py:357-369)."""
        if (self._split_first_linear() and emb.is_cuda and emb.dtype == torch.float32 and emb.shape[0] >= GEMM_TN_MIN_ROWS
                and self.mlp[0].weight.requires_grad and torch.is_grad_enabled() and vertex_features.shape[0] == emb.shape[0]):
            # large training batches: the first Linear layer on the two operands where they lie (no concatenation), its weight
            # gradient on csrc/gemm_tn.hip
            out = LinearSplitTallFn.apply(emb, vertex_features.contiguous(), self.mlp[0].weight, self.mlp[0].bias,
                                          self.message_weight_vs_self)
            for m in self.mlp[1:]:
                out = m(out)
            return self.bn_final(out) if self.bn_final is not None else out
        h = torch.cat((self.message_weight_vs_self * emb, vertex_features), dim=-1) if self.concat_self else emb
        if (self.mlp is not None and isinstance(self.mlp[0], torch.nn.Linear) and h.is_cuda and h.dtype == torch.float32
                and h.shape[0] >= GEMM_TN_MIN_ROWS and self.mlp[0].weight.requires_grad and torch.is_grad_enabled()):
            # the first Linear layer with its weight gradient on csrc/gemm_tn.hip (a reduction over the rows), the rest as it is
            out = LinearTallFn.apply(h, self.mlp[0].weight, self.mlp[0].bias)
            for m in self.mlp[1:]:
                out = m(out)
            return self.bn_final(out) if self.bn_final is not None else out
        out = self.mlp(h) if self.mlp is not None else (torch.matmul(h, self.dim_reduct.transpose(0, 1)) if self.concat_self else h)
        return self.bn_final(out) if self.bn_final is not None else out

    def _split_first_linear(self):
        """Inference tail without the concat buffer: possible when the first module after cat((emb, x)) is a Linear layer."""
        return (self.concat_self and self.mlp is not None and isinstance(self.mlp[0], torch.nn.Linear)
                and self.mlp[0].in_features == self.embed_dim + self.in_channels)

    def _tail_split(self, emb, x):
        """Linear(cat((emb, x))) = emb . W1^T + x . W2^T + b: two GEMMs on the embedding and on x where they lie (no
        [n, embed_dim + in_channels] buffer -- 4 GB written and read again at 4M nodes x 256 features), then the rest of the tail
        (reference fsw_conv.py:357-369)."""
        E = self.embed_dim
        lin = self.mlp[0]
        W = lin.weight.detach()
        y = torch.addmm(lin.bias.detach(), emb, W[:, :E].t()) if lin.bias is not None else emb @ W[:, :E].t()
        y.addmm_(x, W[:, E:].t())
        for m in self.mlp[1:]:
            y = m(y)
        return self.bn_final(y) if self.bn_final is not None else y

    def _tail_buffer(self, buf):
        """The same tail on the concat buffer the kernels filled in place (inference)."""
        if self.mlp is not None:
            out = self.mlp(buf)
        elif self.concat_self:
            out = torch.matmul(buf, self.dim_reduct.transpose(0, 1))
        else:
            out = buf
        return self.bn_final(out) if self.bn_final is not None else out

    def _forward_slice_parallel(self, sp, x, vertex_features, edge_index, edge_features, needs_grad):
        """Slice-axis sharding over the ranks of sp['group'] (dist.py: gather / consumer / training forms)."""
        import torch.distributed as dist
        from . import dist as D
        group = sp['group']
        world = dist.get_world_size(group)
        emb_mod = self.fsw_embed
        n = x.shape[0]
        E = self.embed_dim
        scale = float(self.message_weight_vs_self) if self.concat_self else 1.0
        has_ef = self.edgefeat_dim > 0
        stats = sp.get('stats')
        if stats is not None:
            stats.clear()
        if needs_grad:
            if sp['output'] == 'sharded' or sp['mode'] in ('consumer', 'exchange'):
                raise NotImplementedError("fsw_gnn_amd: training under slice sharding takes the differentiable gather form and returns "
                                          "the replicated output; mode='consumer' / 'exchange' / output='sharded' are inference forms "
                                          "(call under torch.no_grad() or enable_slice_parallel(mode='auto'))")
            graph = self.build_graph(edge_index, n, edge_features if has_ef else None)
            ef_in = edge_features.reshape(edge_index.shape[1], -1) if has_ef else None
            emb = D.sharded_embed_autograd(emb_mod, x, graph, out_scale=1.0, group=group, edge_feat=ef_in)
            return self._tail(emb, vertex_features)
        chunk_rows = 0
        if world > 1 and not has_ef:
            nch = sp['chunks'] if sp['chunks'] else D.default_chunks(n)
            chunk_rows, _ = D.chunk_plan(n, world, nch)
        graph = self.build_graph(edge_index, n, edge_features if has_ef else None, chunk_rows=chunk_rows)
        rank = dist.get_rank(group)
        parts = D.slice_partition(emb_mod.nSlices, world)
        mode = sp['mode']
        prepared = None
        widest = max(b - a for a, b in parts)
        if world > 1 and mode in ('auto', 'consumer') and self._fusable(widest) and parts[0][1] > parts[0][0]:
            wq = D.consumer_weight(self, group)              # queued ahead of the stats read inside prepare()
            prepared = emb_mod.prepare(x, graph, slice_range=parts[rank])
            st = prepared["stats"]
            if prepared["unit_fast"] and st[_lib.STAT_NUM_LDS] == 0 and st[_lib.STAT_NUM_GLOBAL] == 0:
                res, next_module = D.consumer_forward(self, x, graph, prepared, scale, group, sp['output'], stats, wq=wq)
                if stats is not None:
                    stats["mode"] = "consumer"
                if sp['output'] == 'sharded':
                    # R [chunks, rows_per_rank, H]: the remaining modules see ROWS x features (BatchNorm1d would take dim 1 of
                    # a 3-D tensor for its channels), the pad rows past the last node are zeroed again afterwards
                    R, row0 = res
                    rest = self.mlp[next_module:]
                    if any(isinstance(m, torch.nn.modules.batchnorm._BatchNorm) and (m.training or not m.track_running_stats)
                           for m in rest):
                        raise NotImplementedError("fsw_gnn_amd: output='sharded' with a BatchNorm layer in batch-statistics mode "
                                                  "would normalise over this rank's rows only; call .eval() or use "
                                                  "output='replicated'")
                    if len(rest) > 0:
                        shape = R.shape
                        R2 = R.reshape(-1, shape[-1])
                        for m in rest:
                            R2 = m(R2)
                        R = R2.reshape(shape[0], shape[1], -1)
                        for c in range(shape[0]):
                            valid = min(max(n - int(row0[c]), 0), shape[1])
                            if valid < shape[1]:
                                R[c, valid:] = 0
                    return R, row0
                y = res
                for m in self.mlp[next_module:]:
                    y = m(y)
                return y
            if mode == 'consumer':
                raise NotImplementedError("fsw_gnn_amd: the sharded-consumer form needs unit weights and rows of at most %d neighbours"
                                          % _lib.REG_MAX_DEG)
        elif mode == 'consumer' and world > 1:
            raise NotImplementedError("fsw_gnn_amd: the sharded-consumer form needs the fused configuration (csrc/conv_fused.hip)")
        if world > 1 and mode == 'exchange':
            # all-to-all of the slice blocks to the owners of the rows, the whole tail on the owned rows (dist.py)
            tail_mods = list(self.mlp) if self.mlp is not None else []
            if self.bn_final is not None:
                tail_mods.append(self.bn_final)
            if any(isinstance(m, torch.nn.modules.batchnorm._BatchNorm) and (m.training or not m.track_running_stats) for m in tail_mods):
                raise NotImplementedError("fsw_gnn_amd: the exchange form runs the tail on this rank's rows only; a BatchNorm layer in "
                                          "batch-statistics mode needs all rows (call .eval() or use mode='gather')")
            if prepared is None:
                prepared = emb_mod.prepare(x, graph, slice_range=parts[rank])
            res = D.exchange_forward(self, x, graph, prepared, scale, group, sp['output'], stats)
            if stats is not None:
                stats["mode"] = "exchange"
            return res
        if sp['output'] == 'sharded':
            raise NotImplementedError("fsw_gnn_amd: output='sharded' exists for the consumer and exchange forms only")
        if self._split_first_linear():      # the same tail as on one GPU, so the gathered layer stays bit-identical to it
            emb = torch.empty((n, E), dtype=x.dtype, device=x.device)
            D.sharded_embed_into(emb_mod, x, graph, emb, out_scale=scale, group=group, x_copy=None, prepared=prepared, stats=stats)
            if stats is not None:
                stats["mode"] = "gather"
            return self._tail_split(emb, x)
        width = E + self.in_channels if self.concat_self else E
        buf = torch.empty((n, width), dtype=x.dtype, device=x.device)
        xc = buf[:, E:] if self.concat_self else None
        D.sharded_embed_into(emb_mod, x, graph, buf, out_scale=scale, group=group, x_copy=xc, prepared=prepared, stats=stats)
        if stats is not None:
            stats["mode"] = "gather"
        return self._tail_buffer(buf)

    # ------------------------------------------------------------------------------------------------
    def _build_hint(self):
        """This layer's own memory of the graphs it has seen (graph.BuildHint: which CSR build to take); not a parameter, not in
        the state_dict."""
        h = self.__dict__.get('_graph_hint')
        if h is None:
            h = self.__dict__['_graph_hint'] = BuildHint()
        return h

    fuse_linear = True   # class-level switch: set conv.fuse_linear = False to force the unfused kernels
    cache_graph = False  # set conv.cache_graph = True to reuse the CSR while edge_index is unchanged

    def _fusable(self, num_slices=None):
        """Static conditions of the fused embedding + Linear kernel (csrc/conv_fused.hip).  num_slices: the width of the slice block
        one call covers (default: all slices; under slice sharding a rank's block -- a layer too wide for the fused tile on one GPU
        fits it once its slices are spread over the ranks: BASELINE config 4, 1024 slices as 128 per GPU)."""
        emb = self.fsw_embed
        if not (self.fuse_linear and self.mlp is not None and isinstance(self.mlp[0], torch.nn.Linear)):
            return False
        if self.self_loop_weight > 0 or self.edge_weighting != 'unit' or emb.total_mass_pad_thresh > 1.0 or self.edgefeat_dim > 0:
            return False
        if emb.encode_total_mass and emb.total_mass_encoding_method != 'plain':
            return False
        width = emb.nSlices if num_slices is None else int(num_slices)
        return int(_lib.lib().fsw_conv_fused_lds_bytes(width, 1 if emb.encode_total_mass else 0)) <= 64 * 1024

    def _fused_weight(self, col0=0, K=None, want_w2=True):
        """(Wq, W2) of the first Linear layer W = [W1 | W2]: K columns of W1 from column col0 (default: all embed_dim
        columns) packed for 16-byte MFMA operand loads (layout: include/fsw_hip.h, fsw_conv_fused_f32) and a contiguous
        copy of W2.  Packed by ONE kernel launch on every call -- nothing is cached, so in-place edits of the weights
        (EMA / SWA swaps, clipping, weight.data.copy_) are always seen."""
        L = _lib.lib()
        lin = self.mlp[0]
        W = lin.weight.detach()
        Hout, E = W.shape[0], self.embed_dim
        K = E if K is None else K
        ldwq = (Hout + 31) // 32 * 32
        wq = torch.empty(int(L.fsw_packed_linear_floats(K, Hout)), dtype=W.dtype, device=W.device).view(-1, ldwq, 8)
        w2 = None
        if want_w2 and self.concat_self:
            w2 = torch.empty((Hout, self.in_channels), dtype=W.dtype, device=W.device)
        rc = L.fsw_pack_linear_f32(_lib.ptr(W), W.stride(0), Hout, col0, K, _lib.ptr(wq),
                                   ctypes.c_void_p(W.data_ptr() + 4 * E) if w2 is not None else None, self.in_channels,
                                   _lib.ptr(w2), self.in_channels, torch.cuda.current_stream(W.device).cuda_stream)
        _lib.check(rc, "fsw_pack_linear_f32")
        return wq, w2

    def _finish_long_rows(self, x, graph, prepared, scale, yin, y, next_module, yin_by_node=False):
        """Rows above REG_MAX_DEG neighbours of a layer that otherwise runs the fused kernel: their embeddings come from the
        long-row kernels (mid / wave-sort / hub / giant), the first Linear layer for those rows from one GEMM on the gathered
        rows.  A graph with a few hubs keeps the fused kernel for everything else."""
        emb_mod = self.fsw_embed
        lin = self.mlp[0]
        E = self.embed_dim
        bsh = graph.bin_start_host[0]
        p0, p1 = int(bsh[_lib.BIN_MID0]), int(bsh[_lib.NUM_BINS])         # perm positions of the long rows: one contiguous run
        rows = graph.perm[p0:p1].long()
        emb = torch.empty((graph.num_rows, E), dtype=x.dtype, device=x.device)   # only the long rows are written (and read)
        emb_mod.embed_into(x, graph, emb, out_scale=scale, prepared=prepared, long_rows_only=True)
        el = emb.index_select(0, rows)
        W1 = lin.weight.detach()[:, :E]
        if yin is not None:                       # x . W2^T + b of these rows: rows p0..p1-1 of the degree-ordered block (or by node id)
            yl = torch.addmm(yin.index_select(0, rows) if yin_by_node else yin[p0:p1], el, W1.t())
        else:
            yl = torch.addmm(lin.bias.detach(), el, W1.t()) if lin.bias is not None else el @ W1.t()
        if next_module == 2:
            yl = self.mlp[1](yl)
        y.index_copy_(0, rows, yl)

    def _fused_linear(self, graph, prepared, scale, wq, yin, y, yin_by_node=False):
        L = _lib.lib()
        emb = self.fsw_embed
        lin = self.mlp[0]
        act, slope, next_module = 0, 0.0, 1
        if len(self.mlp) > 1 and isinstance(self.mlp[1], torch.nn.LeakyReLU):
            act, slope, next_module = 2, float(self.mlp[1].negative_slope), 2
        elif len(self.mlp) > 1 and isinstance(self.mlp[1], torch.nn.ReLU):
            act, next_module = 1, 2
        has_mass = 1 if emb.encode_total_mass else 0
        bias = emb.bias.detach() if emb.enable_bias else None
        a = emb.make_args(graph, prepared["stats"], prepared["Xp"], prepared["ldp"], emb.freqs.detach(), emb.nSlices,
                          prepared["table"], None, 0, bias.data_ptr() if bias is not None else None, scale, has_mass)
        rc = L.fsw_conv_fused_f32(ctypes.byref(a), wq.data_ptr(), wq.shape[1],
                                  lin.bias.data_ptr() if lin.bias is not None else None, lin.out_features,
                                  yin.data_ptr() if yin is not None else None, yin.stride(0) if yin is not None else 0,
                                  1 if yin_by_node else 0, act, slope, y.data_ptr(), y.stride(0),
                                  torch.cuda.current_stream(y.device).cuda_stream)
        _lib.check(rc, "fsw_conv_fused_f32")
        return next_module

    def enable_node_parallel(self, group=None, enabled=True):
        """Shard the RECIPIENT rows over the ranks of `group`: rank r runs the whole layer for rows [r*ceil(n/G), ...) --
        CSR of its rows, projection of all senders (replicated), the fused embedding + Linear kernel -- and ONE all-gather
        of the out_channels-wide OUTPUT rebuilds the result (4 bytes * out_channels per node instead of the embed_dim-wide
        embedding that slice sharding has to move).  Every row is computed by the same kernel as on one GPU (the x . W2^T
        term by a BLAS GEMM instead of the projection kernel: agreement to 1e-7).
        Needs the configuration of the fused kernel (unit weights, first MLP module a Linear layer)."""
        if enabled and not self._fusable():
            raise NotImplementedError("fsw_gnn_amd: node-parallel needs the fused configuration (csrc/conv_fused.hip); use enable_slice_parallel")
        self._node_parallel = bool(enabled)
        self._node_parallel_group = group
        return self

    def _forward_node_parallel(self, x, edge_index, _emulate=None):
        """_emulate = (rank, world): this rank's work without the collective (timing on a box with fewer GPUs, tools/)."""
        import torch.distributed as dist
        group = getattr(self, '_node_parallel_group', None)
        world, rank = (_emulate[1], _emulate[0]) if _emulate else (dist.get_world_size(group), dist.get_rank(group))
        from .dist import node_block
        n = x.shape[0]
        per, r0, nl = node_block(n, world, rank)
        emb = self.fsw_embed
        lin = self.mlp[0]
        H = lin.out_features
        scale = float(self.message_weight_vs_self) if self.concat_self else 1.0
        y_all = torch.empty((world * per, H), dtype=x.dtype, device=x.device)
        y_loc = torch.empty((per, H), dtype=x.dtype, device=x.device)
        # modules the fused kernel absorbs (Linear + ReLU / LeakyReLU): every rank must stop at the same one
        next_module = 2 if len(self.mlp) > 1 and isinstance(self.mlp[1], (torch.nn.LeakyReLU, torch.nn.ReLU)) else 1
        if nl > 0:
            rel = edge_index[1] - r0
            mine = torch.nonzero((rel >= 0) & (rel < nl)).squeeze(1)                        # one compaction for both endpoints
            sub = edge_index.index_select(1, mine)
            graph = build_csr(sub[1] - r0, sub[0], None, nl, n, want_invperm=True, hint=self._build_hint())   # nl recipient rows, n sender columns
            # recipients outside [0, n) belong to no rank's block: flag them like the CSR build flags a bad endpoint (read with
            # the stats in prepare(): "adjacency index out of range", as on one GPU)
            oob = ((edge_index[1] < 0) | (edge_index[1] >= n)).any().to(torch.int32) * _lib.FLAG_INDEX_RANGE
            graph.stats_dev[_lib.STAT_FLAGS:_lib.STAT_FLAGS + 1].bitwise_or_(oob.reshape(1))
            wq, w2 = self._fused_weight()
            prepared = emb.prepare(x, graph)                    # Xp of ALL senders: replicated work, no communication
            st = prepared["stats"]
            xl = x[r0:r0 + nl]
            if prepared["unit_fast"] and st[_lib.STAT_NUM_LDS] == 0 and st[_lib.STAT_NUM_GLOBAL] == 0:
                yin = None
                if self.concat_self:                            # x . W2^T + b of the local rows, read by (local) node id
                    yin = torch.addmm(lin.bias, xl, w2.t()) if lin.bias is not None else xl @ w2.t()
                assert self._fused_linear(graph, prepared, scale, wq, yin, y_loc[:nl], yin is not None) == next_module
            else:                                               # long rows: unfused kernels on the local rows
                E = self.embed_dim
                buf = torch.empty((nl, E + self.in_channels if self.concat_self else E), dtype=x.dtype, device=x.device)
                emb.embed_into(x, graph, buf, out_scale=scale, prepared=prepared)
                if self.concat_self:
                    buf[:, E:].copy_(xl)
                h = lin(buf)
                y_loc[:nl].copy_(self.mlp[1](h) if next_module == 2 else h)
        if nl < per:
            y_loc[nl:].zero_()
        if _emulate:
            y_all[rank * per:(rank + 1) * per].copy_(y_loc)
        else:
            dist.all_gather_into_tensor(y_all, y_loc, group=group)
        y = y_all[:n]
        for m in self.mlp[next_module:]:
            y = m(y)
        return self.bn_final(y) if self.bn_final is not None else y

    def enable_slice_parallel(self, group=None, enabled=True, mode='auto', chunks=None, output='replicated', stats=None):
        """Shard the slice axis of the embedding over the ranks of `group` (dist.py).

        mode    'gather'    all-gather of the embedding blocks (bit-identical to one GPU), the tail replicated;
                'consumer'  the first Linear layer stays sharded: partial sums reduce-scattered, finished rows all-gathered
                            (needs the fused configuration; raises otherwise);
                'exchange'  ONE all-to-all hands every row block the slice blocks of all ranks, the owner runs the whole tail on its
                            rows, one all-gather of the output rows -- the fewest bytes from 4 ranks on (dist.py); any configuration;
                'auto'      consumer where it applies, else gather.  Training always takes the differentiable gather form.
        chunks  node-range chunks of the pipeline (collective of chunk c under the kernels of chunk c + 1); default by size.
        output  'replicated' (the reference's contract: every rank returns all rows) or, consumer form only, 'sharded':
                returns (rows [chunks, chunk_rows / world, out], first_row [chunks]) -- this rank's finished rows.
        stats   optional dict that every forward fills with the form taken and the bytes each rank sent."""
        assert mode in ('auto', 'gather', 'consumer', 'exchange') and output in ('replicated', 'sharded')
        self._slice_parallel = dict(group=group, mode=mode, chunks=chunks, output=output, stats=stats) if enabled else None
        return self

    # the reference defines these as empty stubs and never calls propagate() (fsw_conv.py:374-381)
    def aggregate(self, inputs, index):
        return

    def message(self, x_j):
        return

    def update(self, aggr_out):
        return


@register_pooling('fsw_readout')
class FSW_readout(FSW_conv):
    def forward(self, vertex_features, graph_index=None, batch_size=None):
        """Global pooling: every vertex sends to the node of its graph (reference fsw_conv.py:451-517)."""
        assert self.edgefeat_dim == 0, 'edgefeat_dim should equal zero in a global readout layer'
        num_vertices = vertex_features.shape[0]
        if graph_index is None:
            assert batch_size is None, 'batch_size must be None when graph_index is None'
            graph_index = torch.zeros(num_vertices, device=vertex_features.device, dtype=torch.int64)
        else:
            assert tuple(graph_index.shape) == (num_vertices,), 'graph_index should be of shape (num_vertices,)'
            assert graph_index.dtype == torch.int64, 'invalid dtype given in graph_index'
        if batch_size is None:
            batch_size = int(graph_index.max().item()) + 1
        emb_mod = self.fsw_embed
        assert vertex_features.device == emb_mod.get_device() and graph_index.device == emb_mod.get_device()
        assert vertex_features.dtype == emb_mod.get_dtype()
        src = torch.arange(num_vertices, device=vertex_features.device, dtype=torch.int64)
        if batch_size == 0 or num_vertices == 0:
            return torch.zeros((batch_size, self.out_channels), dtype=vertex_features.dtype, device=vertex_features.device)
        if emb_mod.get_dtype() == torch.float64:   # float64 build: the reference's adjacency [graph_index, arange] (fsw_conv.py:503-515)
            assert int(graph_index.min()) >= 0 and int(graph_index.max()) < batch_size, 'all entries of graph_index must be in the range 0,...,batch_size-1'
            adj = torch.sparse_coo_tensor(torch.stack((graph_index, src)), torch.ones(num_vertices, dtype=torch.float64, device=src.device),
                                          size=(batch_size, num_vertices)).coalesce()
            emb = emb_mod(vertex_features, W=adj, graph_mode=True)
            if self.mlp is not None:
                return self.mlp(emb)
            return torch.matmul(emb, self.dim_reduct.transpose(0, 1)) if self.concat_self else emb
        graph = build_csr(graph_index.contiguous(), src, None, batch_size, num_vertices)
        needs_grad = torch.is_grad_enabled() and (vertex_features.requires_grad or any(p.requires_grad for p in self.parameters()))
        if needs_grad:
            # training: the same differentiable embedding as FSW_conv (the reference readout is differentiable through
            # self.fsw_embed, fsw_conv.py:503-515); readout segments are long rows, covered by the long-row backward kernels
            emb = emb_mod.embed_autograd(vertex_features.contiguous(), graph)
        else:
            emb = torch.empty((batch_size, self.embed_dim), dtype=vertex_features.dtype, device=vertex_features.device)
            emb_mod.embed_into(vertex_features.contiguous(), graph, emb)
        if graph.flags & 1:
            raise AssertionError('all entries of graph_index must be in the range 0,...,batch_size-1')
        if self.mlp is not None:
            return self.mlp(emb)
        if self.concat_self:
            return torch.matmul(emb, self.dim_reduct.transpose(0, 1))
        return emb
