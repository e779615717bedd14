"""Experiment: per-rank compute time of the recipient-row sharded FSW_conv forward (config 3) without the collective,
for world sizes 1, 2, 4, 8 -- what one GPU of an N-GPU job does before the all-gather of its output rows."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from fsw_gnn_amd import FSW_conv
dev = torch.device("cuda:0")
n, E = bench.N_NODES, bench.N_EDGES
x, ei = bench.make_inputs(n, E, dev)
torch.manual_seed(4321)
conv = FSW_conv(128, 128, embed_dim=257, device=dev)
with torch.no_grad():
    print("single GPU forward: %.3f ms" % bench.timed_ms(lambda: conv(x, ei), 10, dev), flush=True)
    for world in (2, 4, 8):
        for rank in (0, world - 1):
            ms = bench.timed_ms(lambda: conv._forward_node_parallel(x, ei, _emulate=(rank, world)), 10, dev)
            out_mb = 4.0 * 128 * n / 1e6
            print("world %d rank %d: compute %.3f ms; all-gather of %.0f MB per rank incoming (%.0f MB over each of %d links)"
                  % (world, rank, ms, out_mb * (world - 1) / world, out_mb / world, world - 1), flush=True)

# breakdown of rank 0 of 8
from fsw_gnn_amd.graph import build_csr
world, rank = 8, 0
per = -(-n // world); r0 = rank * per; nl = min(r0 + per, n) - r0
with torch.no_grad():
    def filt():
        mine = (ei[1] >= r0) & (ei[1] < r0 + nl)
        return ei[1][mine] - r0, ei[0][mine]
    dl, sl = filt()
    graph = build_csr(dl, sl, None, nl, n, want_invperm=True)
    wq, w2 = conv._fused_weight()
    prepared = conv.fsw_embed.prepare(x, graph)
    lin = conv.mlp[0]
    yin = torch.addmm(lin.bias, x[r0:r0 + nl], w2.t()).index_select(0, graph.perm.long())
    y = torch.empty((nl, 128), device=dev)
    print("rank 0 of 8: edge filter %.3f ms, CSR build %.3f ms, projection+table+stats %.3f ms, x.W2^T %.3f ms, fused kernel %.3f ms" % (
        bench.timed_ms(filt, 10, dev), bench.timed_ms(lambda: build_csr(dl, sl, None, nl, n, want_invperm=True), 10, dev),
        bench.timed_ms(lambda: conv.fsw_embed.prepare(x, graph), 10, dev),
        bench.timed_ms(lambda: torch.addmm(lin.bias, x[r0:r0 + nl], w2.t()).index_select(0, graph.perm.long()), 10, dev),
        bench.timed_ms(lambda: conv._fused_linear(graph, prepared, 1.0, wq, yin, y), 10, dev)))
