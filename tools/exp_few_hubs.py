"""BASELINE config 3's graph plus a handful of hub rows: the layer with the fused kernel for the short rows + long-row kernels for
the hubs (default) against the unfused layer (conv.fuse_linear = False, what every graph with a long row ran before)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from fsw_gnn_amd import FSW_conv
dev = torch.device("cuda:0")
n, E = bench.N_NODES, bench.N_EDGES
x, ei = bench.make_inputs(n, E, dev)
g = torch.Generator(device="cpu").manual_seed(1)
hubs = []
for row, deg in ((5, 40), (6, 300), (7, 1500), (8, 5000), (9, 20000), (10, 60000)):
    hubs.append(torch.stack([torch.randperm(n, generator=g)[:deg], torch.full((deg,), row)]))
ei2 = torch.cat([ei] + [h.to(dev) for h in hubs], dim=1)
conv = FSW_conv(bench.D_FEAT, bench.OUT_CH, embed_dim=bench.EMBED_DIM, device=dev)
with torch.no_grad():
    for name, e in (("config 3", ei), ("config 3 + 6 hub rows (40 .. 60000 neighbours)", ei2)):
        conv.fuse_linear = True
        a = bench.timed_ms(lambda: conv(x, e), 10, dev)
        conv.fuse_linear = False
        b = bench.timed_ms(lambda: conv(x, e), 10, dev)
        print("%-50s fused/mixed %.2f ms   unfused %.2f ms" % (name, a, b), flush=True)
