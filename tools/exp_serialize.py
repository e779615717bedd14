"""Experiment: does blocking the slice axis (Xp chunk resident in the 256 MiB Infinity Cache) speed up the gather?"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fsw_gnn_amd import FSW_conv
import bench
dev = torch.device("cuda:0")
n, E = bench.N_NODES, bench.N_EDGES
x, ei = bench.make_inputs(n, E, dev)
conv = FSW_conv(128, 128, embed_dim=257, device=dev)
with torch.no_grad():
    graph = conv.build_graph(ei, n)
    out = torch.empty((n, 257 + 128), device=dev)
    for ser in (None, 128, 64, 32):
        f = lambda: conv.fsw_embed.embed_into(x, graph, out, serialize_num_slices=ser)
        ms = bench.timed_ms(f, 5, dev)
        print("serialize_num_slices=%s: project+table+embed = %.3f ms" % (ser, ms), flush=True)
