import sys, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from fsw_gnn_amd import FSW_conv
dev = torch.device("cuda:0")
x, ei = bench.make_inputs(bench.N_NODES, bench.N_EDGES, dev)
conv = FSW_conv(128, 128, embed_dim=1025, device=dev)
with torch.no_grad():
    y = conv(x, ei)
    assert torch.isfinite(y).all()
    ms = bench.timed_ms(lambda: conv(x, ei), 5, dev)
print("config 3 of BASELINE.json on one GPU: 1M nodes / 10M edges / 1024 slices: %.2f ms forward = %.3e edge*slices/s" % (ms, 9999936 * 1024 / ms * 1e3))
