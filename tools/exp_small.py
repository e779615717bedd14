"""Timing of BASELINE config 1 (demo_conv.py shape: 10k nodes / 100k edges / 64 features / 128 slices) on one GPU."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from fsw_gnn_amd import FSW_conv
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(1)
n, E = 10_000, 100_000
x = torch.randn((n, 64), generator=g).to(dev)
ei = torch.randint(0, n, (2, E), generator=g, dtype=torch.int64).to(dev)
conv = FSW_conv(64, 64, embed_dim=129, device=dev)
with torch.no_grad():
    print("config 1 (10k nodes / 100k edges / 64 feat / 128 slices): %.3f ms forward" % bench.timed_ms(lambda: conv(x, ei), 50, dev))
