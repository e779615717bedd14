"""Experiment: projection kernel time and HBM rate over feature dims (n = 1M rows, S = 256 slices)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fsw_gnn_amd import _lib
import bench
L = _lib.lib()
dev = torch.device("cuda:0")
n, S = 1_000_000, 256
stream = torch.cuda.current_stream(dev).cuda_stream
for d in (32, 64, 100, 128, 192, 256, 512):
    X = torch.randn((n, d), device=dev)
    V = torch.randn((S, d), device=dev)
    Xp = torch.empty((n, S), device=dev)
    fn = lambda: _lib.check(L.fsw_project_f32(X.data_ptr(), n, d, d, V.data_ptr(), S, d, Xp.data_ptr(), S, None, 0, None, stream), "project")
    ms = bench.timed_ms(fn, 5, dev)
    gb = 4.0 * n * (d + S) / 1e9
    print("d = %4d  %.3f ms  %.0f GB/s of algorithmic traffic, %.1f TFLOP/s" % (d, ms, gb / ms * 1e3, 2.0 * n * d * S / ms / 1e9), flush=True)
