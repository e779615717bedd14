"""Experiment: fsw_gemm_tn_f32 (weight-gradient shape A^T . B, K = 1M rows) -- bf16 x 3 against the fp32 matrix instruction
(FSW_GEMM_TN_EXACT_FP32=1) against torch (hipBLASLt)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from fsw_gnn_amd.fsw_embedding import gemm_tn
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
for K, M, N in ((1_000_000, 256, 128), (1_000_000, 128, 385), (1_000_000, 128, 128)):
    A = torch.randn((K, M), device=dev, generator=g)
    B = torch.randn((K, N), device=dev, generator=g)
    ref = (A[:200_000].double().t() @ B[:200_000].double())
    C = gemm_tn(A[:200_000], B[:200_000])
    err = float((C.double() - ref).abs().max() / ref.abs().max())
    ms = bench.timed_ms(lambda: gemm_tn(A, B), 10, dev)
    ms_t = bench.timed_ms(lambda: A.t() @ B, 5, dev)
    print("K=%d M=%d N=%d: gemm_tn %.3f ms (%.1f TFLOP/s), torch %.3f ms, max rel err %.2e" % (K, M, N, ms, 2.0 * K * M * N / ms / 1e9, ms_t, err), flush=True)
