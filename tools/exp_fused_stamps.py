"""Experiment: phases of k_conv_fused_unit per workgroup (tile of 32 rows) on BASELINE config 3.  Needs the instrumented build
(tools/build_variant.sh fstamps conv_fused "-DFSW_FUSED_STAMPS=1"; FSW_HIP_LIBRARY=_variants/libfsw_hip_fstamps.so)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from fsw_gnn_amd import FSW_conv, _lib
dev = torch.device("cuda:0")
n = bench.N_NODES
x, ei = bench.make_inputs(n, bench.N_EDGES, dev)
torch.manual_seed(4321)
conv = FSW_conv(128, 128, embed_dim=257, device=dev)
graph = conv.build_graph(ei, n)
wq, w2 = conv._fused_weight()
yin = torch.empty((n, 128), device=dev)
y = torch.empty((n, 128), device=dev)
prepared = conv.fsw_embed.prepare(x, graph, linear2=(w2, conv.mlp[0].bias.detach(), yin))
L = _lib.lib()
ms = bench.timed_ms(lambda: conv._fused_linear(graph, prepared, 1.0, wq, yin, y), 5, dev)
NW = 32000
buf = (ctypes.c_ulonglong * (NW * 4 * 8))()
L.fsw_debug_fused_stamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
L.fsw_debug_fused_stamps(buf, NW)
a = np.ctypeslib.as_array(buf).reshape(NW, 4, 8).astype(np.float64)
ok = a[:, :, 1].sum(axis=1) > 0
a = a[ok]
names = ["0 tile search + H init", "1 phase 1 (rows)", "2 Yin loads issued", "3 barrier (slowest wave)", "4 matrix phase",
         "5 staging + 2 barriers", "6 epilogue (Y rows)"]
print("k_conv_fused_unit (instrumented): %.3f ms; %d workgroups" % (ms, a.shape[0]))
tot = a[:, :, :7].sum(axis=2)
print("ticks per workgroup and wavefront: mean %.0f, p10 %.0f, p90 %.0f" % (tot.mean(), np.percentile(tot, 10), np.percentile(tot, 90)))
for i in range(7):
    print("   %-26s mean %8.0f  %5.1f %%   (wavefront 0: %8.0f, wavefront 3: %8.0f)" % (
        names[i], a[:, :, i].mean(), 100.0 * a[:, :, i].mean() / tot.mean(), a[:, 0, i].mean(), a[:, 3, i].mean()))
