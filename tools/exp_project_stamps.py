"""Experiment: where does a tile of k_project_bf3 spend its cycles?  Needs the instrumented build
(tools/build_variant.sh stamps project "-DFSW_PROJECT_STAMPS=1"; FSW_HIP_LIBRARY=_variants/libfsw_hip_stamps.so).
Prints, per wave role (slab waves = matrix + move, helper waves = move only), the mean s_memtime cycles per tile iteration between
consecutive stamps of the kernel's loop (csrc/project.hip: FSW_STAMP)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from fsw_gnn_amd import FSW_conv, _lib
dev = torch.device("cuda:0")
x, ei = bench.make_inputs(bench.N_NODES, bench.N_EDGES, dev)
torch.manual_seed(4321)
conv = FSW_conv(128, 128, embed_dim=257, device=dev)
graph = conv.build_graph(ei, bench.N_NODES)
wq, w2 = conv._fused_weight()
yin = torch.empty((bench.N_NODES, 128), device=dev)
lin2 = (w2, conv.mlp[0].bias.detach(), yin)
L = _lib.lib()
import numpy as np
buf = (ctypes.c_ulonglong * (1024 * 12 * 8))()
L.fsw_debug_project_stamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
conv.fsw_embed.prepare(x, graph, linear2=lin2)
L.fsw_debug_project_stamps(buf)              # clear
reps = 5
ms = bench.timed_ms(lambda: conv.fsw_embed.prepare(x, graph, linear2=lin2), reps, dev)   # timed_ms runs reps + 1 calls
L.fsw_debug_project_stamps(buf)
a = np.ctypeslib.as_array(buf).reshape(1024, 12, 8).astype(np.float64)
names = ["0 wait loads + split + A planes", "1 issue loads t+2", "2 issue matrix instr", "3 drain acc -> staging", "4 barrier",
         "5 staging -> output stores", "6 loop tail"]
print("projection (instrumented): %.3f ms per call" % ms)
for w in range(12):
    it = a[:, w, 7].sum()
    if it == 0:
        continue
    per = a[:, w, :7].sum(axis=0) / it
    print("wave %2d: %7.0f cycles per tile | " % (w, per.sum()) + "  ".join("%s %6.0f" % (names[i].split()[0], per[i]) for i in range(7)))
it = a[:, :, 7].sum()
per = a[:, :, :7].sum(axis=(0, 1)) / it
print("all waves: %.0f cycles per tile iteration (s_memtime ticks)" % per.sum())
for i in range(7):
    print("   %-34s %8.0f  %5.1f %%" % (names[i], per[i], 100.0 * per[i] / per.sum()))
