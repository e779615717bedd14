"""Experiment: where does a line of k_embed_mergepath_w spend its time?  Needs the instrumented build
(tools/build_variant.sh mpstamps embed_hub_2 "-DFSW_MP_STAMPS=1"; FSW_HIP_LIBRARY=_variants/libfsw_hip_mpstamps.so).
Times one degree class of the RMAT-22 graph with general weights (default 8193..16384) and prints the mean s_memtime ticks per line
between the stamps of the kernel (csrc/embed_hub.hip, csrc/merge_path.h: FSW_MP_MARK)."""
import argparse, ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fsw_gnn_amd import FSW_embedding, _lib, synth
from fsw_gnn_amd.graph import build_csr
ap = argparse.ArgumentParser()
ap.add_argument("--lo", type=int, default=8192)
ap.add_argument("--hi", type=int, default=16384)
args = ap.parse_args()
dev = torch.device("cuda", 0)
scale, edges, S, d = 22, 64_000_000, 256, 256
n = 1 << scale
ei = torch.from_numpy(synth.rmat_graph(scale, edges, 7)).to(dev)
w = torch.from_numpy(synth.edge_weights(edges, 9)).to(dev)
x = torch.from_numpy(synth.features(n, d, 3)).to(dev)
emb = FSW_embedding(d, S + 1, device=dev, encode_total_mass=True)
L = _lib.lib()
stream = torch.cuda.current_stream(dev).cuda_stream
degt = torch.bincount(ei[1], minlength=n)
keep = (degt[ei[1]] > args.lo) & (degt[ei[1]] <= args.hi)
sub = ei[:, keep].contiguous()
g = build_csr(sub[1], sub[0], w[keep].contiguous(), n, n)
stg = g.stats()
prepared = emb.prepare(x, g)
out = torch.empty((n, S + 1), dtype=torch.float32, device=dev)
scratch = torch.empty(int(L.fsw_embed_scratch_bytes(int(stg[_lib.STAT_MAX_DEGREE]))), dtype=torch.uint8, device=dev)
a = emb.make_args(g, stg, prepared["Xp"], prepared["ldp"], emb.freqs.detach(), S, prepared["table"], out.data_ptr(), out.stride(0), None, 1.0, 1,
                  scratch=scratch)
a.num_zero_rows = 0
buf = (ctypes.c_ulonglong * (512 * 4 * 16))()
L.fsw_debug_mergepath_stamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
_lib.check(L.fsw_embed_f32(ctypes.byref(a), stream), "embed")
L.fsw_debug_mergepath_stamps(buf)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
_lib.check(L.fsw_embed_f32(ctypes.byref(a), stream), "embed")
ev1.record()
torch.cuda.synchronize()
L.fsw_debug_mergepath_stamps(buf)
s = np.ctypeslib.as_array(buf).reshape(512, 4, 16).astype(np.float64)
names = ["0 row header", "1 blocks gathered", "2 blocks sorted", "3 parked in scratch", "4 fence + barrier", "5 tile boundaries",
         "6 tiles staged", "7 split + serial merge", "8 readout + barrier", "9 level end", "10 sum + store"]
print("class %d..%d: %d edges, %.3f ms (instrumented)" % (args.lo + 1, args.hi, int(keep.sum()), ev0.elapsed_time(ev1)))
lines = s[:, 0, 15].sum()
per = s[:, 0, :11].sum(axis=0) / lines
print("wavefront 0 of every workgroup: %.0f ticks per line (%d lines)" % (per.sum(), int(lines)))
for i in range(11):
    print("   %-26s %9.0f  %5.1f %%" % (names[i], per[i], 100.0 * per[i] / per.sum()))
