import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from tests.test_hip_properties import _rmat_on_device
from fsw_gnn_amd import FSW_conv
dev = torch.device("cuda:0")
n, E_, d, S = 1 << 22, 64_000_000, 256, 256
ei = _rmat_on_device(22, E_, 22, dev)
X = torch.randn((n, d), device=dev, generator=torch.Generator(device=dev).manual_seed(3))
conv = FSW_conv(d, S + 1, mlp_layers=0, concat_self=False, bias=False, device=dev)
with torch.no_grad():
    base = conv(X, ei)
deg = torch.bincount(ei[1], minlength=n)
bad = torch.nonzero(base[:, 0] != deg.float()).flatten()
print("bad rows", bad.numel(), "of", n)
print("degrees of bad rows:", torch.unique(deg[bad])[:50].tolist())
print("values:", base[bad[:10], 0].tolist(), deg[bad[:10]].tolist())
edges = [0,32,256,512,1024,2048,4096,8192,16384,32768,1<<30]
for lo,hi in zip(edges[:-1],edges[1:]):
    m=(deg>lo)&(deg<=hi); print(lo,hi,int(m.sum()), int(((base[:,0]!=deg.float())&m).sum()))
from fsw_gnn_amd import build_csr, _lib
graph = build_csr(ei[1].contiguous(), ei[0].contiguous(), None, n, n)
bs = graph.bin_start.cpu().numpy(); perm = graph.perm.cpu().numpy()
b = _lib.REG_MAX_DEG + 1 + len(_lib.MID_SIZES)
lo, hi = bs[b], bs[b+1]
print("bin", b, "rows", hi-lo)
badset = set(bad.cpu().numpy().tolist())
pos = np.array([p for p in range(lo, hi) if perm[p] in badset]) - lo
print("bad positions: min", pos.min(), "max", pos.max(), "count", pos.size, "first 20", pos[:20].tolist())
print("pos mod 8 histogram", np.bincount(pos % 8, minlength=8).tolist())
print("whole row zero?", float(base[bad[:100]].abs().max()))
d_b = deg[bad].cpu().numpy(); m = ((deg>256)&(deg<=512)); d_g = deg[m & (base[:,0]==deg.float())].cpu().numpy()
print("bad deg min/max", d_b.min(), d_b.max(), "good deg min/max", d_g.min(), d_g.max())
