import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fsw_gnn_amd import FSW_conv, synth
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
dev = torch.device("cuda:0")
t = torch.full((4,), float(rank + 1), device=dev)
dist.all_reduce(t)
torch.cuda.synchronize()
print(rank, "all_reduce cuda gloo ->", t.tolist(), flush=True)
n, E, d = 3000, 30000, 16
ei = torch.from_numpy(synth.er_multigraph(n, E, seed=7)).to(dev)
X = torch.from_numpy(synth.features(n, d, seed=8)).to(dev)
torch.manual_seed(3)
conv = FSW_conv(d, 12, embed_dim=31, device=dev)
def grads(sharded):
    conv.enable_slice_parallel(None, enabled=sharded)
    conv.zero_grad()
    Xg = X.clone().requires_grad_(True)
    out = conv(Xg, ei)
    torch.manual_seed(11)
    (out * torch.randn_like(out)).sum().backward()
    ps = {k: p.grad.clone() for k, p in conv.named_parameters() if p.grad is not None}
    ps["X"] = Xg.grad.clone()
    return out.detach(), ps
o1, g1 = grads(False)
o2, g2 = grads(True)
for k in g1:
    a, b = g2[k], g1[k]
    print(rank, k, tuple(a.shape), "rel", float((a - b).abs().max() / b.abs().max()), flush=True)
    if k == "fsw_embed.projVecs":
        for r0 in range(0, 30, 5):
            print(rank, "  rows", r0, float((a[r0:r0+5] - b[r0:r0+5]).abs().max()), float(b[r0:r0+5].abs().max()), float(a[r0:r0+5].abs().max()), flush=True)
dist.barrier()
