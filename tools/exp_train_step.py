"""Timing of one FSW_conv training step (forward + backward) at BASELINE config 3 -- not the headline metric."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from fsw_gnn_amd import FSW_conv
dev = torch.device("cuda:0")
n, E = bench.N_NODES, bench.N_EDGES
x, ei = bench.make_inputs(n, E, dev)
conv = FSW_conv(128, 128, embed_dim=257, device=dev)
x.requires_grad_(True)
def step():
    conv.zero_grad(set_to_none=True); x.grad = None
    y = conv(x, ei)
    y.square().mean().backward()
for _ in range(2): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print("training step (fwd + bwd), 1M nodes / 10M edges / 256 slices: %.2f ms" % (dt * 1e3))
with torch.no_grad():
    print("inference forward: %.2f ms" % bench.timed_ms(lambda: conv(x.detach(), ei), 5, dev))
