"""Timing of one FSW_conv training step (forward + backward) -- not the headline metric.

    python tools/exp_train_step.py            BASELINE config 3 (ER multigraph, every row on the <= 32 register path)
    python tools/exp_train_step.py --rmat 20  RMAT graph with 2^20 vertices and 10M edges (hubs up to 41 300 neighbours)
    python tools/exp_train_step.py --rmat 22 --edges 64000000 --feat 256 --forward-only [--gcn]   BASELINE config 5's shape
"""
import argparse, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from fsw_gnn_amd import FSW_conv, synth
ap = argparse.ArgumentParser()
ap.add_argument("--rmat", type=int, default=0)
ap.add_argument("--forward-only", action="store_true")
ap.add_argument("--edges", type=int, default=bench.N_EDGES)
ap.add_argument("--feat", type=int, default=128)
ap.add_argument("--gcn", action="store_true", help="edge_weighting='gcn': general edge weights (the (key, weight) kernels)")
args = ap.parse_args()
dev = torch.device("cuda:0")
n, E = bench.N_NODES, args.edges
if args.rmat:
    n = 1 << args.rmat
    ei = torch.from_numpy(synth.rmat_graph(args.rmat, E, 7)).to(dev)
    x = torch.from_numpy(synth.features(n, args.feat, 3)).to(dev)
else:
    x, ei = bench.make_inputs(n, E, dev)
    assert args.feat == 128
conv = FSW_conv(args.feat, 128, embed_dim=257, device=dev, edge_weighting='gcn' if args.gcn else 'unit')
with torch.no_grad():
    print("inference forward, %d nodes / %d edges / %d feat / 256 slices: %.2f ms (max in-degree %d)" % (
        n, E, args.feat, bench.timed_ms(lambda: conv(x, ei), 5, dev), int(torch.bincount(ei[1]).max())), flush=True)
if not args.forward_only:
    x.requires_grad_(True)
    def step():
        conv.zero_grad(set_to_none=True); x.grad = None
        y = conv(x, ei)
        y.square().mean().backward()
    for _ in range(2): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print("training step (fwd + bwd): %.2f ms" % (dt * 1e3))
