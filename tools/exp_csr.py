"""Timing of the two CSR builds (fsw_graph_build: three LSD passes; fsw_graph_build_two_level: partition pass + bucket kernel).

    python tools/exp_csr.py                 BASELINE config 3's edge list (1M rows, 10M random edges)
    python tools/exp_csr.py --rmat 22 --edges 64000000
"""
import argparse, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from fsw_gnn_amd import build_csr, synth
ap = argparse.ArgumentParser()
ap.add_argument("--rmat", type=int, default=0)
ap.add_argument("--edges", type=int, default=bench.N_EDGES)
args = ap.parse_args()
dev = torch.device("cuda:0")
if args.rmat:
    n = 1 << args.rmat
    ei = torch.from_numpy(synth.rmat_graph(args.rmat, args.edges, 7)).to(dev)
else:
    n = bench.N_NODES
    _, ei = bench.make_inputs(n, args.edges, dev)
for algo in ("lsd", "two_level", "lsd", "two_level"):
    ms = bench.timed_ms(lambda: build_csr(ei[1], ei[0], None, n, n, want_invperm=True, algo=algo), 10, dev)
    print("%-10s %d rows / %d edges: %.3f ms" % (algo, n, ei.shape[1], ms), flush=True)
