"""Per-rank compute of the slice-sharded FSW_conv forward on ONE GPU (no process group): rank 0's share of a world of
2 / 4 / 8 with every collective replaced by a local copy (dist.COLLECTIVES_ENABLED = False).  What the multi-GPU step costs
besides its collectives: replicated CSR build, projection of the rank's slices, fused kernel on its slice block and its
columns of the first Linear layer, the local epilogue.  python tools/exp_slice_shard.py [--mode consumer|gather]"""
import argparse, os, sys
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from fsw_gnn_amd import FSW_conv
from fsw_gnn_amd import dist as D

ap = argparse.ArgumentParser()
ap.add_argument("--mode", default="consumer")
ap.add_argument("--chunks", type=int, default=0)
ap.add_argument("--worlds", default="2,4,8")
ap.add_argument("--slices", type=int, default=bench.N_SLICES)
args = ap.parse_args()
dev = torch.device("cuda:0")
x, ei = bench.make_inputs(bench.N_NODES, bench.N_EDGES, dev)
torch.manual_seed(4321)
conv = FSW_conv(bench.D_FEAT, bench.OUT_CH, embed_dim=args.slices + 1, device=dev)
with torch.no_grad():
    print("single GPU: %.3f ms" % bench.timed_ms(lambda: conv(x, ei), 10, dev), flush=True)
    D.COLLECTIVES_ENABLED = False
    for world in [int(w) for w in args.worlds.split(',')]:
        dist.get_world_size = lambda group=None, w=world: w
        dist.get_rank = lambda group=None: 0
        for output in ("replicated", "sharded"):
            if args.mode == "gather" and output == "sharded":
                continue
            st = {}
            conv.enable_slice_parallel(None, mode=args.mode, chunks=args.chunks or None, output=output, stats=st)
            ms = bench.timed_ms(lambda: conv(x, ei), 10, dev)
            print("world %d rank 0 compute (%s form, output %s): %.3f ms; collectives would move %.0f MB per rank (%s)" % (
                world, st.get("mode"), output, ms, st.get("bytes_sent_per_rank", 0) / 1e6, st.get("collective")), flush=True)
