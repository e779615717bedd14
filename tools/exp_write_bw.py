"""Experiment: what does HBM take for the projection's traffic shape?  (2.05 GB per launch: 0.51 GB read, 1.54 GB written.)
Pure fills, a pure read (sum) and a read-1-write-3 elementwise op of the same sizes, timed with HIP events."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0")
n = 1_000_000
x = torch.randn(n, 128, device=dev)
a = torch.empty(n, 256, device=dev)
b = torch.empty(n, 128, device=dev)
big = torch.empty(n, 384, device=dev)
def t(name, fn, gb):
    ms = bench.timed_ms(fn, 20, dev)
    print("%-46s %.3f ms  %.2f TB/s" % (name, ms, gb / ms), flush=True)
t("fill 1.02 GB (n x 256 floats)", lambda: a.fill_(1.0), 1.024)
t("fill 1.54 GB (n x 384 floats)", lambda: big.fill_(1.0), 1.536)
t("read 0.51 GB (sum of n x 128)", lambda: x.sum(), 0.512)
t("copy 0.51 GB -> 0.51 GB", lambda: b.copy_(x), 1.024)
t("read 0.51 GB, write 1.54 GB (repeat 3x)", lambda: torch.cat([x, x, x], dim=1, out=big), 2.048)
