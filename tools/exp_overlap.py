"""Experiment: CSR build and projection of BASELINE config 3 on two streams against one after the other (they are independent
apart from the degree-ordered Y2 block and the shared flags word -- this only measures what concurrency would buy)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from fsw_gnn_amd import _lib, build_csr
L = _lib.lib()
dev = torch.device("cuda:0")
n, E, S, d = bench.N_NODES, bench.N_EDGES, 256, 128
x, ei = bench.make_inputs(n, E, dev)
V = torch.randn((S + 128, d), device=dev)
Xp = torch.empty((n, S + 128), device=dev)
side = torch.cuda.Stream()
main = torch.cuda.current_stream()

def project(stream):
    _lib.check(L.fsw_project_f32(x.data_ptr(), n, d, d, V.data_ptr(), S + 128, d, Xp.data_ptr(), S + 128, None, 0, None, stream.cuda_stream), "project")

def sequential():
    build_csr(ei[1], ei[0], None, n, n, want_invperm=True)
    project(main)

def overlapped(first):
    side.wait_stream(main)
    if first == "csr":
        build_csr(ei[1], ei[0], None, n, n, want_invperm=True)
        project(side)
    else:
        project(side)
        build_csr(ei[1], ei[0], None, n, n, want_invperm=True)
    main.wait_stream(side)

def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

print("projection alone (384 columns)   %.3f ms" % timed(lambda: project(main)))
print("CSR build alone                  %.3f ms" % timed(lambda: build_csr(ei[1], ei[0], None, n, n, want_invperm=True)))
print("one after the other              %.3f ms" % timed(sequential))
print("two streams, CSR enqueued first  %.3f ms" % timed(lambda: overlapped("csr")))
print("two streams, projection first    %.3f ms" % timed(lambda: overlapped("proj")))
