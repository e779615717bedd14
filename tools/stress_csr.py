import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fsw_gnn_amd import build_csr
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(123)
shapes = []
for rows in (32768, 32769, 34815, 34816, 34817, 65535, 65536, 65537, 100000, 131071, 262144, 1048575, 1048576, 1048577, 2097151, 5000000):
    for edges in (262144, 300001, 1000000):
        shapes.append((rows, edges))
bad = 0
for rows, edges in shapes:
    for mode in ("uniform", "onerow", "lastrow", "skew"):
        if mode == "uniform":
            rec = torch.randint(0, rows, (edges,), generator=g)
        elif mode == "onerow":
            rec = torch.full((edges,), rows // 3, dtype=torch.int64)
        elif mode == "lastrow":
            rec = torch.randint(rows - 3, rows, (edges,), generator=g)
        else:
            rec = (torch.rand(edges, generator=g) ** 6 * rows).long().clamp(max=rows - 1)
        snd = torch.randint(0, rows, (edges,), generator=g)
        k = int(torch.randint(0, 3, (1,), generator=g))
        if k:
            idx = torch.randint(0, edges, (7,), generator=g)
            rec[idx[:3]] = rows + 5
            snd[idx[3:]] = -2
        w = torch.rand(edges, generator=g) if (rows + edges) % 2 else None
        r, s_ = rec.to(dev), snd.to(dev)
        wd = None if w is None else w.to(dev)
        a = build_csr(r, s_, wd, rows, rows, algo="lsd")
        b = build_csr(r, s_, wd, rows, rows, algo="two_level")
        sa, sb = a.stats_dev.cpu().tolist(), b.stats_dev.cpu().tolist()
        nnz = sa[6]
        ok = sa == sb and torch.equal(a.rowptr, b.rowptr) and torch.equal(a.col[:nnz], b.col[:nnz]) and torch.equal(a.bin_start, b.bin_start)
        if w is not None:
            ok = ok and torch.equal(a.w[:nnz], b.w[:nnz])
        if not ok:
            bad += 1
            print("MISMATCH", rows, edges, mode, sa, sb, flush=True)
print("done, mismatches:", bad, "of", len(shapes) * 4)
