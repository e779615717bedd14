"""Throughput of the stand-alone segmented cumulative sum (csrc/segcumsum.hip, fsw_segcumsum).

    python tools/bench_segcumsum.py [--elems 256000000] [--mean-seg 10] [--dtype f32|f64] [--ids i64|i32] [--reps 5]

Algorithmic bytes per element: value read + id read + value written (4 + 8 + 4 for float32 / int64 -- the reference's
layout, fsw_embedding.py:2888).  Prints one JSON line; `--elems 2560000000` is BASELINE config 3's full E x S."""
import argparse, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fsw_gnn_amd import segcumsum   # noqa: E402
import bench                        # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--elems", type=int, default=256_000_000)
    ap.add_argument("--mean-seg", type=float, default=10.0)
    ap.add_argument("--dtype", choices=("f32", "f64"), default="f32")
    ap.add_argument("--ids", choices=("i64", "i32"), default="i64")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--reverse", action="store_true")
    ap.add_argument("--no-check", action="store_true",
                    help="skip the 1M-element correctness window: under rocprofv3 every k_segscan_chained call in the kernel "
                         "statistics is then a full-size call (the window's 7 us launch used to drag the average down)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    n = args.elems
    vdt = torch.float32 if args.dtype == "f32" else torch.float64
    idt = torch.int64 if args.ids == "i64" else torch.int32
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    heads = torch.rand(n, device=dev, generator=g) < (1.0 / args.mean_seg)
    ids = torch.cumsum(heads, 0, dtype=torch.int64).to(idt)
    del heads
    vals = torch.rand(n, device=dev, generator=g, dtype=vdt)

    def run():
        segcumsum(vals, ids, reverse=args.reverse)

    ms = bench.timed_ms(run, args.reps, dev)
    bpe = 2 * vals.element_size() + ids.element_size()
    err = None
    if not args.no_check:
        # a window of the FULL-SIZE result against torch (no extra launch of the kernel: the window is cut out of a full call)
        m = min(n, 1_000_000)
        full = segcumsum(vals, ids, reverse=args.reverse)
        lo = (n - m) if args.reverse else 0
        got = full[lo:lo + m].double()
        if args.reverse:
            got = got.flip(0)
        del full
        vv, ii = vals[lo:lo + m].double(), ids[lo:lo + m].long()
        if args.reverse:
            vv, ii = vv.flip(0), ii.flip(0)
        cs = torch.cumsum(vv, 0)
        first = torch.ones(m, dtype=torch.bool, device=dev)
        first[1:] = ii[1:] != ii[:-1]
        start_idx = torch.cummax(torch.where(first, torch.arange(m, device=dev), torch.zeros((), dtype=torch.long, device=dev)), 0).values
        ref = cs - (cs[start_idx] - vv[start_idx])
        err = float((got - ref).abs().max() / ref.abs().max())
    print(json.dumps({"kernel": "k_segscan_chained", "elements": n, "mean_segment": args.mean_seg, "values": args.dtype, "ids": args.ids,
                      "reverse": args.reverse, "ms": ms, "algorithmic_bytes_per_element": bpe, "GBps": n * bpe / ms / 1e6,
                      "frac_of_8TBps": n * bpe / ms / 1e6 / 8000.0, "max_rel_err_vs_torch_1M_window": err}))


if __name__ == "__main__":
    main()
