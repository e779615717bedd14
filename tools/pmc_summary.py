#!/usr/bin/env python3
"""Turns rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter_collection.csv files into profiles/pmc_traffic_latest.json.

    python tools/pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> [out.json]

Units and corrections as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes: both counters are in KiB;
on gfx950 FETCH_SIZE reports half of the bytes of a streamed read (TCC_EA0_RDREQ x 64 B for 128-B requests), so it
is doubled; WRITE_SIZE is exact.  The doubling is calibrated in the same run on k_project, whose algorithmic reads
are known (X read once per 128-slice column tile).
"""
import collections
import csv
import json
import sys


def per_kernel_mean(path):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


def main():
    fetch, write = per_kernel_mean(sys.argv[1]), per_kernel_mean(sys.argv[2])
    out = {"units": "bytes per launch; FETCH_SIZE KiB x 1024 x 2 (gfx950 streamed-read correction) + WRITE_SIZE KiB x 1024"}
    rows = []
    for k in sorted(fetch, key=lambda k: -fetch[k])[:12]:
        f, w = fetch[k] * 1024.0, write.get(k, 0.0) * 1024.0
        rows.append({"kernel": k[:80], "fetch_raw_bytes": f, "write_bytes": w, "hbm_bytes_corrected": 2 * f + w})
        for name in ("k_embed_reg_unit", "k_conv_fused_unit", "k_project", "k_rs_downsweep", "k_rs_upsweep", "k_segscan_chained"):
            if name in k:
                out.setdefault(name + "_hbm_bytes_per_launch", 2 * f + w)
    out["kernels"] = rows
    dst = sys.argv[3] if len(sys.argv) > 3 else "profiles/pmc_traffic_latest.json"
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "kernels"}))


if __name__ == "__main__":
    main()
