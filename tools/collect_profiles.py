"""Copies the summaries of gpurun_out/refresh_TAG/ (tools/refresh_profiles.sh) into profiles/ as rNN_TAG_*:

    python tools/collect_profiles.py TAG [prefix]     e.g.  python tools/collect_profiles.py v1 r02

The PMC traffic file records where it came from (git HEAD and date of the capture) so that bench.py can say so
("traffic_source") when it quotes the number.
"""
import datetime
import glob
import json
import os
import shutil
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
prefix = sys.argv[2] if len(sys.argv) > 2 else "r02"
src = os.path.join(root, "gpurun_out", "refresh_" + tag)
dst = os.path.join(root, "profiles")
name = "%s_%s" % (prefix, tag)


def stats_of(sub):
    return glob.glob(os.path.join(src, sub, "**", "*kernel_stats.csv"), recursive=True)[0]


shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, name + "_bench.json"))
shutil.copy(stats_of("stats"), os.path.join(dst, name + "_bench_kernel_stats.csv"))
fetch = glob.glob(os.path.join(src, "pmc_fetch", "**", "*counter_collection.csv"), recursive=True)[0]
write = glob.glob(os.path.join(src, "pmc_write", "**", "*counter_collection.csv"), recursive=True)[0]
latest = os.path.join(dst, "pmc_traffic_latest.json")
subprocess.check_call([sys.executable, os.path.join(root, "tools", "pmc_summary.py"), fetch, write, latest])
head = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
t = json.load(open(latest))
t["source"] = {"git_head": head, "captured": datetime.datetime.now().isoformat(timespec="seconds"),
               "how": "tools/refresh_profiles.sh %s: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes of bench.py" % tag}
# the stand-alone segmented scan has its own PMC passes (2.56e8 elements per launch)
try:
    import csv as _csv

    def _mean(path, needle):
        v = [float(r["Counter_Value"]) for r in _csv.DictReader(open(path)) if needle in r["Kernel_Name"]]
        return sum(v) / len(v)
    sf = glob.glob(os.path.join(src, "seg_pmc_fetch", "**", "*counter_collection.csv"), recursive=True)[0]
    sw = glob.glob(os.path.join(src, "seg_pmc_write", "**", "*counter_collection.csv"), recursive=True)[0]
    f_, w_ = _mean(sf, "k_segscan_chained") * 1024.0, _mean(sw, "k_segscan_chained") * 1024.0
    t["k_segscan_chained_hbm_bytes_per_launch"] = 2 * f_ + w_
    t["k_segscan_chained_elements_per_launch"] = 256000000
    t["k_segscan_chained_algorithmic_bytes_per_launch"] = 16.0 * 256000000
except (IndexError, ZeroDivisionError, KeyError):
    print("no segcumsum PMC passes in", src)
json.dump(t, open(latest, "w"), indent=1)
shutil.copy(latest, os.path.join(dst, name + "_pmc_traffic.json"))
for sub, out in (("rmat22", "_rmat22_forward_kernel_stats.csv"), ("segcumsum", "_segcumsum_kernel_stats.csv"),
                 ("train", "_train_step_kernel_stats.csv")):
    try:
        shutil.copy(stats_of(sub), os.path.join(dst, name + out))
    except IndexError:
        print("no", sub, "profile in", src)
for f in ("segcumsum.json", "skew_rmat20.log", "slice_shard_consumer.log", "slice_shard_exchange.log"):
    if os.path.isfile(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, name + "_" + f))
print("profiles/%s_* written" % name)
