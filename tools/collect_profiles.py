"""Copies the summaries of gpurun_out/refresh_TAG/ (tools/refresh_profiles.sh) into profiles/ as rNN_TAG_*:

    python tools/collect_profiles.py TAG [prefix]     e.g.  python tools/collect_profiles.py v5 r01
"""
import glob
import os
import shutil
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
prefix = sys.argv[2] if len(sys.argv) > 2 else "r01"
src = os.path.join(root, "gpurun_out", "refresh_" + tag)
dst = os.path.join(root, "profiles")
name = "%s_%s" % (prefix, tag)
shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, name + "_bench.json"))
stats = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)
shutil.copy(stats[0], os.path.join(dst, name + "_bench_kernel_stats.csv"))
fetch = glob.glob(os.path.join(src, "pmc_fetch", "**", "*counter_collection.csv"), recursive=True)[0]
write = glob.glob(os.path.join(src, "pmc_write", "**", "*counter_collection.csv"), recursive=True)[0]
subprocess.check_call([sys.executable, os.path.join(root, "tools", "pmc_summary.py"), fetch, write,
                       os.path.join(dst, "pmc_traffic_latest.json")])
shutil.copy(os.path.join(dst, "pmc_traffic_latest.json"), os.path.join(dst, name + "_pmc_traffic.json"))
print("profiles/%s_* written" % name)
