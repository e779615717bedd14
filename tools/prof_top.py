"""Prints the top kernels of a rocprofv3 --kernel-trace --stats run:  python tools/prof_top.py <dir> [N] [calls-divisor]"""
import csv, glob, sys
d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 15
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("file", f, "total ms", tot / 1e6)
for r in rows[:n]:
    print("%-78s calls %5s  total %9.3f ms  avg %9.3f ms  %5.1f%%" % (r["Name"][:78], r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                                  float(r["AverageNs"]) / 1e6, float(r["Percentage"])))
