"""Prints the top kernels of a rocprofv3 run (rocpd .db output):  python tools/prof_top.py <results.db> [n]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
for name, calls, total, avg, pct in db.execute("select name, total_calls, total_duration, average, percentage from top_kernels limit %d" % n):
    print("%-74s calls %4d  avg %10.1f us  %5.1f %%" % (name[:74], calls, avg, pct))
