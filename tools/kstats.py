#!/usr/bin/env python3
"""Per-kernel totals of a `rocprofv3 --kernel-trace` run (CSV output or the default rocpd SQLite database).
Usage: python tools/kstats.py <rocprof output dir> [top-N]"""
import collections
import csv
import glob
import sqlite3
import sys

d = sys.argv[1]
agg = collections.defaultdict(lambda: [0, 0.0])
csvs = sorted(glob.glob(d + "/**/*_kernel_trace.csv", recursive=True), key=lambda f: __import__("os").path.getmtime(f))[-1:]   # newest run only
dbs = sorted(glob.glob(d + "/**/*_results.db", recursive=True), key=lambda f: __import__("os").path.getmtime(f))[-1:]
if csvs:
    for r in csv.DictReader(open(csvs[0])):
        key = (r["Kernel_Name"].split("(")[0][:44], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])
        agg[key][0] += 1
        agg[key][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
elif dbs:
    con = sqlite3.connect(dbs[0])
    for name, gx, gy, gz, dur in con.execute("select name, grid_x, grid_y, grid_z, duration from kernels"):
        key = (name.split("(")[0][:44], str(gx), str(gy), str(gz))
        agg[key][0] += 1
        agg[key][1] += dur / 1e3
else:
    sys.exit("no kernel trace under " + d)
tot = sum(v[1] for v in agg.values())
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:n]:
    print("%-46s grid %8s %6s %3s calls %4d total %9.1f us avg %9.1f us %5.1f%%"
          % (k[0], k[1], k[2], k[3], v[0], v[1], v[1] / v[0], 100 * v[1] / tot))
print("total GPU ms", tot / 1e3)
