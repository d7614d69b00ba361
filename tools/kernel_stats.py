#!/usr/bin/env python
"""Top kernels of a rocprofv3 --kernel-trace --stats run as CSV on stdout: python tools/kernel_stats.py <dir> [top]"""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
w = csv.writer(sys.stdout)
w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
for r in rows[:top]:
    w.writerow([r["Name"][:160], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
