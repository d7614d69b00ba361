#!/usr/bin/env python
"""Average the counters of a rocprofv3 --pmc run per kernel: python tools/pmc_summary.py <dir> [kernel-substring]"""
import collections
import csv
import glob
import json
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
sub = sys.argv[2] if len(sys.argv) > 2 else ""
out = {}
for k, cs in acc.items():
    if sub in k:
        out[k[:120]] = {c: sum(v) / len(v) for c, v in cs.items()}
        out[k[:120]]["dispatches"] = len(next(iter(cs.values())))
print(json.dumps(out, indent=1))
