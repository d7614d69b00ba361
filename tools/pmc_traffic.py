#!/usr/bin/env python
"""Combine the two rocprofv3 --pmc passes of tools/pmc_traffic.sh with bench.py's own byte count:
    python tools/pmc_traffic.py <fetch_dir> <write_dir> <bench_pmc_mode.json>
FETCH_SIZE on gfx950: KB, and 64 B counted per 128-B request of a wide coalesced stream -> bytes = KB * 1024 * 2
(MI355X_MICROARCH.md, HBM section).  WRITE_SIZE: KB, uncalibrated."""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import gemv_source_hash  # noqa: E402


def per_kernel(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter and "gemv" in row["Kernel_Name"]:
                acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return acc


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
bench = json.loads([l for l in open(sys.argv[3]) if l.startswith("{")][-1])
total_fetch = sum(sum(v) for v in fetch.values()) * 1024 * 2
launches = sum(len(v) for v in fetch.values())
out = {
    "command": "tools/pmc_traffic.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE (WRITE_SIZE in a separate pass) -- python bench.py --pmc-mode 6",
    "correction": "gfx950: FETCH_SIZE in KB, 64 B counted per 128-B request of a wide coalesced stream -> bytes = KB * 1024 * 2; WRITE_SIZE in KB, uncorrected",
    "source_sha256": gemv_source_hash(),
    "kernels": [{"kernel": k, "launches_in_trace": len(v), "FETCH_SIZE_KB_avg": sum(v) / len(v),
                 "hbm_read_bytes_per_launch_corrected": sum(v) / len(v) * 2048,
                 "WRITE_SIZE_KB_avg": (sum(write[k]) / len(write[k])) if k in write else None} for k, v in sorted(fetch.items())],
    "bench": bench,
    "all_gemv_launches": {"launches_in_trace": launches, "launches_counted_by_bench": bench["gemv_launches"],
                          "hbm_read_bytes": total_fetch, "algorithmic_bytes": bench["gemv_algorithmic_bytes"],
                          "note": "the trace also holds the one logits-row GEMV of the prefill (0.43 GB), counted on both sides only "
                                  "when bench.py's hook was already on; ratio uses the decode-step launches' share",
                          "traffic_over_algorithmic": None},
}
# the prefill's single lm_head row is launched before the hook is switched on: remove its (known) bytes from the PMC side
extra = launches - bench["gemv_launches"]
lm_head = 32274 * 6656 * 2
out["all_gemv_launches"]["traffic_over_algorithmic"] = (total_fetch - extra * lm_head) / bench["gemv_algorithmic_bytes"]
print(json.dumps(out, indent=1))
