#!/usr/bin/env python
"""HBM traffic of the prefill's MFMA GEMMs from a `rocprofv3 --kernel-trace --pmc FETCH_SIZE` pass over
`python bench.py --pmc-prefill N` (separate pass, kernel trace only, as MI355X_MICROARCH.md prescribes):
    python tools/pmc_gemm_traffic.py <fetch_dir> <bench_pmc_prefill.json>
FETCH_SIZE on gfx950: KB, and 64 B counted per 128-B request of a wide coalesced stream -> bytes = KB * 1024 * 2."""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import gemm_source_hash  # noqa: E402

acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if row["Counter_Name"] == "FETCH_SIZE" and ("gemm_pp_kernel" in k or "gemm_w4_kernel" in k or "gemm2_kernel" in k or "reduce_kernel" in k):
            acc[k].append(float(row["Counter_Value"]))
bench = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
n = bench["pmc_prefill_calls"]
total = sum(sum(v) for v in acc.values()) * 2048 / n
out = {
    "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python bench.py --pmc-prefill %d" % n,
    "correction": "gfx950: FETCH_SIZE in KB, 64 B counted per 128-B request of a wide coalesced stream -> bytes = KB * 1024 * 2",
    "source_sha256": gemm_source_hash(),
    "S": bench["S"],
    "kernels": [{"kernel": k[:140], "launches_per_prefill": len(v) / n, "hbm_read_bytes_per_launch_corrected": sum(v) / len(v) * 2048}
                for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))],
    "hbm_read_bytes_per_prefill": total,
    "algorithmic_gemm_bytes_per_prefill": bench["gemm_algorithmic_bytes_per_prefill"],
    "traffic_over_algorithmic": total / bench["gemm_algorithmic_bytes_per_prefill"],
}
print(json.dumps(out, indent=1))
