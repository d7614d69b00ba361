#!/usr/bin/env python
"""MFMA pipe utilisation per kernel class from a `rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE` run:
    python tools/pmc_mfma.py <dir> "<note>"
utilisation = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs), summed over the launches of a class.
Launches of torch's own kernels (synthetic-weight generation at load time: thousands of them, outside every timed region) are left
out of "all" -- round 2's whole-leg figure of 0.126 had them in the denominator, which is most of why it sat below the FLOP
fraction (0.19): over the engine's kernels alone the same counters give 0.16, and FLOP / (busy cycles x 1024) reproduces the
13.5 TFLOP per step."""
import collections
import csv
import glob
import json
import sys


def klass(name):
    if "gemm_pp_kernel" in name:
        return "gemm256 (ping-pong)"
    if "gemm2_kernel" in name:
        return "gemm (128/256x128/128x64 tiles)"
    if "reduce_kernel" in name:
        return "splitk_reduce"
    if "flash_kernel" in name:
        return "flash"
    if "gemv" in name:
        return "gemv"
    return "other"


acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "at::native" in row["Kernel_Name"] or "rocclr" in row["Kernel_Name"]:
            continue                                   # torch kernels of the weight generation: not the leg
        k = klass(row["Kernel_Name"])
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        if row["Counter_Name"] == "GRBM_GUI_ACTIVE":
            n[k] += 1
out = {}
tb = tg = 0.0
for k, c in acc.items():
    b, g = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), c.get("GRBM_GUI_ACTIVE", 0.0)
    out[k] = {"launches": n[k], "SQ_VALU_MFMA_BUSY_CYCLES": b, "GRBM_GUI_ACTIVE": g, "mfma_busy_frac": b / 1024 / (g / 8) if g else 0.0}
    tb += b
    tg += g
out["all"] = {"mfma_busy_frac": tb / 1024 / (tg / 8) if tg else 0.0}
out["note"] = sys.argv[2] if len(sys.argv) > 2 else ""
print(json.dumps(out, indent=1))
