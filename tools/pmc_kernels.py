#!/usr/bin/env python
"""Per-kernel counter table from one or more `rocprofv3 --kernel-trace --pmc ... --output-format csv` passes over the same
command (SQ has 8 counter slots per pass, FETCH_SIZE takes 3 of the 4 TCC slots: several passes, merged here by kernel):

    python tools/pmc_kernels.py "<note>" <pass dir> [<pass dir> ...] > profiles/r04_denoise_pmc_kernels.json

Rows = (kernel instantiation, grid size) of the engine's own kernels (torch's weight-generation kernels are left out), sorted by
total time.  Per row: launches, average duration in the profiled run (counter collection serialises launches and lowers the clock:
durations are for ranking, not for quoting), and per-launch averages of every counter plus
  mfma_busy     = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)     matrix-pipe occupancy over the launch
  wait_lds      = SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES      wave-cycles stalled issuing LDS instructions
  wait_any      = SQ_WAIT_ANY / SQ_WAVE_CYCLES           wave-cycles parked at s_waitcnt / s_barrier
  wait_inst_any = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES      issue stalls (MFMA read-after-write, pipe busy, ...)
  active        = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES
  lds_conflict  = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  fetch_MB      = 2 x FETCH_SIZE (KiB) / 1024: MI355X_MICROARCH.md, gfx950 counts 128-byte requests at 64 bytes"""
import collections
import csv
import glob
import json
import re
import sys


def short(name):
    n = name.replace("void ", "").replace("(anonymous namespace)::", "")
    n = re.sub(r"TileCfg<2, 2, 2, 2, 2(, 1)?>", "CfgB", n)
    n = re.sub(r"TileCfg<4, 2, 2, 2, 3(, 1)?>", "CfgC", n)
    n = re.sub(r"TileCfg<2, 2, 2, 1, 3, 2>", "CfgK", n)
    n = re.sub(r"\(.*$", "", n)
    return n.replace("(EmuEpilogue)", "").replace(" ", "")


def main():
    note, dirs = sys.argv[1], sys.argv[2:]
    acc = collections.defaultdict(lambda: collections.defaultdict(float))      # key -> counter -> sum
    cnt = collections.defaultdict(lambda: collections.defaultdict(int))        # key -> counter -> launches seen
    dur = collections.defaultdict(float)
    nl = collections.defaultdict(int)
    for d in dirs:
        seen = set()
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                kn = row["Kernel_Name"]
                if "at::native" in kn or "rocclr" in kn or "at::cuda" in kn:
                    continue
                key = (short(kn), int(row["Grid_Size"]))
                c = row["Counter_Name"]
                acc[key][c] += float(row["Counter_Value"])
                cnt[key][c] += 1
                disp = (f, row["Dispatch_Id"])
                if d == dirs[0] and disp not in seen:
                    seen.add(disp)
                    dur[key] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
                    nl[key] += 1
    rows = []
    for key in acc:
        a = {c: acc[key][c] / max(1, cnt[key][c]) for c in acc[key]}
        g = a.get("GRBM_GUI_ACTIVE", 0.0)
        wc = a.get("SQ_WAVE_CYCLES", 0.0)
        r = {"kernel": key[0], "grid": key[1], "launches": nl[key], "avg_us_profiled": dur[key] / max(1, nl[key]),
             "total_ms_profiled": dur[key] / 1e3}
        if g and "SQ_VALU_MFMA_BUSY_CYCLES" in a:
            r["mfma_busy"] = a["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (g / 8)
        for nm, c in (("wait_lds", "SQ_WAIT_INST_LDS"), ("wait_any", "SQ_WAIT_ANY"), ("wait_inst_any", "SQ_WAIT_INST_ANY"),
                      ("active", "SQ_ACTIVE_INST_ANY")):
            if wc and c in a:
                r[nm] = a[c] / wc
        if a.get("SQ_LDS_IDX_ACTIVE") and "SQ_LDS_BANK_CONFLICT" in a:
            r["lds_conflict"] = a["SQ_LDS_BANK_CONFLICT"] / a["SQ_LDS_IDX_ACTIVE"]
        if "FETCH_SIZE" in a:
            r["fetch_MB"] = 2.0 * a["FETCH_SIZE"] / 1024.0
        if "WRITE_SIZE" in a:
            r["write_MB_uncalibrated"] = a["WRITE_SIZE"] / 1024.0
        r["counters_per_launch"] = {c: round(v, 1) for c, v in sorted(a.items())}
        rows.append(r)
    rows.sort(key=lambda r: -r["total_ms_profiled"])
    tot = sum(r["total_ms_profiled"] for r in rows)
    print(json.dumps({"note": note, "passes": dirs, "total_ms_profiled": tot, "kernels": rows}, indent=1))
    # human-readable digest on stderr
    for r in rows[:24]:
        print(f"{r['total_ms_profiled']:8.2f} ms {r['launches']:5d} x {r['avg_us_profiled']:7.1f} us  mfma {r.get('mfma_busy', float('nan')):.3f} "
              f"wait_any {r.get('wait_any', float('nan')):.2f} wait_inst {r.get('wait_inst_any', float('nan')):.2f} wait_lds {r.get('wait_lds', float('nan')):.2f} "
              f"active {r.get('active', float('nan')):.2f} fetch {r.get('fetch_MB', float('nan')):7.1f} MB  {r['kernel'][:90]} grid {r['grid']}", file=sys.stderr)


if __name__ == "__main__":
    main()
