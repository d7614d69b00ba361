#!/usr/bin/env python
"""Per-kernel register / LDS / spill table of one .hip source (hipcc -Rpass-analysis=kernel-resource-usage):
    python tools/resusage.py emu_amd/csrc/gemm.hip [filter]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function", "-c", src,
                    "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + sys.argv[3:], capture_output=True, text=True)
if r.returncode:
    print(r.stderr[-4000:])
    sys.exit(1)
cur = None
rows = []
for line in r.stderr.splitlines():
    m = re.search(r"remark:\s+(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)", line)
    if not m:
        continue
    k, v = m.groups()
    if k == "Function Name":
        cur = {"name": v}
        rows.append(cur)
    elif cur is not None:
        cur[k] = v
names = subprocess.run(["/usr/bin/c++filt"] + [x["name"] for x in rows], capture_output=True, text=True).stdout.splitlines()
for x, d in zip(rows, names):
    d = d.replace("(anonymous namespace)::", "").replace("(EmuEpilogue)", "")
    d = re.sub(r"TileCfg<2, 2, 2, 2, 2(, 1)?>", "CfgB", d)
    d = re.sub(r"TileCfg<4, 2, 2, 2, 3(, 1)?>", "CfgC", d)
    d = re.sub(r"TileCfg<2, 2, 2, 1, 3, 2>", "CfgK", d)
    d = re.sub(r"\(.*$", "", d).replace("void ", "")
    if flt and flt not in d:
        continue
    print(f"{d[:64]:64s} vgpr {x.get('VGPRs','?'):>3} agpr {x.get('AGPRs','?'):>3} scratch {x.get('ScratchSize [bytes/lane]','?'):>4} spill {x.get('VGPRs Spill','?'):>3} "
          f"sgpr {x.get('TotalSGPRs','?'):>3} occ {x.get('Occupancy [waves/SIMD]','?')} lds {x.get('LDS Size [bytes/block]','?')}")
