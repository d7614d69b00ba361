#!/usr/bin/env python
"""Audit of emu_amd/csrc/gemm_w4.hip after every edit: the kernels hold 256 accumulator registers per lane in AGPRs behind asm
MFMAs with class constraints; what must not happen is hipcc moving them -- no v_accvgpr_* and no scratch access inside the main
loops (the basic blocks that carry the 32x32x16 MFMAs), no spills anywhere.  Compiles the file to assembly (about 3 minutes).

    python tools/w4_audit.py
"""
import os
import re
import subprocess
import sys
import tempfile

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "emu_amd", "csrc", "gemm_w4.hip")
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "gemm_w4.s")
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only",
                        "-I" + os.path.dirname(src), src, "-o", out], capture_output=True, text=True)
    if r.returncode:
        print(r.stderr[-3000:])
        sys.exit(1)
    text = open(out).read().split("\n")
bad = 0
kernels = 0
cur = None
block = []          # lines of the current basic block


def close(block, cur):
    n_mfma = sum("v_mfma_f32_32x32x16_bf16" in l for l in block)
    if n_mfma < 16:
        return 0
    moves = [l for l in block if re.search(r"v_accvgpr_(read|write|mov)|scratch_(load|store)", l)]
    for l in moves[:4]:
        print(f"{cur}: in a main-loop block ({n_mfma} MFMAs): {l.strip()}")
    return len(moves)


for l in text:
    m = re.match(r"^(_ZN\S*gemm_w4_kernel\S*):", l)
    if m:
        cur, block = m.group(1), []
        kernels += 1
        continue
    if cur is None:
        continue
    if re.match(r"^\.LBB", l) or "s_cbranch" in l or "s_endpgm" in l:
        bad += close(block, cur)
        block = []
        if "s_endpgm" in l:
            cur = None
        continue
    block.append(l)
for l in text:
    m = re.search(r"\.vgpr_spill_count:\s+(\d+)", l)
    if m and int(m.group(1)):
        print("spill count", m.group(1))
        bad += 1
    m = re.search(r"\.private_segment_fixed_size:\s+(\d+)", l)
    if m and int(m.group(1)):
        print("scratch bytes", m.group(1))
        bad += 1
print(f"{kernels} kernels audited, {bad} findings")
sys.exit(1 if bad else 0)
