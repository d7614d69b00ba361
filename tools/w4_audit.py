#!/usr/bin/env python
"""Audit of emu_amd/csrc/gemm_w4.hip after every edit: the kernels hold 256 accumulator registers per lane in AGPRs behind asm
MFMAs with class constraints.  Fails on: a v_accvgpr_* move or a scratch access inside a basic block that carries the 32x32x16
MFMAs (the main loops), or a VALU write of an MFMA's A / B register within the two instructions ahead of an asm MFMA (hipcc pads
wait states for its own MFMAs only).  Reports the spilled registers per kernel.  Compiles the file to assembly (about 3 minutes).

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


def regs(tok):
    """registers named by an operand: v12 -> {12}, v[4:7] -> {4..7}"""
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def mfma_hazards(text):
    """A VALU write of a VGPR needs wait states before an MFMA reads it as A / B; hipcc pads its own MFMAs, not an asm statement's.
    Report every asm MFMA whose A / B registers are written by a VALU instruction in the two instructions ahead of it."""
    found = 0
    ins = []            # (line, text) of real instructions
    for i, l in enumerate(text):
        t = l.split(";")[0].strip()
        if t and not t.startswith(".") and not t.endswith(":"):
            ins.append((i, t))
    for k, (i, t) in enumerate(ins):
        if not t.startswith("v_mfma"):
            continue
        ops = [o.strip() for o in t.split(None, 1)[1].split(",")]
        src = regs(ops[1]) | regs(ops[2])
        for back in (1, 2):
            if k - back < 0:
                break
            pt = ins[k - back][1]
            if pt.startswith("v_") and not pt.startswith("v_mfma") and not pt.startswith("v_cmp"):
                dst = regs(pt.split(None, 1)[1].split(",")[0].strip())
                if dst & src:
                    print(f"line {i}: {pt}  ->  {t}")
                    found += 1
    return found



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
bad += mfma_hazards(text)
# spills outside the MFMA blocks are reported, not counted: a handful of long-lived lane constants parked across the loop cost two
# scratch accesses per kernel; the kernels the dispatch takes (light epilogues) have 0-51 of them, the folded-LayerNorm forms ~200
joined = "\n".join(text)
for m in re.finditer(r"\.name:\s+(\S*gemm_w4_kernel\S*)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_spill_count:\s+(\d+)", joined):
    mm = re.search(r"kernelILi(\d)ELb(\d)ELi(\d+)E", m.group(1))
    print(f"EPI {mm.group(1)} CONV {mm.group(2)} FX {mm.group(3):>2}: scratch {m.group(2):>4} B, {m.group(3):>3} spilled registers (outside the MFMA blocks)")
print(f"{kernels} kernels audited, {bad} findings")
sys.exit(1 if bad else 0)
