#!/usr/bin/env python
"""Per-CU timeline of ONE persistent engine launch (csrc/decode_engine.hip, EngArgs::dbg bit 1): kernel entry, first / last fill
issued, loader done, activation vector ready, first fill consumed, op 0 done (consumer 0), last consumer done -- microseconds from the
earliest entry, min / median / max over the CUs.  Usage: EMU_ENGINE_DBG=2 python tools/engine_trace.py [tp] [case: qkv|o|gu|down|mlp]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["EMU_ENGINE_DBG"] = str(int(os.environ.get("EMU_ENGINE_DBG", "0")) | 2)
from emu_amd import ops
from emu_amd.llama import EmuHipContext

BF16 = torch.bfloat16
tp = int(sys.argv[1]) if len(sys.argv) > 1 else 8
case = sys.argv[2] if len(sys.argv) > 2 else "gu"
dev = torch.device("cuda", 0)
ctx = EmuHipContext(dev, 0, 1)
H, F = 6656, 17920
hl = -(-52 // tp)
HD, Fl = hl * 128, F // tp
r = lambda *sh, scale=1.0: (torch.randn(*sh, device=dev) * scale).to(BF16)
flush = [r(6656, 6656, scale=0.02) for _ in range(6)]
xf = r(1, 6656)
specs = {"qkv": [dict(w=r(3 * HD, H, scale=0.02), x=r(1, H), gain=r(H), eps=1e-6, epi=ops.EPI_NONE)],
         "o": [dict(w=r(H, HD, scale=0.02), x=r(1, HD), epi=ops.EPI_RESID, res=r(1, H))],
         "gu": [dict(w=r(2 * Fl, H, scale=0.02), x=r(1, H), gain=r(H), eps=1e-6, epi=ops.EPI_SWIGLU)],
         "down": [dict(w=r(H, Fl, scale=0.02), x=r(1, Fl), epi=ops.EPI_RESID, res=r(1, H))]}
specs["mlp"] = [specs["gu"][0], dict(w=specs["down"][0]["w"], x=None, epi=ops.EPI_RESID, res=specs["down"][0]["res"])]
# a whole layer's projections as one chain (the all-reduces as plain hand-offs): o_proj -> gate/up -> down -> the next layer's qkv
specs["layer"] = [specs["o"][0], dict(w=specs["gu"][0]["w"], x=None, gain=specs["gu"][0]["gain"], eps=1e-6, epi=ops.EPI_SWIGLU),
                  dict(w=specs["down"][0]["w"], x=None, epi=ops.EPI_RESID, res=specs["down"][0]["res"]),
                  dict(w=specs["qkv"][0]["w"], x=None, gain=specs["qkv"][0]["gain"], eps=1e-6, epi=ops.EPI_NONE)]
names = ["entry", "first fill issued", "last fill issued", "loader done"] + [f"x ready (op {i})" for i in range(6)] + [f"op {i} done (consumer 0)" for i in range(6)] + [f"op {i} done (last consumer)" for i in range(6)] + [f"op {i} input swept (consumer 0)" for i in range(6)]
for rep in range(3):
    err = torch.zeros(2 + 32 * 512, device=dev, dtype=torch.int64)
    for w in flush:
        ops.linear(xf, w)
    ops.gemv_chain(ctx.handle, specs[case], err=err.view(torch.int32))
    torch.cuda.synchronize()
    t = err[1:1 + 32 * 256].view(256, 32).cpu().double()
    t0 = t[:, 0].min()
    print(f"--- {case} tp{tp} run {rep}: give-ups {int(err.view(torch.int32)[0])}")
    for k, nm in enumerate(names):
        col = (t[:, k] - t0) / 100.0                      # 100 MHz ticks -> us
        col = col[t[:, k] > 0]
        if len(col):
            print(f"  {nm:24s} min {col.min():6.2f}  med {col.median():6.2f}  max {col.max():6.2f} us")
