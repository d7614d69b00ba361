#!/usr/bin/env python
"""A/B aid: few-row (2..16) linears of the LLaMA-33B decoder under emu_gemm_tune settings: 0 = the shipped dispatch (4..16 rows
on the LDS-DMA + MFMA stream of gemv_thin.hip), 4 = that stream switched off (v_dot2c block kernel / register-fed MFMA, the
dispatch before round 3), 256 * v = variant v of the stream.  Prints HIP-event times over rotating weight copies and the largest
relative difference to the first setting's output.
Usage: python tools/thin_ab.py [tunes, default 4,0] [rows, default 2,3,4,5,8,12,16]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emu_amd import ops
from emu_amd._lib import lib

tunes = [int(t) for t in (sys.argv[1] if len(sys.argv) > 1 else "4,0").split(",")]
rows = [int(t) for t in (sys.argv[2] if len(sys.argv) > 2 else "2,3,4,5,8,12,16").split(",")]
L = lib()
sk = torch.zeros(64 << 20, dtype=torch.float32, device="cuda")
L.emu_set_splitk_scratch(sk.data_ptr(), sk.numel() * 4)
g = torch.Generator(device="cuda").manual_seed(0)
r = lambda *sh, scale=1.0: (torch.randn(*sh, device="cuda", generator=g) * scale).to(torch.bfloat16)
shapes = [("qkv", 19968, 6656, 0), ("o", 6656, 6656, 1), ("gateup", 35840, 6656, 2), ("down", 6656, 17920, 1)]
print("rows shape   " + "".join(f"  tune {t:<5d} us  TB/s " for t in tunes) + "  maxdiff vs first")
for M in rows:
    for name, N, K, epi in shapes:
        nc = max(1, int(600e6 // (N * K * 2)) + 1)
        ws = [r(N, K, scale=0.02) for _ in range(nc)]
        x = r(M, K)
        res = r(M, N) if epi == 1 else None
        outs, times = [], []
        for tune in tunes:
            L.emu_gemm_tune(tune)
            outs.append(ops.linear(x, ws[0], res=res, epi=epi).float())
            for _ in range(3):
                ops.linear(x, ws[0], res=res, epi=epi)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for i in range(30):
                ops.linear(x, ws[i % nc], res=res, epi=epi)
            b.record(); torch.cuda.synchronize()
            times.append(a.elapsed_time(b) / 30 * 1e3)
        L.emu_gemm_tune(0)
        err = max(float((outs[0] - o).abs().max() / outs[0].abs().max()) for o in outs)
        print(f"M{M:2d} {name:7s} " + "".join(f"  {t:10.1f} {2.0*N*K/t/1e6:6.2f} " for t in times) + f"  {err:.4f}", flush=True)
        del ws
