#!/usr/bin/env python
"""Same-run A/B of the UNet denoise step under different launch-fusion masks (emu_unet_set_fusion): one engine, one set of
weights, the variants alternated inside one process (box-to-box variance of the MFMA-bound legs is ~12 %, so only same-run
comparisons are evidence).  A variant is a fusion mask, optionally followed by "t<N>" = emu_gemm_tune(N) (e.g. "3t1"), or
"fp8" = the transformer blocks W8A8 (emu_unet_use_fp8).
Usage: python tools/unet_ab.py [steps] [variant,variant,...] [rounds]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emu_amd import synth
from emu_amd.llama import EmuHipContext
from emu_amd.unet import UNetCfg, UNetEngine, unet_param_shapes

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
masks = sys.argv[2].split(",") if len(sys.argv) > 2 else ["0", "3"]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda", 0)
eng = UNetEngine(UNetCfg(), EmuHipContext(dev))
eng.load_state_dict(synth.iter_synth(unet_param_shapes(UNetCfg()), seed=0, device=dev, dtype=torch.bfloat16))
g = torch.Generator().manual_seed(3)
prompt = torch.randn(2, 64, 1792, generator=g).to(torch.bfloat16).to(dev)
lat0 = (torch.randn(1, 4, 128, 128, generator=torch.Generator().manual_seed(4)) * 13.0).to(torch.bfloat16).to(dev)
best = {}
with torch.no_grad():
    for r in range(rounds):
        for m in masks:
            from emu_amd._lib import lib
            if m.startswith("fp8"):                      # "fp8" = W8A8 blocks with the V^T / cross-attention epilogues, "fp8m0" = without
                lib().emu_gemm_tune(0)
                eng.use_fp8(True)
                got = "fp8 + fusion %d" % eng.set_fusion(int(m[4:]) if m[3:4] == "m" else 7)
            else:
                eng.use_fp8(False) if getattr(eng, "fp8", False) else None
                fm, _, tn = m.partition("t")
                lib().emu_gemm_tune(int(tn or 0))
                got = eng.set_fusion(int(fm))
            eng.set_timesteps(50)
            eng.set_context(prompt, 1024, 1024)
            lat = lat0.clone()
            eng.denoise(lat, 3.0, use_graph=True, steps=3)
            eng.set_timesteps(50)
            lat.copy_(lat0)
            torch.cuda.synchronize(); t = time.perf_counter()
            eng.denoise(lat, 3.0, use_graph=True, steps=steps)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t) / steps * 1e3
            best[m] = min(best.get(m, 1e9), dt)
            print(f"round {r} fusion mask {m} (in effect {got}): {dt:.3f} ms/step  finite {bool(torch.isfinite(lat.float()).all())}", flush=True)
print("best:", {m: round(v, 3) for m, v in best.items()})
