#!/usr/bin/env python
"""Isolate: W8A8 UNet step under hipGraph replay gives NaN at a 32x32 latent (eager is fine)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from emu_amd import synth
from emu_amd.llama import EmuHipContext
from emu_amd.unet import UNetCfg, UNetEngine, unet_param_shapes
BF16 = torch.bfloat16
dev = torch.device("cuda", 0)
cfg = UNetCfg()
eng = UNetEngine(cfg, EmuHipContext(dev))
eng.load_state_dict(synth.iter_synth(unet_param_shapes(cfg), seed=0, device=dev, dtype=BF16))
g = torch.Generator().manual_seed(3)
prompt = torch.randn(2, 64, 1792, generator=g).to(BF16).to(dev)
for H in (32, 64, 128):
    for mode in ("bf16 mask0", "fp8"):
        for nsteps in (2, 4):
            sch = eng.set_timesteps(nsteps)
            eng.set_context(prompt, 8 * H, 8 * H)
            lat0 = (torch.randn(1, 4, H, H, generator=g) * sch.init_noise_sigma).to(BF16).to(dev)
            if mode == "fp8":
                eng.use_fp8(True)
            else:
                eng.use_fp8(False); eng.set_fusion(0)
            a = eng.denoise(lat0.clone(), 3.0, use_graph=False).clone()
            eng.set_timesteps(nsteps)
            b = eng.denoise(lat0.clone(), 3.0, use_graph=True).clone()
            eng.set_timesteps(nsteps)
            c = eng.denoise(lat0.clone(), 3.0, use_graph=True).clone()
            print(f"H={H} {mode} steps={nsteps}: eager finite {bool(torch.isfinite(a.float()).all())} graph finite "
                  f"{bool(torch.isfinite(b.float()).all())} equal {torch.equal(a, b)} second graph call equal {torch.equal(a, c)}", flush=True)
eng.use_fp8(False); eng.set_fusion(7)
