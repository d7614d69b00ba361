#!/usr/bin/env python
"""Time the full-size VAE decode (128x128x4 latents -> 1024x1024 image) on synthetic weights (debug aid)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emu_amd import synth
from emu_amd.llama import EmuHipContext
from emu_amd.vae import VaeCfg, VaeDecoder, vae_decoder_param_shapes

dev = torch.device("cuda", 0)
cfg = VaeCfg()
dec = VaeDecoder(cfg, EmuHipContext(dev))
W = synth.synth_state_dict(vae_decoder_param_shapes(cfg), seed=1)
dec.load_state_dict(W, strict=True)
z = torch.randn(1, cfg.latent_channels, 128, 128, device=dev).to(torch.bfloat16)
for _ in range(2):
    out = dec.decode_latents(z)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(3):
    out = dec.decode_latents(z)
torch.cuda.synchronize()
print("vae decode ms:", (time.perf_counter() - t) / 3 * 1e3, tuple(out.shape))
