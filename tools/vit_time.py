#!/usr/bin/env python
"""Time the EVA-CLIP ViT encode alone (synthetic weights of the true shape, 1 x 448 x 448): python tools/vit_time.py [reps]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emu_amd import CLIPVisionCfg, synth
from emu_amd.llama import EmuHipContext
from emu_amd.vit import VitEngine

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda", 0)
v = CLIPVisionCfg(n_query=256, v_query=64)
eng = VitEngine(v, EmuHipContext(dev))
eng.load_weights(synth.iter_synth(synth.vit_param_shapes(v), seed=0, device=dev, dtype=torch.bfloat16))
img = torch.randn(1, 3, v.image_size, v.image_size, device=dev)
ts = []
with torch.no_grad():
    for i in range(reps + 2):
        torch.cuda.synchronize(); t = time.perf_counter()
        eng.forward(img)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t) * 1e3)
print(f"vit encode: min {min(ts[2:]):.2f} ms  median {sorted(ts[2:])[len(ts[2:]) // 2]:.2f} ms  [{os.environ.get('EMU_TMP_FORCE', '')}]", flush=True)
