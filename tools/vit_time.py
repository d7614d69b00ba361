#!/usr/bin/env python
"""Time the EVA-CLIP ViT encode alone (synthetic weights of the true shape, 1 x 448 x 448): python tools/vit_time.py [reps] [--fp8]
(--fp8: the blocks' GEMMs W8A8, emu_vit_use_fp8)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emu_amd import CLIPVisionCfg, synth
from emu_amd.llama import EmuHipContext
from emu_amd.vit import VitEngine

fp8 = "--fp8" in sys.argv
graph = "--graph" in sys.argv                    # replay the encode from a hipGraph (what VitEngine.forward does for repeated shapes)
args = [x for x in sys.argv[1:] if x not in ("--fp8", "--graph")]
reps = int(args[0]) if args else 8
dev = torch.device("cuda", 0)
v = CLIPVisionCfg(n_query=256, v_query=64)
eng = VitEngine(v, EmuHipContext(dev))
eng.load_weights(synth.iter_synth(synth.vit_param_shapes(v), seed=0, device=dev, dtype=torch.bfloat16))
img = torch.randn(1, 3, v.image_size, v.image_size, device=dev)
if fp8:
    eng.use_fp8(True)
if os.environ.get("EMU_VIT_FUSION"):                 # A/B: emu_vit_set_fusion mask (0 = the launch sequence of rounds 1-3, 3 = default)
    eng.set_fusion(int(os.environ["EMU_VIT_FUSION"]))
batch = int(os.environ.get("EMU_VIT_BATCH", "1"))        # images per encode (configs[2]: 4)
if batch > 1:
    img = img.expand(batch, -1, -1, -1).contiguous()
from emu_amd._lib import lib
for tune in [int(x) for x in os.environ.get("EMU_TUNES", "0").split(",")]:        # same-run A/B of emu_gemm_tune masks (65536 = no prefetch)
    lib().emu_gemm_tune(tune)
    ts = []
    with torch.no_grad():
        run = lambda: eng.forward(img)
        if graph:
            eng.forward(img); torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = eng.forward(img)
            run = g.replay
        for i in range(reps + 2):
            torch.cuda.synchronize(); t = time.perf_counter()
            run()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t) * 1e3)
    print(f"vit encode{' (W8A8 blocks)' if fp8 else ''}{' (hipGraph replay)' if graph else ''}: min {min(ts[2:]):.2f} ms  median {sorted(ts[2:])[len(ts[2:]) // 2]:.2f} ms  "
          f"[fusion mask {os.environ.get('EMU_VIT_FUSION', '3')}, batch {batch}, tune {tune}]", flush=True)
