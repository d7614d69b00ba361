#!/usr/bin/env python
"""Do two denoise loops overlap on one GPU?  Two independent UNet engines (own weights, workspaces, hipGraphs), one image each,
replayed (a) back to back on one stream, (b) alternately on two streams.  If (b) is much faster than (a), the ~6 us launch floors
of the ~1130 small launches per step are hideable by concurrency (e.g. the two CFG halves of ONE image on two streams).
Usage: python tools/unet_2stream.py [steps]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emu_amd import synth
from emu_amd.llama import EmuHipContext
from emu_amd.unet import UNetCfg, UNetEngine, unet_param_shapes

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda", 0)
ctx = EmuHipContext(dev)
cfg = UNetCfg()
engs, lats, graphs = [], [], []
for i in range(2):
    e = UNetEngine(cfg, ctx)
    e.load_state_dict(synth.iter_synth(unet_param_shapes(cfg), seed=i, device=dev, dtype=torch.bfloat16))
    e.set_timesteps(50)
    e.set_context(torch.randn(2, 64, 1792).to(torch.bfloat16).to(dev), 1024, 1024)
    lat = (torch.randn(1, 4, 128, 128) * 13.0).to(torch.bfloat16).to(dev)
    with torch.no_grad():
        e.denoise(lat, 3.0, use_graph=True, steps=3)
    engs.append(e); lats.append(lat); graphs.append(e._graph[1])
torch.cuda.synchronize()


def run(two_streams):
    sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(steps):
        with torch.cuda.stream(sa):
            graphs[0].replay()
        with torch.cuda.stream(sb if two_streams else sa):
            graphs[1].replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / steps * 1e3


for _ in range(2):
    a = run(False)
    b = run(True)
    print(f"two images, one stream: {a:.2f} ms per pair of steps   two streams: {b:.2f} ms   speed-up {a / b:.3f}", flush=True)
