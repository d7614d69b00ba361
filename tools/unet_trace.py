#!/usr/bin/env python
"""Per-workgroup timelines of GEMM launches INSIDE a real denoise step (cold weights, the real predecessor's outputs, the clocks of a
sustained run) -- what tools/gemm_trace.py measures in isolation.  Needs the -DEMU_TRACE twin:
    EMU_HIP_TOOLS=1 EMU_HIP_LIB=emu_amd/csrc/libemu_hip_trace.so python tools/unet_trace.py
One eager step records the launch profile (shape of every GEMM launch, in order); then, per interesting shape, the step is run
again with only that launch traced."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emu_amd import synth  # noqa: E402
from emu_amd._lib import lib  # noqa: E402
from emu_amd.llama import EmuHipContext  # noqa: E402
from emu_amd.unet import UNetCfg, UNetEngine, unet_param_shapes  # noqa: E402

L = lib()
assert L.emu_gemm_trace_built() == 1, "load the -DEMU_TRACE twin library"
dev = torch.device("cuda", 0)
eng = UNetEngine(UNetCfg(), EmuHipContext(dev))
eng.load_state_dict(synth.iter_synth(unet_param_shapes(UNetCfg()), seed=0, device=dev, dtype=torch.bfloat16))
prompt = torch.randn(2, 64, 1792, generator=torch.Generator().manual_seed(3)).to(torch.bfloat16).to(dev)
lat0 = (torch.randn(1, 4, 128, 128, generator=torch.Generator().manual_seed(4)) * 13.0).to(torch.bfloat16).to(dev)
buf = torch.zeros(8192 * 8, dtype=torch.int64, device=dev)


def step(n=1):
    eng.set_timesteps(50)
    eng.set_context(prompt, 1024, 1024)
    lat = lat0.clone()
    eng.denoise(lat, 3.0, use_graph=False, steps=n)
    torch.cuda.synchronize()


def pct(t, qs=(0.0, 0.5, 0.9, 1.0)):
    t = t.double().sort().values
    return "/".join(f"{float(t[min(len(t) - 1, int(q * (len(t) - 1) + 0.5))]):.2f}" for q in qs)


with torch.no_grad():
    step(2)                                                        # warm
    # which launch index is what: the trace hook counts GEMM / conv launches in order, and a step's order is fixed.  Index by
    # position inside the first 32^2-level transformer (down block 2): found by tracing a window of launches and reading grid sizes
    # is overkill -- walk a range and print what each traced launch looked like.
    targets = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else [])]
    if not targets:
        targets = list(range(60, 100))                             # a window that covers the first 32^2 transformer block
    for n in targets:
        buf.zero_()
        L.emu_gemm_trace(buf.data_ptr())
        L.emu_gemm_trace_nth(n + 0)
        # set_context's own GEMMs come first in a step() call: count them out by tracing relative to the denoise call only
        eng.set_timesteps(50)
        eng.set_context(prompt, 1024, 1024)
        L.emu_gemm_trace_nth(n)
        lat = lat0.clone()
        eng.denoise(lat, 3.0, use_graph=False, steps=1)
        torch.cuda.synchronize()
        L.emu_gemm_trace(None)
        t = buf.view(-1, 8)
        t = t[t[:, 0] != 0].cpu()
        if t.shape[0] == 0:
            print(f"launch {n}: (not a traced kernel)")
            continue
        t0 = t[:, 0].min()
        tick = 0.01
        print(f"launch {n}: {t.shape[0]} workgroups, span {float(t[:, 3].max() - t0) * tick:.2f} us; entry skew {pct((t[:, 0] - t0) * tick)}; "
              f"first tile {pct((t[:, 1] - t[:, 0]) * tick)}; loop {pct((t[:, 2] - t[:, 1]) * tick)}; epilogue {pct((t[:, 3] - t[:, 2]) * tick)}; "
              f"lifetime {pct((t[:, 3] - t[:, 0]) * tick)} us", flush=True)
