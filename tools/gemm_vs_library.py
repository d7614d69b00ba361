#!/usr/bin/env python
"""This repo's bf16 GEMM kernels beside the vendor library (torch.nn.functional.linear -> hipBLASLt / rocBLAS) on the hot path's own
shapes, same process, same operands, HIP events: a yardstick next to the roofline fraction -- these shapes are skinny (M = 770 ... 2048
rows, K as short as 1280), and how far a hand-written tile is from the peak says little without what the tuned library reaches
on the same problem.  The library runs the PLAIN product; ours is timed plain and with the epilogue the path fuses into it
(residual / SwiGLU / GELU / GEGLU), which the library path would pay as extra launches.  Headroom reference only: nothing in
the product calls the library.  python tools/gemm_vs_library.py [--iters 30] [--filter unet]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emu_amd import ops  # noqa: E402
from emu_amd._lib import lib  # noqa: E402

BF16 = torch.bfloat16
ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--filter", default="")
args = ap.parse_args()
sk = torch.zeros(512 * 288 * 256, dtype=torch.float32, device="cuda")
lib().emu_set_splitk_scratch(sk.data_ptr(), sk.numel() * 4)


def timeit(fn, iters):
    best = 1e9
    for _ in range(3):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / iters * 1e-3)
    return best


# name, M, N, K, fused epilogue on the path (emu_hip.h EPI_*: 0 none, 1 residual, 2 SwiGLU, 4 GELU, 5 GEGLU)
SHAPES = [("prefill S=770 qkv", 770, 19968, 6656, 0), ("prefill S=770 o_proj", 770, 6656, 6656, 1),
          ("prefill S=770 gate/up", 770, 35840, 6656, 2), ("prefill S=770 down", 770, 6656, 17920, 1),
          ("prefill S=1544 gate/up", 1544, 35840, 6656, 2), ("prefill S=1544 down", 1544, 6656, 17920, 1),
          ("vit qkv", 1025, 5376, 1792, 0), ("vit proj", 1025, 1792, 1792, 1), ("vit fc1", 1025, 15360, 1792, 4), ("vit fc2", 1025, 1792, 15360, 1),
          ("vit x4 fc1", 4100, 15360, 1792, 4), ("vit x4 fc2", 4100, 1792, 15360, 1),
          ("unet32 qkv", 2048, 3840, 1280, 0), ("unet32 out-proj", 2048, 1280, 1280, 1), ("unet32 geglu", 2048, 10240, 1280, 5),
          ("unet32 ff-out", 2048, 1280, 5120, 1), ("unet64 out-proj", 8192, 640, 640, 1), ("unet64 geglu", 8192, 5120, 640, 5),
          ("square 4096", 4096, 4096, 4096, 0), ("square 8192", 8192, 8192, 8192, 0)]
print(f"{'shape':40s} {'library us':>10s} {'ours us':>9s} {'ours+epi us':>11s}   TFLOP/s library / ours / ours+epi   ours+epi vs library")
for name, M, N, K, epi in SHAPES:
    if args.filter and args.filter not in name:
        continue
    x = torch.randn(M, K, device="cuda").to(BF16)
    w = (torch.randn(N, K, device="cuda") * 0.02).to(BF16)
    res = torch.randn(M, N, device="cuda").to(BF16) if epi == 1 else None
    out0 = torch.empty(M, N, device="cuda", dtype=BF16)
    oute = torch.empty(M, N // 2 if epi in (2, 5) else N, device="cuda", dtype=BF16)
    fl = 2.0 * M * N * K
    t_lib = timeit(lambda: torch.nn.functional.linear(x, w), args.iters)
    t_0 = timeit(lambda: ops.linear(x, w, out=out0), args.iters)
    t_e = timeit(lambda: ops.linear(x, w, res=res, epi=epi, out=oute), args.iters) if epi else t_0
    print(f"{name + f' {M}x{N}x{K}':40s} {t_lib * 1e6:10.1f} {t_0 * 1e6:9.1f} {t_e * 1e6:11.1f}   "
          f"{fl / t_lib / 1e12:7.0f} / {fl / t_0 / 1e12:5.0f} / {fl / t_e / 1e12:5.0f}            {t_lib / t_e:5.2f}x", flush=True)
    del x, w, res, out0, oute
