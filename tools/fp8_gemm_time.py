#!/usr/bin/env python
"""fp8 x fp8 MFMA GEMM (emu_linear_fp8_bf16) vs the bf16 GEMM at the prefill / ViT / UNet shapes: HIP-event timing, random data.

    python tools/fp8_gemm_time.py [--filter unet] [--cfgs 0,K,B,C,S,P] [--cold-mb 800]

--cfgs pins the fp8 GEMM's tile configuration (emu_gemm_force_config; 0 = the heuristic); --cold-mb cycles every launch through
that many MB of weight copies, which is what a launch inside the model sees (every weight matrix is cold in HBM)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emu_amd import ops  # noqa: E402
from emu_amd._lib import lib  # noqa: E402

BF16 = torch.bfloat16
ap = argparse.ArgumentParser()
ap.add_argument("--filter", default="")
ap.add_argument("--cfgs", default="0")
ap.add_argument("--cold-mb", type=int, default=0)
ap.add_argument("--iters", type=int, default=30)
args = ap.parse_args()
L = lib()


def timeit(fn, iters):
    best = 1e9
    for _ in range(3):
        for i in range(3):
            fn(i)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(iters):
            fn(i)
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / iters * 1e-3)
    return best


sk = torch.zeros(256 * 288 * 256, dtype=torch.float32, device="cuda")
L.emu_set_splitk_scratch(sk.data_ptr(), sk.numel() * 4)
SHAPES = [("prefill qkv", 770, 19968, 6656, 0), ("prefill o", 770, 6656, 6656, 1), ("prefill gateup", 770, 35840, 6656, 2),
          ("prefill down", 770, 6656, 17920, 1), ("prefill1544 gateup", 1544, 35840, 6656, 2),
          ("vit qkv", 1025, 5376, 1792, 0), ("vit proj", 1025, 1792, 1792, 1), ("vit fc1", 1025, 15360, 1792, 4),
          ("vit fc2", 1025, 1792, 15360, 1),
          ("unet32 qkv", 2048, 3840, 1280, 0), ("unet32 out", 2048, 1280, 1280, 1), ("unet32 geglu", 2048, 10240, 1280, 5),
          ("unet32 ffout", 2048, 1280, 5120, 1), ("unet64 qkv", 8192, 1920, 640, 0), ("unet64 out", 8192, 640, 640, 1),
          ("unet64 geglu", 8192, 5120, 640, 5), ("unet64 ffout", 8192, 640, 2560, 1), ("square8192", 8192, 8192, 8192, 0)]
cfgs = args.cfgs.split(",")
print(f"{'shape':42s} {'bf16 us':>8s} " + " ".join(f"{'fp8:' + c:>8s}" for c in cfgs) + f" {'quant(A)':>9s}   (TFLOP/s of the best fp8)")
for name, M, N, K, epi in SHAPES:
    if args.filter and args.filter not in name:
        continue
    x = torch.randn(M, K, device="cuda").to(BF16)
    ncopy = max(1, min(64, (args.cold_mb << 20) // (N * K * 2) + 1)) if args.cold_mb else 1
    ws16 = [(torch.randn(N, K, device="cuda") * 0.02).to(BF16) for _ in range(ncopy)]
    res = torch.randn(M, N, device="cuda").to(BF16) if epi == 1 else None
    x8, xs = ops.quantize_fp8_rows(x)
    ws8 = [ops.quantize_fp8_rows(w) for w in ws16]
    out = torch.empty(M, N // 2 if epi in (2, 5) else N, device="cuda", dtype=BF16)
    fl = 2.0 * M * N * K
    L.emu_gemm_force_config(0)
    t16 = timeit(lambda i: ops.linear(x, ws16[i % ncopy], res=res, epi=epi, out=out), args.iters)
    t8 = []
    for c in cfgs:
        L.emu_gemm_force_config(0 if c == "0" else ord(c))
        t8.append(timeit(lambda i: ops.linear_fp8(x8, xs, ws8[i % ncopy][0], ws8[i % ncopy][1], res=res, epi=epi, out=out), args.iters))
    L.emu_gemm_force_config(0)
    tq = timeit(lambda i: ops.quantize_fp8_rows(x), args.iters)
    print(f"{name + f' {M}x{N}x{K} e{epi}':42s} {t16 * 1e6:8.1f} " + " ".join(f"{t * 1e6:8.1f}" for t in t8) +
          f" {tq * 1e6:9.1f}   {fl / min(t8) / 1e12:6.0f}  (bf16 {fl / t16 / 1e12:5.0f})", flush=True)
    del ws16, ws8
