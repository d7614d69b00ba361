#!/usr/bin/env python
"""fp8 x fp8 MFMA GEMM (emu_linear_fp8_bf16) vs the bf16 GEMM at the prefill / UNet shapes: HIP-event timing, random data."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emu_amd import ops  # noqa: E402
from emu_amd._lib import lib  # noqa: E402

BF16 = torch.bfloat16


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


sk = torch.zeros(256 * 288 * 256, dtype=torch.float32, device="cuda")
lib().emu_set_splitk_scratch(sk.data_ptr(), sk.numel() * 4)
print(f"{'shape':44s} {'bf16':>10s} {'fp8':>10s} {'quant(A)':>10s}")
for name, M, N, K, epi in [("prefill qkv", 770, 19968, 6656, 0), ("prefill o", 770, 6656, 6656, 1), ("prefill gateup", 770, 35840, 6656, 2),
                           ("prefill down", 770, 6656, 17920, 1), ("prefill1544 gateup", 1544, 35840, 6656, 2),
                           ("vit fc1", 1025, 15360, 1792, 4), ("unet32 geglu", 2048, 10240, 1280, 5), ("unet64 geglu", 8192, 5120, 640, 5),
                           ("square8192", 8192, 8192, 8192, 0)]:
    x = torch.randn(M, K, device="cuda").to(BF16)
    w = (torch.randn(N, K, device="cuda") * 0.02).to(BF16)
    res = torch.randn(M, N, device="cuda").to(BF16) if epi == 1 else None
    x8, xs = ops.quantize_fp8_rows(x)
    w8, ws = ops.quantize_fp8_rows(w)
    fl = 2.0 * M * N * K
    t16 = timeit(lambda: ops.linear(x, w, res=res, epi=epi))
    t8 = timeit(lambda: ops.linear_fp8(x8, xs, w8, ws, res=res, epi=epi))
    tq = timeit(lambda: ops.quantize_fp8_rows(x))
    print(f"{name + f' M{M} N{N} K{K} e{epi}':44s} {fl / t16 / 1e12:7.0f} TF {fl / t8 / 1e12:7.0f} TF {tq * 1e6:7.1f} us", flush=True)
