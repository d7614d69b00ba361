#!/usr/bin/env python
"""Does touching a GEMM's weights just BEFORE the launch (from other CUs: the data lands in the 256 MB infinity cache and in some
XCDs' L2) recover what cold weights cost?  Per shape, GEMM time measured with events around the GEMM launch alone:
  warm    : the same weight matrix every launch
  cold    : cycling through > 600 MB of copies (what a forward pass over a multi-GB model sees)
  touched : cold, but a strided read of the matrix (one dword per 128-byte line) runs on the same stream right before the GEMM
  touched-early : the touch of launch i + 1 is issued BEFORE GEMM i (a whole GEMM lies between the touch and its consumer)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emu_amd import ops  # noqa: E402
from emu_amd._lib import lib  # noqa: E402

SHAPES = [("unet32 attn-out", 2048, 1280, 1280, 1), ("unet32 qkv", 2048, 3840, 1280, 0), ("unet32 geglu", 2048, 10240, 1280, 5),
          ("unet32 ff-out", 2048, 1280, 5120, 1), ("unet64 attn-out", 8192, 640, 640, 1), ("unet64 geglu", 8192, 5120, 640, 5),
          ("vit fc1", 1025, 15360, 1792, 4), ("prefill o", 770, 6656, 6656, 1)]
dev = torch.device("cuda", 0)
L = lib()
sk = torch.zeros(512 * 288 * 256, dtype=torch.float32, device=dev)
L.emu_set_splitk_scratch(sk.data_ptr(), sk.numel() * 4)
g = torch.Generator(device=dev).manual_seed(0)
N_IT = 40


def touch(w):
    return w.view(torch.int32).view(-1)[::32].sum()


print(f"{'case':34s} {'warm':>8s} {'cold':>8s} {'touched':>8s} {'early':>8s}   (us per GEMM launch)")
for name, M, N, K, epi in SHAPES:
    x = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
    w0 = (torch.randn(N, K, device=dev, generator=g) * K ** -0.5).to(torch.bfloat16)
    ws = [w0] + [w0.clone() for _ in range(max(2, 700_000_000 // (w0.numel() * 2)))]
    bias = torch.randn(N, device=dev, generator=g).to(torch.bfloat16) if epi in (1, 4, 5) else None
    res = torch.randn(M, N, device=dev, generator=g).to(torch.bfloat16) if epi == 1 else None
    out = torch.empty(M, N // 2 if epi in (2, 5) else N, device=dev, dtype=torch.bfloat16)

    def run(mode):
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(N_IT)]
        for i in range(N_IT + 3):
            w = ws[0] if mode == "warm" else ws[i % len(ws)]
            if mode == "touched":
                touch(w)
            if mode == "early":
                touch(ws[(i + 1) % len(ws)])
            j = i - 3
            if j >= 0:
                ev[j][0].record()
            ops.linear(x, w, bias=bias, res=res, epi=epi, out=out)
            if j >= 0:
                ev[j][1].record()
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
        return ts[len(ts) // 2]
    r = {m: run(m) for m in ("warm", "cold", "touched", "early")}
    print(f"{name + f' {M}x{N}x{K}':34s} {r['warm']:8.1f} {r['cold']:8.1f} {r['touched']:8.1f} {r['early']:8.1f}", flush=True)
