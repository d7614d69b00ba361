#!/usr/bin/env python
"""Does the 256 MB infinity cache (MALL) serve a weight-streaming GEMV faster than HBM does?  For each weight size: time ONE GEMV
launch (HIP events around it) cold (after ~700 MB of other weights went through) and warm (the same weights streamed by the
launch just before).  Usage: python tools/mall_probe.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emu_amd import ops

BF16 = torch.bfloat16
dev = "cuda"
flush_w = [(torch.randn(6656, 6656, device=dev) * 0.02).to(BF16) for _ in range(8)]      # 8 x 88.6 MB
xf = torch.randn(1, 6656, device=dev).to(BF16)


def flush():
    for w in flush_w:
        ops.linear(xf, w)


def pair(fn):
    """[flush] a [fn] b [fn] c on one busy queue: cold = b - a, warm = c - b"""
    a, b, c = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    flush()
    a.record(); fn(); b.record(); fn(); c.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3, b.elapsed_time(c) * 1e3


for name, N, K in (("o_proj", 6656, 6656), ("qkv", 19968, 6656), ("gate/up", 35840, 6656), ("down", 6656, 17920),
                   ("half o_proj", 3328, 6656), ("2x o_proj", 13312, 6656)):
    w = (torch.randn(N, K, device=dev) * 0.02).to(BF16)
    x = torch.randn(1, K, device=dev).to(BF16)
    out = torch.empty(1, N, device=dev, dtype=BF16)
    cold, warm = [], []
    for _ in range(10):
        c, h = pair(lambda: ops.linear(x, w, out=out))
        cold.append(c); warm.append(h)
    mb = N * K * 2 / 1e6
    c, h = sorted(cold)[len(cold) // 2], sorted(warm)[len(warm) // 2]
    print(f"{name:12s} {mb:7.1f} MB  cold {c:7.1f} us = {mb / c:6.2f} TB/s   warm {h:7.1f} us = {mb / h:6.2f} TB/s", flush=True)
