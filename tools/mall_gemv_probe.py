#!/usr/bin/env python
"""What does a weight-streaming GEMV gain when its matrix was TOUCHED into the infinity cache (default-policy loads,
ops.prefetch) by the launch before it -- and how fast does the touch itself run?  Round 2's tools/mall_probe.py asked the
question with the GEMV's own non-temporal stream as the warming pass (answer: nothing); here the warming pass is the successor
prefetch the decode step carries.  Shapes: the four projections of a TP = 8 shard, of a TP = 1 layer where they fit, lm_head.
Usage: python tools/mall_gemv_probe.py"""
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


def ev():
    return torch.cuda.Event(enable_timing=True)


def med(v):
    return sorted(v)[len(v) // 2]


print("--- the touch alone (cold matrix): us and TB/s by workgroup count", flush=True)
for mb in (12, 30, 60, 89):
    w = torch.empty(mb * 1000 * 1000 // 2, device=dev, dtype=BF16).normal_()
    for wgs in (64, 128, 256, 512, 1024, 2048):
        ts = []
        for _ in range(7):
            flush()
            a, b = ev(), ev()
            a.record(); ops.prefetch(w, wgs); b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 1e3)
        t = med(ts)
        print(f"touch {mb:3d} MB, {wgs:5d} workgroups: {t:7.1f} us = {mb / t:5.2f} TB/s", flush=True)

print("--- GEMV cold / after its own stream (nt) / after a touch", flush=True)
H, F = 6656, 17920
cases = [("tp8 qkv", 3 * 7 * 128, H, True, ops.EPI_NONE), ("tp8 o", H, 7 * 128, False, ops.EPI_NONE),
         ("tp8 gate/up", 2 * (F // 8), H, True, ops.EPI_SWIGLU), ("tp8 down", H, F // 8, False, ops.EPI_NONE),
         ("tp4 gate/up", 2 * (F // 4), H, True, ops.EPI_SWIGLU), ("tp4 down", H, F // 4, False, ops.EPI_NONE),
         ("tp2 o", H, H // 2, False, ops.EPI_NONE),
         ("tp1 o", H, H, False, ops.EPI_NONE), ("tp1 down", H, F, False, ops.EPI_NONE), ("tp2 qkv", 3 * H // 2, H, True, ops.EPI_NONE)]
for name, N, K, norm, epi in cases:
    w = (torch.randn(N, K, device=dev) * 0.02).to(BF16)
    x = torch.randn(1, K, device=dev).to(BF16)
    g = torch.ones(K, device=dev, dtype=BF16) if norm else None
    out = torch.empty(1, N // 2 if epi == ops.EPI_SWIGLU else N, device=dev, dtype=BF16)
    run = lambda: ops.linear(x, w, norm_w=g, eps=1e-6, epi=epi, out=out)
    cold, own, hot = [], [], []
    for _ in range(9):
        a, b, c = ev(), ev(), ev()
        flush()
        a.record(); run(); b.record(); run(); c.record()
        torch.cuda.synchronize()
        cold.append(a.elapsed_time(b) * 1e3); own.append(b.elapsed_time(c) * 1e3)
        flush()
        ops.prefetch(w, 1024)
        a.record(); run(); b.record()
        torch.cuda.synchronize()
        hot.append(a.elapsed_time(b) * 1e3)
    mb = N * K * 2 / 1e6
    c_, o_, h_ = med(cold), med(own), med(hot)
    print(f"{name:12s} {mb:7.1f} MB  cold {c_:6.1f} us = {mb / c_:5.2f} TB/s | own stream before {o_:6.1f} us = {mb / o_:5.2f} | touched {h_:6.1f} us = {mb / h_:5.2f} TB/s",
          flush=True)
