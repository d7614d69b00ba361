#!/usr/bin/env python
"""The persistent weight-streaming engine (csrc/decode_engine.hip, emu_gemv_chain_bf16) against the launches it replaces, on the
projections of a tensor-parallel shard's decode layer: bit equality, then time per chain (HIP events, weights rotated through
> 600 MB).  (1) every projection alone; (2) gate/up -> down in one launch (one granule edge) vs two launches.
Usage: python tools/engine_probe.py [tp] [iters]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emu_amd import ops
from emu_amd.llama import EmuHipContext

BF16 = torch.bfloat16
tp = int(sys.argv[1]) if len(sys.argv) > 1 else 8
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device("cuda", 0)
ctx = EmuHipContext(dev, 0, 1)
H, F, heads = 6656, 17920, 52
hl = -(-heads // tp)
HD, Fl = hl * 128, F // tp


def r(*shape, scale=1.0):
    return (torch.randn(*shape, device=dev) * scale).to(BF16)


def timeit(fn, n):
    """us per call: n calls (rotating weights) captured into one hipGraph, the graph replayed 5 times (the host is out of the loop)"""
    for _ in range(2):
        fn(0)
    torch.cuda.synchronize()
    keep = []
    g = torch.cuda.CUDAGraph()
    cap = torch.cuda.Stream(device=dev)
    with torch.cuda.graph(g, stream=cap):
        for i in range(n):
            keep.append(fn(i))
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / (5 * n) * 1e3


cases = [("qkv+norm", 3 * HD, H, ops.EPI_NONE, True), ("o+res", H, HD, ops.EPI_RESID, False),
         ("gateup+norm+swiglu", 2 * Fl, H, ops.EPI_SWIGLU, True), ("down+res", H, Fl, ops.EPI_RESID, False)]
mats = {}
for name, N, K, epi, norm in cases:
    ncopy = max(2, int(600e6 // (N * K * 2)) + 1)
    ws = [r(N, K, scale=0.02) for _ in range(ncopy)]
    x = r(1, K)
    g = r(K) if norm else None
    res = r(1, N) if epi == ops.EPI_RESID else None
    mats[name] = (ws, x, g, res, epi)
    want = ops.linear(x, ws[0], res=res, norm_w=g, eps=1e-6, epi=epi)
    outs, err, _ = ops.gemv_chain(ctx.handle, [dict(w=ws[0], x=x, gain=g, eps=1e-6, epi=epi, res=res)])
    torch.cuda.synchronize()
    same = torch.equal(outs[0], want)
    t_l = timeit(lambda i: ops.linear(x, ws[i % ncopy], res=res, norm_w=g, eps=1e-6, epi=epi), iters)
    t_e = timeit(lambda i: ops.gemv_chain(ctx.handle, [dict(w=ws[i % ncopy], x=x, gain=g, eps=1e-6, epi=epi, res=res)], err=err), iters)
    mb = N * K * 2 / 1e6
    print(f"tp{tp} {name:20s} N{N:6d} K{K:5d} {mb:6.1f} MB | launch {t_l:6.1f} us {mb / t_l:5.2f} TB/s | engine {t_e:6.1f} us {mb / t_e:5.2f} TB/s | "
          f"{'bits equal' if same else 'BITS DIFFER max ' + str(float((outs[0].float() - want.float()).abs().max()))} | give-ups {int(err.item())}", flush=True)

# gate/up -> down
ws_g, xg, gg, _, _ = mats["gateup+norm+swiglu"]
ws_d, _, _, res_d, _ = mats["down+res"]
act = ops.linear(xg, ws_g[0], norm_w=gg, eps=1e-6, epi=ops.EPI_SWIGLU)
want = ops.linear(act, ws_d[0], res=res_d, epi=ops.EPI_RESID)
chain = lambda i, err=None: ops.gemv_chain(ctx.handle, [dict(w=ws_g[i % len(ws_g)], x=xg, gain=gg, eps=1e-6, epi=ops.EPI_SWIGLU),
                                                        dict(w=ws_d[i % len(ws_d)], x=None, epi=ops.EPI_RESID, res=res_d)], err=err)
outs, err, _ = chain(0)
torch.cuda.synchronize()
same = torch.equal(outs[1], want)


def two(i):
    a_ = ops.linear(xg, ws_g[i % len(ws_g)], norm_w=gg, eps=1e-6, epi=ops.EPI_SWIGLU)
    ops.linear(a_, ws_d[i % len(ws_d)], res=res_d, epi=ops.EPI_RESID)


t_l = timeit(two, iters)
t_e = timeit(lambda i: chain(i, err), iters)
mb = (2 * Fl * H + H * Fl) * 2 / 1e6
print(f"tp{tp} gate/up -> down       {mb:6.1f} MB | two launches {t_l:6.1f} us {mb / t_l:5.2f} TB/s | one engine launch {t_e:6.1f} us {mb / t_e:5.2f} TB/s | "
      f"{'bits equal' if same else 'BITS DIFFER max ' + str(float((outs[1].float() - want.float()).abs().max()))} | give-ups {int(err.item())}", flush=True)

# a whole layer's projections (all-reduces as plain hand-offs): o_proj -> gate/up -> down -> next qkv, one launch vs four
ws_o, xo, _, res_o, _ = mats["o+res"]
ws_q, _, gq, _, _ = mats["qkv+norm"]


def four(i):
    h = ops.linear(xo, ws_o[i % len(ws_o)], res=res_o, epi=ops.EPI_RESID)
    a_ = ops.linear(h, ws_g[i % len(ws_g)], norm_w=gg, eps=1e-6, epi=ops.EPI_SWIGLU)
    h2 = ops.linear(a_, ws_d[i % len(ws_d)], res=res_d, epi=ops.EPI_RESID)
    return ops.linear(h2, ws_q[i % len(ws_q)], norm_w=gq, eps=1e-6)


lchain = lambda i, err=None: ops.gemv_chain(ctx.handle, [dict(w=ws_o[i % len(ws_o)], x=xo, epi=ops.EPI_RESID, res=res_o),
                                                         dict(w=ws_g[i % len(ws_g)], x=None, gain=gg, eps=1e-6, epi=ops.EPI_SWIGLU),
                                                         dict(w=ws_d[i % len(ws_d)], x=None, epi=ops.EPI_RESID, res=res_d),
                                                         dict(w=ws_q[i % len(ws_q)], x=None, gain=gq, eps=1e-6, epi=ops.EPI_NONE)], err=err)
want = four(0)
outs, err, _ = lchain(0)
torch.cuda.synchronize()
same = torch.equal(outs[3], want)
t_l = timeit(four, iters)
t_e = timeit(lambda i: lchain(i, err), iters)
mb = (H * HD + 2 * Fl * H + H * Fl + 3 * HD * H) * 2 / 1e6
print(f"tp{tp} o -> gate/up -> down -> qkv {mb:6.1f} MB | four launches {t_l:6.1f} us {mb / t_l:5.2f} TB/s | one engine launch {t_e:6.1f} us {mb / t_e:5.2f} TB/s | "
      f"{'bits equal' if same else 'BITS DIFFER max ' + str(float((outs[3].float() - want.float()).abs().max()))} | give-ups {int(err.item())}", flush=True)
