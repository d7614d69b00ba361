#!/usr/bin/env python
"""Per-workgroup timeline of the GEMM kernels at the shapes the UNet / prefill launch (needs the -DEMU_TRACE twin library):

    python -m emu_amd.build --trace
    EMU_HIP_TOOLS=1 EMU_HIP_LIB=emu_amd/csrc/libemu_hip_trace.so python tools/gemm_trace.py [--shapes unet|prefill|all] [--cfgs 0,K,C,...]

For every (shape, tile configuration): the kernel's wall time (HIP events over 20 launches), then ONE traced launch: when each
workgroup entered, had its first k tile in LDS, left its main loop and had its stores out (s_memrealtime, 10 ns ticks), which XCD
it ran on, and the shader clock it saw.  Answers "where do the 22 us of a 6.7 GFLOP GEMM go": dispatch skew, first-touch
latency, main-loop rate per k tile, epilogue."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emu_amd import ops  # noqa: E402
from emu_amd._lib import lib  # noqa: E402

SHAPES = {
    "unet": [("attn-out 32^2", 2048, 1280, 1280, 1), ("qkv 32^2", 2048, 3840, 1280, 0), ("geglu 32^2", 2048, 10240, 1280, 5),
             ("ff-out 32^2", 2048, 1280, 5120, 1), ("attn-out 64^2", 8192, 640, 640, 1), ("geglu 64^2", 8192, 5120, 640, 5)],
    "prefill": [("qkv S770", 770, 19968, 6656, 0), ("o S770", 770, 6656, 6656, 1), ("gate/up S770", 770, 35840, 6656, 2),
                ("down S770", 770, 6656, 17920, 1), ("vit fc1", 1025, 15360, 1792, 4), ("vit qkv", 1025, 6144, 1792, 0)],
}


def pct(t, qs=(0.0, 0.5, 0.9, 1.0)):
    t = t.double().sort().values
    return "/".join(f"{float(t[min(len(t) - 1, int(q * (len(t) - 1) + 0.5))]):.2f}" for q in qs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="unet")
    ap.add_argument("--cfgs", default="0")
    ap.add_argument("--dump", default="")
    ap.add_argument("--only", default="", help="substring of the case name")
    ap.add_argument("--tune", type=int, default=0, help="emu_gemm_tune mask (8 = staged epilogue off)")
    ap.add_argument("--fx", default="", help="ln: LayerNorm folded into the GEMM (consumer side); lnvt: + V^T store of the last third of "
                    "the columns (the UNet's qkv projection); stats: row statistics out (producer side)")
    a = ap.parse_args()
    L = lib()
    L.emu_gemm_tune(a.tune)
    assert L.emu_gemm_trace_built() == 1, "load the -DEMU_TRACE twin: EMU_HIP_TOOLS=1 EMU_HIP_LIB=.../libemu_hip_trace.so"
    dev = torch.device("cuda", 0)
    sk = torch.zeros(512 * 288 * 256, dtype=torch.float32, device=dev)
    L.emu_set_splitk_scratch(sk.data_ptr(), sk.numel() * 4)
    buf = torch.zeros(8192 * 8, dtype=torch.int64, device=dev)
    shapes = sum((SHAPES[k] for k in (SHAPES if a.shapes == "all" else a.shapes.split(","))), [])
    g = torch.Generator(device=dev).manual_seed(0)
    for name, M, N, K, epi in shapes:
        if a.only and a.only not in name:
            continue
        x = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
        w = (torch.randn(N, K, device=dev, generator=g) * K ** -0.5).to(torch.bfloat16)
        bias = torch.randn(N, device=dev, generator=g).to(torch.bfloat16) if epi in (1, 4, 5) else None
        res = torch.randn(M, N, device=dev, generator=g).to(torch.bfloat16) if epi == 1 else None
        out = torch.empty(M, N // 2 if epi in (2, 5) else N, device=dev, dtype=torch.bfloat16)
        fn = lambda: ops.linear(x, w, bias=bias, res=res, epi=epi, out=out)
        if a.fx:
            st = torch.randn(K // 128, M, 2, device=dev, generator=g).abs() + 1.0
            cvec, dvec = torch.randn(N, device=dev, generator=g), torch.randn(N, device=dev, generator=g)
            if a.fx == "stats":
                so = torch.zeros(N // 128, M, 2, device=dev)
                fn = lambda: ops.linear_fused(x, w, bias=bias, res=res, epi=epi, out=out, stats_out=so)
            elif a.fx == "ln":
                fn = lambda: ops.linear_fused(x, w, epi=epi, out=out, ln=(cvec, dvec, st, 1e-5))
            elif a.fx == "lnvt":
                S_ = M // 2
                vt = torch.zeros(2, N // 3, S_, device=dev, dtype=torch.bfloat16)
                fn = lambda: ops.linear_fused(x, w, epi=epi, out=out, ln=(cvec, dvec, st, 1e-5), vt=(vt, 2 * N // 3, S_))
        for c in a.cfgs.split(","):
            L.emu_gemm_force_config(0 if c == "0" else ord(c))
            L.emu_gemm_trace(None)
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 20
            buf.zero_()
            L.emu_gemm_trace(buf.data_ptr())
            fn()
            torch.cuda.synchronize()
            L.emu_gemm_trace(None)
            t = buf.view(-1, 8)
            live = t[:, 0] != 0
            t = t[live].cpu()
            n = t.shape[0]
            fl = 2.0 * M * N * K
            print(f"\n== {name}  M={M} N={N} K={K} epi={epi} cfg={c}: {us:.1f} us/launch (events, incl. the launch gap) = "
                  f"{fl / us / 1e6:.0f} TFLOP/s; {n} traced workgroups (a second kernel of a split launch overwrites the first's records)")
            if n == 0:
                continue
            t0 = t[:, 0].min()
            tick = 0.01                                             # us per s_memrealtime tick
            span = float(t[:, 3].max() - t0) * tick
            xcd = (t[:, 4] & 15)
            print(f"   kernel span (first entry -> last store) {span:.2f} us;  entry skew min/p50/p90/max {pct((t[:, 0] - t0) * tick)} us")
            print(f"   first k tile landed after            {pct((t[:, 1] - t[:, 0]) * tick)} us")
            print(f"   main loop                            {pct((t[:, 2] - t[:, 1]) * tick)} us")
            print(f"   epilogue (loop end -> stores done)   {pct((t[:, 3] - t[:, 2]) * tick)} us")
            print(f"   workgroup lifetime                   {pct((t[:, 3] - t[:, 0]) * tick)} us;  exit time {pct((t[:, 3] - t0) * tick)} us")
            dt = (t[:, 2] - t[:, 0]).double() * tick
            clk = (t[:, 6] - t[:, 5]).double() / dt.clamp_min(1e-3)
            print(f"   shader clock seen by the workgroups  {pct(clk / 1e3)} GHz")
            per = []
            for xx in range(8):
                sel = xcd == xx
                if int(sel.sum()):
                    per.append(f"x{xx}:{int(sel.sum())}wg/{float((t[sel, 3].max() - t0)) * tick:.1f}us")
            print("   per XCD (workgroups / last exit): " + " ".join(per))
            cu = ((t[:, 4] >> 16) & 0xff)                          # HW_ID[15:8] = SE_ID | SH_ID | CU_ID
            both = (xcd.long() << 8) | cu.long()
            uniq, cnt = torch.unique(both, return_counts=True)
            print(f"   distinct CUs (XCD, SE/SH/CU) {len(uniq)}; workgroups per CU max {int(cnt.max())}, "
                  f"CUs with 2+ workgroups {int((cnt > 1).sum())}")
            if a.dump:
                torch.save({"name": name, "cfg": c, "t": t}, f"{a.dump}_{name.replace(' ', '_').replace('/', '_')}_{c}.pt")
    L.emu_gemm_force_config(0)


if __name__ == "__main__":
    main()
