#!/usr/bin/env python
"""Per-kernel micro-benchmarks at the bench's true shapes (HIP-event timing, random data).

    python tools/kbench.py [--filter gemv] [--iters 50]

Prints one line per case: time, and the achieved fraction of the bound that applies (HBM 8 TB/s for the weight
streaming GEMVs / decode attention, bf16 MFMA 2.5 PFLOP/s for GEMM / conv / flash attention).
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emu_amd import ops  # noqa: E402

BF16 = torch.bfloat16
HBM, MFMA = 8.0e12, 2.5e15


def timeit(fn, iters):
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


def r(*shape, scale=1.0):
    return (torch.randn(*shape, device="cuda") * scale).to(BF16)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--filter", default="")
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--tune", type=int, default=0, help="emu_gemm_tune mask for the whole run (A/B aid)")
    a = ap.parse_args()
    cases = []
    from emu_amd._lib import lib as _lib
    _lib().emu_gemm_tune(a.tune)
    _sk = torch.zeros(256 * 288 * 256, dtype=torch.float32, device="cuda")      # split-K scratch, as the engines carry
    if not os.environ.get("EMU_KBENCH_NO_SCRATCH"):
        _lib().emu_set_splitk_scratch(_sk.data_ptr(), _sk.numel() * 4)

    def gemv(name, M, N, K, epi=0, norm=False):
        # rotate over several weight copies so the 256 MB infinity cache cannot serve the "stream"
        n_copy = max(1, int(600e6 // (N * K * 2)) + 1)
        ws = [r(N, K, scale=0.02) for _ in range(n_copy)]
        x = r(M, K)
        res = r(M, N) if epi == 1 else None
        g = r(K) if norm else None
        st = {"i": 0}

        def fn():
            st["i"] = (st["i"] + 1) % n_copy
            ops.linear(x, ws[st["i"]], res=res, norm_w=g, eps=1e-6, epi=epi)
        cases.append((f"gemv {name} M{M} N{N} K{K}", fn, 2.0 * N * K, "hbm"))

    def gemv8(name, M, N, K, epi=0, norm=False):
        n_copy = max(1, int(600e6 // (N * K)) + 1)
        ws = [ops.quantize_fp8_rows(r(N, K, scale=0.02)) for _ in range(n_copy)]
        x = r(M, K)
        res = r(M, N) if epi == 1 else None
        g = r(K) if norm else None
        st = {"i": 0}

        def fn():
            st["i"] = (st["i"] + 1) % n_copy
            q, sc = ws[st["i"]]
            ops.linear_fp8w(x, q, sc, res=res, norm_w=g, eps=1e-6, epi=epi)
        cases.append((f"gemv-fp8 {name} M{M} N{N} K{K}", fn, 1.0 * N * K, "hbm"))

    def gemm(name, M, N, K, epi=0):
        x, w = r(M, K), r(N, K, scale=0.02)
        res = r(M, N) if epi == 1 else None
        cases.append((f"gemm {name} M{M} N{N} K{K}", lambda: ops.linear(x, w, res=res, epi=epi), 2.0 * M * N * K, "mfma"))

    def conv(name, B, H, Cin, Cout, mode=1):
        x, w = r(B, H, H, Cin), r(Cout, 3, 3, Cin, scale=0.02)
        Ho = H if mode == 1 else (H // 2 if mode == 2 else 2 * H)
        cases.append((f"conv3x3 {name} {H}x{H} {Cin}->{Cout} mode{mode}", lambda: ops.conv3x3_nhwc(x, w, mode=mode),
                      2.0 * B * Ho * Ho * Cout * 9 * Cin, "mfma"))

    def flash(name, B, S, H, D, causal, Sk=None):
        Sk = Sk or S
        q, k, v = r(B, S, H, D), r(B, Sk, H, D), r(B, Sk, H, D)
        fl = 4.0 * B * H * S * Sk * D * (0.5 if causal else 1.0)
        cases.append((f"flash {name} B{B} S{S}x{Sk} H{H} D{D}", lambda: ops.flash_attn(q, k, v, causal, D ** -0.5), fl, "mfma"))

    # ---- decode (LLaMA-33B, TP=1 and TP=8 shard shapes)
    gemv("qkv+norm", 1, 19968, 6656, 0, True)
    gemv("o+res", 1, 6656, 6656, 1)
    gemv("gateup+norm+swiglu", 1, 35840, 6656, 2, True)
    gemv("down+res", 1, 6656, 17920, 1)
    gemv("lm_head+norm", 1, 32274, 6656, 0, True)
    gemv8("qkv+norm", 1, 19968, 6656, 0, True)
    gemv8("o+res", 1, 6656, 6656, 1)
    gemv8("gateup+norm+swiglu", 1, 35840, 6656, 2, True)
    gemv8("down+res", 1, 6656, 17920, 1)
    gemv8("lm_head+norm", 1, 32274, 6656, 0, True)
    gemv("qkv rows2", 2, 19968, 6656, 0, False)
    gemv("qkv rows3", 3, 19968, 6656, 0, False)
    gemv("qkv rows8", 8, 19968, 6656, 0, False)
    gemv("qkv beams5", 5, 19968, 6656, 0, False)
    gemv("gateup beams5", 5, 35840, 6656, 2, False)
    gemv("down beams5", 5, 6656, 17920, 1, False)
    gemv("o beams5", 5, 6656, 6656, 1, False)
    gemv("tp8 qkv", 1, 2688, 6656, 0, True)
    gemv("tp8 o", 1, 6656, 896, 1)
    gemv("tp8 gateup", 1, 4480, 6656, 2, True)
    gemv("tp8 down", 1, 6656, 2240, 1)
    # ---- prefill S=770 / ViT N=1025 / UNet
    gemm("short-prompt qkv", 40, 19968, 6656)
    gemm("short-prompt gateup", 40, 35840, 6656, 2)
    gemm("short-prompt down", 40, 6656, 17920, 1)
    gemm("prefill qkv", 770, 19968, 6656)
    gemm("prefill o", 770, 6656, 6656, 1)
    gemm("prefill gateup", 770, 35840, 6656, 2)
    gemm("prefill down", 770, 6656, 17920, 1)
    gemm("prefill1544 gateup", 1544, 35840, 6656, 2)
    gemm("vit qkv", 1025, 6144, 1792)
    gemm("vit fc1", 1025, 15360, 1792, 4)
    gemm("vit fc2", 1025, 1792, 15360)
    gemm("unet 32^2 attn-out", 2048, 1280, 1280, 1)
    gemm("unet 32^2 qkv", 2048, 3840, 1280)
    gemm("unet 32^2 geglu", 2048, 10240, 1280, 5)
    gemm("unet 32^2 ff-out", 2048, 1280, 5120, 1)
    gemm("unet 64^2 qkv", 8192, 1920, 640)
    gemm("unet 64^2 geglu", 8192, 5120, 640, 5)
    conv("unet lvl2", 2, 32, 1280, 1280)
    conv("unet lvl2 cat", 2, 32, 2560, 1280)
    conv("unet lvl1", 2, 64, 640, 640)
    conv("unet lvl0", 2, 128, 320, 320)
    conv("unet lvl0 cat", 2, 128, 960, 320)
    conv("unet up", 2, 32, 1280, 1280, 3)
    flash("llama prefill", 1, 770, 52, 128, True)
    flash("vit", 1, 1025, 16, 128, False)
    flash("unet 64^2 self", 2, 4096, 10, 64, False)
    flash("unet 32^2 self", 2, 1024, 20, 64, False)
    flash("unet 32^2 cross", 2, 1024, 20, 64, False, Sk=64)

    # ---- decode attention (fused rope + append + attention), ctx 800
    def dec(name, H):
        from emu_amd._lib import lib, check
        from emu_amd.llama import rope_tables
        D, S_max, ctx = 128, 2048, 800
        qkv = r(1, 3 * H * D)
        cos, sin = rope_tables(D, S_max, 1e4, "cuda")
        kc, vc = r(1, H, S_max, D), r(1, H, S_max, D)
        pos = torch.tensor([ctx - 1], dtype=torch.int32, device="cuda")
        import ctypes as C
        L = lib()
        cfg_bytes = 2 * ctx * H * D * 2
        # reach the kernel through the LLaMA engine entry used in production: build a 1-layer engine
        from emu_amd import synth
        from emu_amd.conf.emu_conf import LlamaCfg
        from emu_amd.llama import EmuHipContext, LlamaEngine
        l = LlamaCfg(hidden_size=H * D, intermediate_size=64, num_attention_heads=H, num_hidden_layers=1)
        eng = LlamaEngine(l, 64, EmuHipContext(torch.device("cuda", 0)))
        eng.load_weights(synth.iter_synth(synth.llama_param_shapes(l, 64), device="cuda", dtype=BF16))
        eng.alloc_kv(1, S_max)
        x = r(1, H * D)
        slot = torch.tensor([ctx - 1], dtype=torch.int32, device="cuda")
        cases.append((f"decode layer(attn-dominated) {name} H{H} ctx{ctx}",
                      lambda: eng.forward(x.clone(), 1, 1, pos, slot, None, ctx=S_max), cfg_bytes, "hbm"))
    dec("tp1", 52)

    print(f"{'case':58s} {'time':>10s} {'achieved':>14s} {'frac':>7s}")
    for name, fn, work, bound in cases:
        if a.filter and a.filter not in name:
            continue
        t = timeit(fn, a.iters)
        if bound == "hbm":
            print(f"{name:58s} {t * 1e6:8.1f}us {work / t / 1e12:9.2f} TB/s {work / t / HBM:7.3f}")
        else:
            print(f"{name:58s} {t * 1e6:8.1f}us {work / t / 1e12:9.1f} TF/s {work / t / MFMA:7.3f}")


if __name__ == "__main__":
    main()
