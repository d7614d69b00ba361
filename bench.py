#!/usr/bin/env python
"""Headline benchmark: Emu2-Chat 37B greedy text decode (BASELINE.json configs[1]) on MI355X.

    python bench.py --gpus 1 --steps 64 --warmup 8
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload: synthetic Emu2-Chat weights (EVA-CLIP 4B ViT + LLaMA-33B, seeded
N(0, 0.02^2), generated on the GPU), one 448x448 image (256 visual tokens) spliced into a 512-token prompt
(S = 770), greedy decode.  A "step" is one decoded token (batch 1) through all 60 decoder layers + lm_head.
The untimed setup runs the ViT encode and the S=770 prefill (reported separately); the timed region is
exactly K decode steps between barrier + synchronize pairs; rank 0 prints ONE JSON line.

N > 1: tensor parallel over the N GPUs (heads/ffn sharded, RCCL all-reduce over xGMI): total work is fixed, so
"scaling" is "strong".  `roofline` is the weight-streaming GEMV (dominant kernel, HBM-bound); `cpu_baseline`
is the CPU oracle (oracle/emu2_ref.py) timed on this box's host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12      # B/s, MI355X spec (MI355X_MICROARCH.md)
MFMA_BF16_PEAK = 2.5e15
MFMA_FP8_PEAK = 5.0e15    # dense fp8 (MI355X_MICROARCH.md); priced against for the W8A8 legs only


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=64)
    p.add_argument("--warmup", type=int, default=8)
    p.add_argument("--layers", type=int, default=60, help="debug only: fewer decoder layers => result INVALID")
    p.add_argument("--vit-layers", type=int, default=64, help="debug only")
    p.add_argument("--prompt-tokens", type=int, default=512)
    p.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-denoise", action="store_true", help="skip the UNet denoise leg")
    p.add_argument("--denoise-steps", type=int, default=50)
    p.add_argument("--only-denoise", action="store_true", help="profiling aid: run just the UNet leg (prints its object)")
    p.add_argument("--unet-fusion", type=int, default=-1, help="A/B aid: emu_unet_set_fusion mask for the denoise leg (default: all)")
    p.add_argument("--gemm-tune", type=int, default=0, help="A/B aid: emu_gemm_tune mask for the whole run (0 = the shipped dispatch; "
                   "anything else marks the line invalid)")
    p.add_argument("--no-beam", action="store_true", help="skip the extra 5-beam leg (the reference's default decoding mode)")
    p.add_argument("--no-fp8", action="store_true", help="skip the extra fp8-weight decode leg (never the headline value)")
    p.add_argument("--cpu-seconds", type=float, default=12.0)
    p.add_argument("--tp-prefill-leg", type=int, default=0, metavar="S", help="N > 1 only, opt-in: after everything else, time an S-row prefill "
                   "(e.g. 1544 = BASELINE configs[2]) under the serial all-reduce schedule and as two concurrent lanes (emu_llama_set_tp_overlap); "
                   "reported in config.tp.prefill_schedules.  Off by default: the two-lane schedule has never run on more than one GPU")
    p.add_argument("--no-legs", action="store_true", help="skip the extra BASELINE-config legs (S=1544 prefill, generate_image, "
                                                          "VAE decode, any-to-image)")
    p.add_argument("--pmc-prefill", type=int, default=0, metavar="N", help="profiling aid for rocprofv3 --pmc passes: load the decoder only, "
                   "run exactly N prefills of the S=770 prompt shape and print the algorithmic bytes of their GEMMs")
    p.add_argument("--pmc-mode", type=int, default=0, metavar="N", help="profiling aid for rocprofv3 --pmc passes: after the "
                   "prefill run exactly N eager decode steps and print the algorithmic bytes of every GEMV launch of the process")
    return p.parse_args()


def gemm_source_hash():
    """sha256 over the sources of the MFMA GEMM kernels: the prefill leg's PMC traffic ratio is only quoted for exactly this code."""
    import hashlib
    h = hashlib.sha256()
    for f in ("gemm.hip", "gemm256.hip", "gemm_w4.hip", "gemm_tile.h", "common.h"):
        h.update(open(os.path.join(ROOT, "emu_amd", "csrc", f), "rb").read())
    return h.hexdigest()


def prefill_gemm_bytes(lcfg, S):
    """Algorithmic bytes of the GEMMs of one S-token prefill: every weight once, every GEMM's activation operands once
    (input rows, output rows, the residual of o_proj / down_proj)."""
    H, F, L = lcfg.hidden_size, lcfg.intermediate_size, lcfg.num_hidden_layers
    w = 4 * H * H + 3 * H * F
    act = S * ((H + 3 * H) + (H + 2 * H) + (H + F) + (F + 2 * H))       # qkv | o (+res) | gate/up (SwiGLU output F) | down (+res)
    return 2 * L * (w + act)


def gemv_source_hash():
    """sha256 over the sources of the kernel the roofline object describes: a PMC traffic ratio is only quoted when it was
    measured on exactly this code."""
    import hashlib
    h = hashlib.sha256()
    for f in ("gemv.hip", "common.h"):
        h.update(open(os.path.join(ROOT, "emu_amd", "csrc", f), "rb").read())
    return h.hexdigest()


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def cpu_baseline(seconds: float, ctx_len: int, vocab: int):
    """CPU oracle (port of the reference algorithm, oracle/emu2_ref.py) on the host cores: ONE true-shape
    LLaMA-33B decoder layer, cached decode step at ctx_len; tokens/s is extrapolated as
    1 / (60 * t_layer + t_lm_head).  The (dtype, thread-count) pair is the fastest of a short calibration
    (torch CPU bf16 mat-vec can be far slower than fp32, and 256 threads slower than 32); `cores` reports the
    threads actually used.  Bounded to roughly `seconds` of measurement after calibration."""
    from emu_amd import synth
    from emu_amd.conf.emu_conf import LlamaCfg
    from oracle import emu2_ref as R
    ncpu = os.cpu_count() or 1
    l = LlamaCfg(num_hidden_layers=1)
    shapes = {k: s for k, s in synth.llama_param_shapes(l, vocab).items() if "embed_tokens" not in k}
    W32 = {k: synth.synth_tensor(k, s, seed=0) for k, s in shapes.items()}
    cfg = R.LlamaCfg(layers=1, vocab=vocab)
    g = torch.Generator().manual_seed(0)
    k0 = torch.randn(1, cfg.heads, ctx_len, cfg.head_dim, generator=g)
    v0 = torch.randn(1, cfg.heads, ctx_len, cfg.head_dim, generator=g)
    x0 = torch.randn(1, 1, cfg.hidden, generator=g)
    pos = torch.tensor([[ctx_len]])
    mask = torch.ones(1, ctx_len + 1, dtype=torch.long)

    def make(dtype):
        W = W32 if dtype == torch.float32 else {k: t.to(dtype) for k, t in W32.items()}
        kk, vv, x = k0.to(dtype), v0.to(dtype), x0.to(dtype)

        def one_layer():
            c = R.KVCache(1)
            c.k[0], c.v[0] = kk, vv
            return R.llama_model(x, mask, W, cfg, position_ids=pos, cache=c, final_norm=False)

        def head():
            h = R.rms_norm(x, W["decoder.lm.model.norm.weight"], cfg.rms_eps)
            return torch.nn.functional.linear(h, W["decoder.lm.lm_head.weight"])
        return one_layer, head

    cands = []
    for dtype in (torch.float32, torch.bfloat16):
        for th in sorted({ncpu, min(ncpu, 64), min(ncpu, 32), min(ncpu, 16)}, reverse=True):
            cands.append((dtype, th))
    best = None
    t_cal = time.perf_counter()
    with torch.no_grad():
        for dtype, th in cands:
            if time.perf_counter() - t_cal > seconds:            # calibration budget
                break
            torch.set_num_threads(th)
            one_layer, head = make(dtype)
            one_layer()
            t0 = time.perf_counter(); one_layer(); dt = time.perf_counter() - t0
            if best is None or dt < best[0]:
                best = (dt, dtype, th)
        _, dtype, th = best
        torch.set_num_threads(th)
        W = W32
        one_layer, head = make(dtype)
        one_layer(); head()
        t0 = time.perf_counter(); n = 0
        while time.perf_counter() - t0 < seconds * 0.8 or n < 2:
            one_layer(); n += 1
        t_layer = (time.perf_counter() - t0) / n
        t0 = time.perf_counter(); m = 0
        while time.perf_counter() - t0 < seconds * 0.2 or m < 2:
            head(); m += 1
        t_head = (time.perf_counter() - t0) / m
    tok_s = 1.0 / (60 * t_layer + t_head)
    dn = "fp32" if dtype == torch.float32 else "bf16"
    port = {"value": tok_s, "unit": "tokens/s", "cores": th, "kind": "port", "layer_ms": t_layer * 1e3,
            "sample": f"1 true-shape LLaMA-33B decoder layer x{n} + lm_head x{m} ({dn}, ctx {ctx_len}, batch 1) on {th} of "
                      f"{ncpu} host threads (fastest of a dtype/thread calibration); tokens/s = 1/(60*{t_layer * 1e3:.1f} ms "
                      f"+ {t_head * 1e3:.1f} ms)"}
    # the layer the REFERENCE executes: transformers' own LlamaDecoderLayer (Emu2/emu/lm.py builds LlamaForCausalLM; the
    # arithmetic is the library's), same shapes / dtype / threads / cached context, timed beside the port
    try:
        t_ref, n_ref = _reference_layer_time(W, dtype, cfg, ctx_len, k0, v0, x0, seconds * 0.5)
        ref_tok_s = 1.0 / (60 * t_ref + t_head)
        return {"value": ref_tok_s, "unit": "tokens/s", "cores": th, "kind": "reference", "layer_ms": t_ref * 1e3,
                "sample": f"transformers {__import__('transformers').__version__} LlamaDecoderLayer (eager attention, DynamicCache with {ctx_len} "
                          f"cached positions) at the LLaMA-33B shape x{n_ref} + lm_head x{m} ({dn}, batch 1) on {th} of {ncpu} host threads; "
                          f"tokens/s = 1/(60*{t_ref * 1e3:.1f} ms + {t_head * 1e3:.1f} ms)",
                "port": port}
    except Exception as e:                                  # library API drift: keep the port
        port["reference_note"] = f"transformers layer not timed: {type(e).__name__}: {e}"
        return port


def _reference_layer_time(W32, dtype, cfg, ctx_len, k0, v0, x0, seconds):
    """Seconds per cached decode step of transformers' LlamaDecoderLayer holding the port's weights."""
    import inspect
    from transformers import LlamaConfig
    from transformers.cache_utils import DynamicCache
    from transformers.models.llama import modeling_llama as ML
    lc = LlamaConfig(hidden_size=cfg.hidden, intermediate_size=cfg.ffn, num_attention_heads=cfg.heads, num_key_value_heads=cfg.heads,
                     num_hidden_layers=1, vocab_size=cfg.vocab, max_position_embeddings=cfg.max_pos, rms_norm_eps=cfg.rms_eps,
                     rope_theta=cfg.rope_theta)
    lc._attn_implementation = "eager"
    layer = ML.LlamaDecoderLayer(lc, 0).eval()
    pre = "decoder.lm.model.layers.0."
    sd = {k[len(pre):]: v for k, v in W32.items() if k.startswith(pre)}
    layer.load_state_dict(sd, strict=True)
    layer = layer.to(dtype)
    rot = ML.LlamaRotaryEmbedding(lc)
    x = x0.to(dtype)
    pos = torch.tensor([[ctx_len]])
    cos, sin = rot(x, pos)
    kk, vv = k0.to(dtype), v0.to(dtype)

    def fresh():
        c = DynamicCache(config=lc) if "config" in inspect.signature(DynamicCache.__init__).parameters else DynamicCache()
        c.update(kk, vv, 0)
        return c

    def step():
        c = fresh()                                         # the layer appends one position per call
        t0 = time.perf_counter()
        layer(x, attention_mask=None, position_ids=pos, past_key_values=c, use_cache=True, position_embeddings=(cos, sin))
        return time.perf_counter() - t0
    with torch.no_grad():
        step()
        tot, n = 0.0, 0
        t_end = time.perf_counter() + seconds
        while time.perf_counter() < t_end or n < 2:
            tot += step(); n += 1
    return tot / n, n


def _filled(shapes, seed=0):
    """Weights for a CPU TIMING run: every tensor its own memory, filled from one small random pool by block copies (drawing
    2.5 B normals on one host thread would take longer than the measurement); norm gains 1, values N(0, 0.02^2)."""
    g = torch.Generator().manual_seed(seed)
    pool = torch.randn(1 << 22, generator=g) * 0.02
    out = {}
    for k, sh in shapes.items():
        n = 1
        for d in sh:
            n *= d
        if k.endswith(("norm.weight", "norm1.weight", "norm2.weight", "norm3.weight", "layernorm.weight", "norm_out.weight")) and len(sh) == 1:
            out[k] = torch.ones(sh)
        else:
            t = torch.empty(n)
            for o in range(0, n, pool.numel()):
                m = min(pool.numel(), n - o)
                t[o:o + m] = pool[:m]
            out[k] = t.view(sh)
    return out


def cpu_baseline_legs(seconds: float, threads: int, S: int):
    """The reference's CPU path for the non-decode legs, on a bounded sample each: one UNet forward of the
    CFG pair (restated diffusers-0.24 arithmetic, oracle/unet_ref.py: diffusers is not installable here) -> denoise steps/s; one
    true-shape EVA-CLIP block -> images/s = 1 / (64 blocks); one LLaMA-33B decoder layer over the S prompt rows -> prefill ms =
    60 layers.  fp32 on `threads` host threads (the decode baseline's calibrated thread count)."""
    from emu_amd import synth
    from emu_amd.conf.emu_conf import CLIPVisionCfg, LlamaCfg
    from oracle import emu2_ref as R
    from oracle import unet_ref as U
    ncpu = os.cpu_count() or 1
    torch.set_num_threads(threads)
    out = {}
    with torch.no_grad():
        # ---- ViT block
        v = CLIPVisionCfg()
        vc = R.VitCfg()                                                # the EVA-CLIP defaults (448 / 14 / 1792 / 64 blocks / MLP 15360)
        shapes = {k: sh for k, sh in synth.vit_param_shapes(v).items() if k.startswith("visual.blocks.0.")}
        W = _filled(shapes)
        x = torch.randn(1, vc.tokens, vc.width) * 0.5
        R.vit_block(x, W, 0, vc)
        t0 = time.perf_counter(); n = 0
        while time.perf_counter() - t0 < seconds * 0.15 or n < 2:
            R.vit_block(x, W, 0, vc); n += 1
        tb = (time.perf_counter() - t0) / n
        out["vit"] = {"value": 1.0 / (vc.layers * tb), "unit": "images/s", "cores": threads, "kind": "port",
                      "sample": f"1 true-shape EVA-CLIP block (1025 x 1792, 16 heads, MLP 15360; oracle/emu2_ref.vit_block) x{n}, fp32 on "
                                f"{threads} of {ncpu} host threads; images/s = 1 / ({vc.layers} x {tb * 1e3:.1f} ms)"}
        del W
        # ---- prefill layer
        l = LlamaCfg(num_hidden_layers=1)
        shapes = {k: sh for k, sh in synth.llama_param_shapes(l, 8).items() if ".layers.0." in k}
        W = _filled(shapes)
        cfg = R.LlamaCfg(layers=1, vocab=8)
        x = torch.randn(1, S, cfg.hidden) * 0.1
        mask = torch.ones(1, S, dtype=torch.long)
        R.llama_model(x, mask, W, cfg, final_norm=False)
        t0 = time.perf_counter(); n = 0
        while time.perf_counter() - t0 < seconds * 0.25 or n < 1:
            R.llama_model(x, mask, W, cfg, final_norm=False); n += 1
        tl = (time.perf_counter() - t0) / n
        out["prefill"] = {"value": 60 * tl * 1e3, "unit": "ms", "cores": threads, "kind": "port", "higher_is_better": False,
                          "sample": f"1 true-shape LLaMA-33B decoder layer over the S = {S} prompt rows (oracle/emu2_ref.llama_layer, "
                                    f"eager fp32 attention) x{n}, fp32 on {threads} of {ncpu} host threads; prefill = 60 x {tl * 1e3:.0f} ms"}
        del W
        # ---- UNet forward of the CFG pair
        ucfg = U.UNetCfg()
        W = _filled(U.unet_param_shapes(ucfg))
        g = torch.Generator().manual_seed(3)
        prompt = torch.randn(2, 64, 1792, generator=g)
        lat = torch.randn(2, 4, 128, 128, generator=g)
        tid = torch.tensor([1024, 1024, 0, 0, 1024, 1024] * 2)
        t0 = time.perf_counter()
        U.unet_forward(lat[:1], torch.tensor(981.0), prompt[:1], prompt[:1].mean(1), tid[:6], W, ucfg)
        t1 = time.perf_counter() - t0                                  # the cond half alone (also the warm-up)
        if 2.2 * t1 < seconds:
            t0 = time.perf_counter()
            U.unet_forward(lat, torch.tensor(981.0), prompt, prompt.mean(1), tid, W, ucfg)
            t2 = time.perf_counter() - t0
            how = f"one forward of the CFG pair (batch 2) = {t2:.1f} s"
        else:
            t2 = 2 * t1
            how = f"one forward of the cond half (batch 1) = {t1:.1f} s, doubled for the CFG pair"
        del W
        # ---- VAE decode (restated AutoencoderKL decoder, oracle/vae_ref.py), the whole 1024 x 1024 image once
        try:
            from oracle import vae_ref as V
            vc_ = V.VaeCfg()
            Wv = _filled(V.vae_decoder_param_shapes(vc_))
            z = torch.randn(1, 4, 128, 128, generator=g)
            t0 = time.perf_counter()
            V.decode_latents(z, Wv, vc_)
            tv = time.perf_counter() - t0
            out["vae"] = {"value": tv * 1e3, "unit": "ms", "cores": threads, "kind": "port", "higher_is_better": False,
                          "sample": f"restated AutoencoderKL.decode (oracle/vae_ref.decode_latents: diffusers is not installable here), latents "
                                    f"[1, 4, 128, 128] -> 1024 x 1024, ONE call (cold), fp32 on {threads} of {ncpu} host threads"}
            del Wv
        except Exception as e:
            out["vae"] = {"value": None, "unit": "ms", "cores": threads, "kind": "port", "sample": f"failed: {type(e).__name__}: {e}"}
        out["denoise"] = {"value": 1.0 / t2, "unit": "steps/s", "cores": threads, "kind": "port",
                          "sample": f"restated SDXL-style UNet (2.53 B parameters, oracle/unet_ref.unet_forward: diffusers 0.24 is not installable "
                                    f"here, so the port is the reference arithmetic), 128 x 128 latents, 64 context tokens, fp32 on {threads} of "
                                    f"{ncpu} host threads: {how}; guidance + Euler update are negligible beside it"}
    return out


def vae_decode_flops(h: int = 128, chans=(128, 256, 512, 512), layers: int = 2, latent: int = 4) -> float:
    """Algorithmic FLOPs of AutoencoderKL.decode for an h x h latent (vae/config.json): 3x3 convs 2 * pixels * Cout * 9 * Cin, the mid
    block's one-head attention over h^2 tokens (4 projections + QK^T + PV), 1x1 shortcuts; norms / activations excluded.
    10.47 TFLOP at h = 128 (a 1024 x 1024 image)."""
    conv = lambda px, ci, co, k=9: 2.0 * px * co * k * ci
    top = chans[-1]
    px = h * h
    f = conv(px, latent, top) + 2 * 2 * conv(px, top, top) + 4 * 2.0 * px * top * top + 4.0 * px * px * top
    rev, cur = list(reversed(chans)), chans[-1]
    for i, c in enumerate(rev):
        for j in range(layers + 1):
            cin = cur if j == 0 else c
            f += conv(px, cin, c) + conv(px, c, c) + (conv(px, cin, c, 1) if cin != c else 0.0)
        cur = c
        if i < len(rev) - 1:
            px *= 4
            f += conv(px, c, c)
    return f + conv(px, chans[0], 3)


VIT_FLOPS_PER_IMAGE = 9.39e12       # BASELINE.md section 2 / SURVEY 8d: EVA-CLIP 64 blocks x 1025 tokens x 1792
UNET_FLOPS_PER_STEP = 13.48e12      # one denoise step (CFG batch 2) at 128x128 latents, 64 ctx tokens
UNET_WEIGHT_BYTES = 2.0 * 2.53e9      # 2.53 B parameters, bf16


def denoise_leg(ctx, dev, steps, world, dist, fusion=-1):
    """prompt_embeds randn(2,64,1792) seed 3, latents randn(1,4,128,128) seed 4, 50 Euler steps,
    CFG 3.0, 1024x1024; synthetic UNet weights (2.53 B params) generated on the GPU.  The timed region is exactly the
    `steps`-iteration loop (scale_model_input -> UNet -> CFG -> Euler step per iteration), hipGraph replayed.
    Across GPUs this path does not shard: every rank runs an independent replica; aggregate = sum."""
    from emu_amd import synth
    from emu_amd.unet import UNetCfg, UNetEngine, unet_param_shapes
    cfg = UNetCfg()
    eng = UNetEngine(cfg, ctx)
    eng.load_state_dict(synth.iter_synth(unet_param_shapes(cfg), seed=0, device=dev, dtype=torch.bfloat16))
    fusion = eng.set_fusion(7 if fusion < 0 else fusion)
    g = torch.Generator().manual_seed(3)
    prompt = torch.randn(2, 64, 1792, generator=g).to(torch.bfloat16).to(dev)
    sch = eng.set_timesteps(steps)
    eng.set_context(prompt, 1024, 1024)
    g = torch.Generator().manual_seed(4)
    lat0 = (torch.randn(1, 4, 128, 128, generator=g) * sch.init_noise_sigma).to(torch.bfloat16).to(dev)

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    lat = lat0.clone()
    with torch.no_grad():
        eng.denoise(lat, 3.0, use_graph=True, steps=3)                  # warm-up: eager step + graph capture + replay
        eng.set_timesteps(steps)
        lat.copy_(lat0)
        sync(); t = time.perf_counter()
        eng.denoise(lat, 3.0, use_graph=True, steps=steps)
        sync(); dt = time.perf_counter() - t
    if dist is not None:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    per_gpu = steps / dt
    finite = bool(torch.isfinite(lat.float()).all())
    denoise_leg.engine = eng
    # per-kernel table of the same engine in the same process: two eagerly launched steps with HIP events around every GEMM /
    # conv / attention launch (emu_profile_launches), aggregated by shape
    table = None
    try:
        import ctypes as C
        from emu_amd._lib import ProfRowC, lib
        L = lib()
        with torch.no_grad():
            eng.set_timesteps(steps)
            lat.copy_(lat0)
            eng.denoise(lat, 3.0, use_graph=False, steps=1)
            L.emu_profile_launches(1)
            eng.denoise(lat, 3.0, use_graph=False, steps=2)
            torch.cuda.synchronize()
            rows = (ProfRowC * 256)()
            n = L.emu_profile_launches_read(rows, 256)
            L.emu_profile_launches(0)
        rs = sorted((rows[i] for i in range(max(0, n))), key=lambda r: -r.ms)
        tot = sum(r.ms for r in rs) / 2
        epi_names = {0: "", 1: "+res", 2: "swiglu", 3: "silu", 4: "gelu", 5: "geglu"}
        table = {"source": "this run: 2 eager steps, HIP events around each launch (emu_profile_launches); frac = FLOP / time / 2.5 PFLOP/s",
                 "mfma_launch_ms_per_step": tot, "share_of_step": tot / (dt / steps * 1e3),
                 "top": [{"kernel": (f"{r.klass.decode()} {r.M}x{r.N}x{r.K} {epi_names.get(r.tag & 255, '')}"
                                     f"{' fx' + str(r.tag >> 8) if (r.tag >> 8) and r.klass != b'attn' else ''}"
                                     f"{' heads*batch ' + str(r.tag) if r.klass == b'attn' else ''}").strip(),
                          "calls_per_step": r.launches / 2, "us": r.ms * 1e3 / r.launches, "ms_per_step": r.ms / 2,
                          "gflop": r.flops / r.launches / 1e9, "frac": r.flops / (r.ms * 1e-3) / MFMA_BF16_PEAK} for r in rs[:10]]}
    except Exception as e:                                  # the table is a convenience, never a reason to lose the leg
        table = {"note": f"kernel table failed: {e}"}
    # ---- extra (never the value of this leg): the 70 transformer blocks W8A8 (BASELINE.json configs[4] names an fp8 MFMA path;
    # the reference itself is bf16): same latents, same schedule, final latents against the bf16 run's
    fp8 = None
    if not getattr(denoise_leg, "no_fp8", False):
        try:
            with torch.no_grad():
                eng.set_timesteps(steps); lat.copy_(lat0)
                eng.denoise(lat, 3.0, use_graph=True, steps=steps)
                lat_bf16 = lat.clone()
                eng.use_fp8(True)
                eng.set_timesteps(steps); lat.copy_(lat0)
                eng.denoise(lat, 3.0, use_graph=True, steps=3)
                eng.set_timesteps(steps); lat.copy_(lat0)
                sync(); t = time.perf_counter()
                eng.denoise(lat, 3.0, use_graph=True, steps=steps)
                sync(); dt8 = time.perf_counter() - t
            err = float((lat.float() - lat_bf16.float()).norm() / lat_bf16.float().norm())
            fp8 = {"ms_per_step": dt8 / steps * 1e3, "steps_per_s_per_gpu": steps / dt8,
                   "rel_l2_of_final_latents_vs_bf16_run": err, "finite_output": bool(torch.isfinite(lat.float()).all()),
                   "what": "the six GEMMs of each of the 70 transformer blocks on fp8 operands (v_mfma_scale_f32_32x32x64_f8f6f4, per-row "
                           "e4m3 scales on weights and activation rows; the blocks' LayerNorms emit the fp8 rows, attention outputs and "
                           "the GEGLU product are quantised by a launch of their own); convs, GroupNorm, attention, proj_in / proj_out bf16",
                   "note": "extra leg on random-init weights: the distance is what per-row e4m3 costs on unstructured matrices over "
                           f"{steps} steps, not a quality claim; never this leg's value",
                   "roofline": {"bound": "mfma", "achieved": UNET_FLOPS_PER_STEP * steps / dt8 / 1e12, "peak": MFMA_FP8_PEAK / 1e12,
                                "unit": "TFLOP/s", "frac": UNET_FLOPS_PER_STEP * steps / dt8 / MFMA_FP8_PEAK, "traffic": None,
                                "note": "whole step against the dense fp8 peak although only the transformer blocks' GEMMs (7.9 of the "
                                        "13.48 TFLOP) run fp8: the convs and attention are bf16, so this understates the fp8 GEMMs"}}
        except Exception as e:
            fp8 = {"ms_per_step": None, "note": f"fp8 transformer-block leg failed: {e}"}
        finally:
            try:
                eng.use_fp8(False)
            except Exception:
                pass
    return {"metric": "diffusion denoise steps/sec (UNet fwd CFG batch 2 + guidance + Euler step, 1024x1024, 64 ctx tokens)",
            "value": per_gpu * world, "unit": "steps/s", "per_gpu": per_gpu, "steps": steps, "ms_per_step": dt / steps * 1e3,
            "scaling": "replicas only (independent images per GPU)", "launch": "hipGraph replay", "finite_output": finite,
            "kernels": table, "fp8_transformer_blocks": fp8,
            "fusion": {"mask": fusion, "layernorm_folded_into_gemm": bool(fusion & 1), "v_transpose_in_qkv_epilogue": bool(fusion & 2),
                       "cross_attention_in_to_q_epilogue": bool(fusion & 4)},
            "roofline": {"bound": "mfma", "achieved": UNET_FLOPS_PER_STEP * per_gpu / 1e12, "peak": MFMA_BF16_PEAK / 1e12,
                         "unit": "TFLOP/s", "frac": UNET_FLOPS_PER_STEP * per_gpu / MFMA_BF16_PEAK,
                         "flops_per_step": UNET_FLOPS_PER_STEP,
                         # BASELINE.json's north_star words the UNet target as an HBM fraction; the step is MFMA-bound (13.48 TFLOP
                         # over 5.05 GB of weights = 2670 FLOP per weight byte, the chip's balance point is ~310), so the HBM view of
                         # the same measurement is reported beside it and is small by construction
                         "hbm_view": {"algorithmic_weight_bytes_per_step": UNET_WEIGHT_BYTES, "frac_of_hbm_peak":
                                      UNET_WEIGHT_BYTES * per_gpu / HBM_PEAK,
                                      "note": "every UNet weight read once per step (2.53 B parameters, bf16); activations stay in L2 / MALL"}}}


def config_legs(m, lm, ctx, dev, vcfg, lcfg, img, unet_eng):
    """Timed legs for the other BASELINE.json configs (reported next to the headline, never as `value`):
    configs[2] few-shot prefill (4 images + 512 tokens, S = 1544), configs[3]/[4] the visual-embedding regression
    `generate_image` (prompt + [IMG] prefill, then n_query - 1 = 63 KV-cached steps through project_down / project_up:
    every step streams all decoder weights), the VAE decode, and the any-to-image chain end to end
    (generate_image for the prompt and for the negative prompt -> 50 denoise steps -> VAE decode)."""
    from emu_amd import synth
    from emu_amd.constants import IMAGE_TOKEN_ID, IMG_END_TOKEN_ID, IMG_TOKEN_ID
    from emu_amd.vae import VaeCfg, VaeDecoder, vae_decoder_param_shapes
    out = {}

    def timed(fn, reps=2):
        best = None
        for _ in range(reps):
            torch.cuda.synchronize(); t = time.perf_counter()
            r = fn()
            torch.cuda.synchronize(); dt = time.perf_counter() - t
            best = dt if best is None else min(best, dt)
        return best, r

    with torch.no_grad():
        # ---- configs[2]: 4 images + 512-token prompt, S = 512 + 4 * 258 = 1544
        g = torch.Generator().manual_seed(2)
        text_ids = torch.randint(3, 32000, (512,), generator=g)
        block = torch.tensor([IMG_TOKEN_ID] + [IMAGE_TOKEN_ID] * vcfg.n_query + [IMG_END_TOKEN_ID])
        parts = []
        for i in range(4):
            parts += [text_ids[i * 128:(i + 1) * 128], block]
        ids = torch.cat(parts)[None]
        S = ids.shape[1]
        imgs = img.expand(4, -1, -1, -1).contiguous()
        m.encode_image(imgs)                                 # first call at this batch: workspace allocation, cold code
        def enc4():                                          # steady state like the other legs: mean of 4 back-to-back calls
            for _ in range(4):
                r = m.encode_image(imgs)
            return r
        t_vit, _ = timed(enc4, reps=2)
        t_vit /= 4
        x = m._prompt_embeds(ids, imgs, vcfg.n_query)
        mask = torch.ones(1, S, dtype=torch.long)
        s_max = lm.kv_capacity(S + 8)
        t_pf, _ = timed(lambda: lm.prefill(x.view(1, S, -1), mask, s_max))
        fl = lcfg.num_hidden_layers * (2 * S * (4 * lcfg.hidden_size ** 2 + 3 * lcfg.hidden_size * lcfg.intermediate_size)
                                       + 2 * S * S * lcfg.hidden_size)
        out["prefill_fewshot_S1544"] = {"config": "BASELINE.json configs[2]: 4 images + 512-token prompt", "S": S,
                                        "prefill_ms": t_pf * 1e3, "tflops": fl / t_pf / 1e12, "mfma_frac": fl / t_pf / MFMA_BF16_PEAK,
                                        "vit_encode_4_images_ms": t_vit * 1e3,
                                        "vit_note": "ONE batched encode of the 4 images (M = 4100 rows per GEMM), mean of 4 back-to-back calls"}
        # ---- generate_image (Emu2-Gen: n_query 64), 20-token prompt: S0 + 1 prefill + 63 cached steps
        nq_saved = m.n_query
        m.n_query = 64
        try:
            pid = torch.randint(3, 32000, (1, 20), generator=g)
            t_gi, emb = timed(lambda: m.generate_image_ids(pid))
            wb = lm.weight_bytes_per_token() - 2 * lm.lm_head.numel()              # no lm_head in the regression loop
            out["generate_image"] = {"config": "emu.py:92-153, KV-cached: 21-token prefill + 63 steps (project_down -> project_up "
                                               "-> 60 layers), 64 visual embeddings out", "ms": t_gi * 1e3,
                                     "ms_per_step": t_gi * 1e3 / 64, "weight_stream_GBps": 63 * wb / t_gi / 1e9,
                                     "frac_of_hbm_peak": 63 * wb / t_gi / HBM_PEAK, "finite": bool(torch.isfinite(emb.float()).all()),
                                     "weight_bytes_per_step": wb,
                                     "roofline": {"bound": "hbm", "achieved": 63 * wb / t_gi / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                                                  "frac": 63 * wb / t_gi / HBM_PEAK, "traffic": None,
                                                  "note": "63 cached steps x the decoder's weight bytes over the WHOLE call (the 21-token "
                                                          "prefill and the 64 project_down / project_up pairs included in the time)"}}
            # ---- VAE decode + any-to-image end to end
            vcfg_ = VaeCfg()
            vae = VaeDecoder(vcfg_, ctx)
            vae.load_state_dict(synth.synth_state_dict(vae_decoder_param_shapes(vcfg_), seed=1), strict=True)
            z = torch.randn(1, 4, 128, 128, device=dev).to(torch.bfloat16)
            t_vae, _ = timed(lambda: vae.decode_latents(z))
            vf = vae_decode_flops()
            out["vae_decode"] = {"config": "AutoencoderKL.decode, latents [1,4,128,128] -> 1024x1024", "ms": t_vae * 1e3,
                                 "roofline": {"bound": "mfma", "achieved": vf / t_vae / 1e12, "peak": MFMA_BF16_PEAK / 1e12, "unit": "TFLOP/s",
                                              "frac": vf / t_vae / MFMA_BF16_PEAK, "flops": vf, "traffic": None,
                                              "note": "whole decode against the bf16 MFMA peak; the 1024^2 x 128-channel level is HBM-heavy "
                                                      "(0.5 GB of activations per conv) and GroupNorm takes three passes"}}
            if unet_eng is not None:
                neg = torch.randint(3, 32000, (1, 1), generator=g)

                # prompt and negative prompt as the pipeline's first call runs them (emu_amd/diffusion.py): two left-padded rows
                # of one batch, each on its own positions, one weight stream per step for both
                both_ids = torch.cat((pid, torch.cat((torch.full((1, 19), 32000, dtype=pid.dtype), neg), dim=1)), dim=0)
                both_mask = torch.ones(2, 20, dtype=torch.long)
                both_mask[1, :19] = 0

                def chain():
                    prompt = m.generate_image_ids(both_ids, None, both_mask).to(torch.bfloat16)      # [cond, uncond]
                    sch = unet_eng.set_timesteps(50)
                    unet_eng.set_context(prompt, 1024, 1024)
                    lat = (torch.randn(1, 4, 128, 128, device=dev) * sch.init_noise_sigma).to(torch.bfloat16).contiguous()
                    lat = unet_eng.denoise(lat, 3.0, use_graph=True)
                    return vae.decode_latents(lat)
                t_e2e, image = timed(chain)
                out["any_to_image_e2e"] = {"config": "BASELINE.json configs[4] at TP=1, bf16: generate_image of prompt + negative prompt (uncached "
                                                     "here) as two rows of one batch -> 50-step CFG denoise (hipGraph) -> VAE decode, "
                                                     "1024x1024", "ms": t_e2e * 1e3, "finite": bool(torch.isfinite(image.float()).all())}
                # the same chain on the fp8 paths that exist (extra, never a headline): the decoder's weight stream as e4m3
                # (generate_image's 63 cached two-row steps), the UNet's 70 transformer blocks W8A8; prefill, convs, VAE bf16
                if not getattr(config_legs, "no_fp8", False):
                    try:
                        lm.use_fp8(True)
                        unet_eng.use_fp8(True)
                        t_e2e8, image8 = timed(chain)
                        out["any_to_image_e2e_fp8"] = {"config": "the same chain with the decoder's weight stream in e4m3 (per-row scales; "
                                                       "generate_image's cached steps) and the UNet's transformer-block GEMMs W8A8; "
                                                       "prefill, convs, attention, VAE bf16", "ms": t_e2e8 * 1e3,
                                                       "finite": bool(torch.isfinite(image8.float()).all()),
                                                       "note": "extra leg; the reference runs bf16 end to end"}
                    except Exception as e:
                        out["any_to_image_e2e_fp8"] = {"ms": None, "note": f"failed: {e}"}
                    finally:
                        lm.use_fp8(False)
                        unet_eng.use_fp8(False)
        finally:
            m.n_query = nq_saved
    return out


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` by itself: become the launcher -- one rank process per GPU under torch.distributed.run, the
        # form the driver uses (rank 0 prints the ONE JSON line; its stdout / stderr pass through; the exit code is the job's)
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd, env=env).returncode)
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch `python bench.py --gpus N` (it spawns the ranks itself) or "
                         "`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU path)")
    if world > torch.cuda.device_count() and os.environ.get("EMU_TP_SHARED_GPU") != "1":
        raise SystemExit(f"--gpus {world} but {torch.cuda.device_count()} GPU(s) visible (EMU_TP_SHARED_GPU=1 runs all ranks on cuda:0: "
                         "validation of the data path only, never a measurement)")
    # validation hook, not a measurement: EMU_TP_SHARED_GPU=1 puts every rank on cuda:0 (RCCL refuses that, so gloo rendezvous
    # and the peer-to-peer all-reduce for every message) to run the full-size TP data path on a 1-GPU box
    shared = world > 1 and os.environ.get("EMU_TP_SHARED_GPU") == "1"
    if shared:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if shared:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    rdev = "cpu" if shared else dev

    from emu_amd import CLIPVisionCfg, LlamaCfg, TextDecoderCfg, synth
    from emu_amd._lib import lib, check
    from emu_amd.constants import IMAGE_TOKEN_ID, IMG_END_TOKEN_ID, IMG_TOKEN_ID, OPENAI_DATASET_MEAN, OPENAI_DATASET_STD, VOCAB_EMU2_CHAT
    from emu_amd.emu import EmuModel
    from emu_amd.llama import EmuHipContext, GreedyState
    from emu_amd import ops
    import ctypes as C

    ctx = EmuHipContext(dev, rank, world)
    if a.gemm_tune:
        lib().emu_gemm_tune(a.gemm_tune)
    if a.pmc_prefill:
        from emu_amd.llama import LlamaEngine
        lcfg = LlamaCfg(num_hidden_layers=a.layers)
        lm = LlamaEngine(lcfg, VOCAB_EMU2_CHAT, ctx)
        lm.load_weights(synth.iter_synth(synth.llama_param_shapes(lcfg, VOCAB_EMU2_CHAT), seed=0, device=dev, dtype=torch.bfloat16))
        S = a.prompt_tokens + 258
        x = (torch.randn(1, S, lcfg.hidden_size, device=dev) * 0.02).to(torch.bfloat16)
        with torch.no_grad():
            for _ in range(a.pmc_prefill):
                lm.prefill(x, torch.ones(1, S, dtype=torch.long), lm.kv_capacity(S + 8))
        torch.cuda.synchronize()
        print(json.dumps({"pmc_prefill_calls": a.pmc_prefill, "S": S, "gemm_algorithmic_bytes_per_prefill": prefill_gemm_bytes(lcfg, S)}),
              flush=True)
        return
    if a.only_denoise:
        denoise_leg.no_fp8 = a.no_fp8
        d = denoise_leg(ctx, dev, a.denoise_steps, world, dist if world > 1 else None, a.unet_fusion)
        if rank == 0:
            print(json.dumps(d), flush=True)
        return
    if world > 1:
        def bcast(b):
            box = [b]
            dist.broadcast_object_list(box, src=0)
            return box[0]

        def allgather(b):
            box = [None] * world
            dist.all_gather_object(box, b)
            return box
        # RCCL for the prefill-sized messages; the one-shot peer-to-peer all-reduce (csrc/p2p.hip) for the 13 KB decode ones when
        # its self-test passes on every rank (EMU_TP_P2P=0 keeps everything on RCCL)
        ctx.init_tp(bcast, allgather_bytes=allgather if os.environ.get("EMU_TP_P2P", "1") != "0" else None, rccl=not shared)
        log(f"rank {rank}: decode all-reduce = {'one-shot P2P over IPC' if ctx.p2p else 'RCCL'}")

    vcfg = CLIPVisionCfg(n_query=256, v_query=64, layers=a.vit_layers)          # Emu2-Chat: n_query 256 (chat.py:221-223)
    lcfg = LlamaCfg(num_hidden_layers=a.layers)
    t0 = time.time()
    m = EmuModel(vcfg, TextDecoderCfg(instruct=True), llama_cfg=lcfg, device=dev, ctx=ctx)
    shapes = synth.emu_param_shapes(vcfg, lcfg, VOCAB_EMU2_CHAT)
    m.load_weights(synth.iter_synth(shapes, seed=0, device=dev, dtype=torch.bfloat16), strict=True)
    torch.cuda.synchronize()
    lm = m.decoder.lm
    log(f"rank {rank}: weights ready in {time.time() - t0:.1f}s, shard bytes/token {lm.weight_bytes_per_token() / 1e9:.2f} GB, "
        f"mem {torch.cuda.memory_allocated() / 2**30:.1f} GiB")

    if os.environ.get("EMU_DECODE_TAIL") == "1":            # A/B aid: decode attention with the in-kernel split merge (one launch per layer less)
        lm.set_decode_tail(True)
    # ---- synthetic prompt: 512 random ids + [IMG] 256x<image> [/IMG]
    g = torch.Generator().manual_seed(2)
    text_ids = torch.randint(3, 32000, (a.prompt_tokens,), generator=g)
    half = a.prompt_tokens // 2
    block = torch.tensor([IMG_TOKEN_ID] + [IMAGE_TOKEN_ID] * vcfg.n_query + [IMG_END_TOKEN_ID])
    ids = torch.cat([text_ids[:half], block, text_ids[half:]])[None]
    S = ids.shape[1]
    mask = torch.ones(1, S, dtype=torch.long)
    gi = torch.Generator().manual_seed(1)
    img = torch.rand(1, 3, 448, 448, generator=gi)
    img = (img - torch.tensor(OPENAI_DATASET_MEAN)[None, :, None, None]) / torch.tensor(OPENAI_DATASET_STD)[None, :, None, None]
    img = img.to(dev)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- untimed: ViT encode + prefill (reported separately)
    with torch.no_grad():
        sync(); t = time.perf_counter()
        enc = m.encode_image(img)
        sync(); vit_ms = (time.perf_counter() - t) * 1e3
        x = m._prompt_embeds(ids, img, vcfg.n_query)
        sync(); t = time.perf_counter()
        # what generate() would allocate for this request; a run longer than the model's 2048 positions decodes in
        # segments that rewind to the end of the prompt (the decoder cannot address past max_position_embeddings)
        max_pos = lcfg.max_position_embeddings
        seg = max(1, min(a.warmup + a.steps + 16, max_pos - S - 16))
        s_max = lm.kv_capacity(S + seg + 8)
        lm.alloc_kv(1, s_max)
        sync(); t = time.perf_counter()
        hidden, kstart, next_pos = lm.prefill(x.view(1, S, -1), mask, s_max)
        sync(); prefill_ms = (time.perf_counter() - t) * 1e3
        # steady state of each (the first call includes lazy code-object loads, and a single 15 / 50 ms burst after an idle
        # synchronisation runs on a clock that is still ramping): 4 back-to-back calls, mean -- like every other leg's W + K steps
        sync(); t = time.perf_counter()
        for _ in range(4):
            m.encode_image(img)
        sync(); vit_ms2 = (time.perf_counter() - t) * 1e3 / 4
        sync(); t = time.perf_counter()
        for _ in range(4):
            hidden, kstart, next_pos = lm.prefill(x.view(1, S, -1), mask, s_max)
        sync(); prefill_ms2 = (time.perf_counter() - t) * 1e3 / 4
        logits = lm.logits(hidden[:, -1, :])
        cur = ops.argmax(logits, suppress_id=2)
    total = min(a.warmup + a.steps + 8, seg + 8)
    out_ids = torch.zeros(total + 1, 1, device=dev, dtype=torch.int32)
    out_ids[0] = cur
    st = GreedyState(lm, 1, cur, next_pos, S, kstart, out_ids)
    if a.pmc_mode:
        # everything up to here launched no M <= 8 GEMV except the logits row above; count from now
        check(lib().emu_profile_gemv(1), "emu_profile_gemv")
        with torch.no_grad():
            for _ in range(a.pmc_mode):
                st.step()
        torch.cuda.synchronize()
        ms, wb, nl = C.c_double(), C.c_double(), C.c_long()
        check(lib().emu_profile_gemv_read(C.byref(ms), C.byref(wb), C.byref(nl)), "emu_profile_gemv_read")
        if rank == 0:
            print(json.dumps({"pmc_mode_steps": a.pmc_mode, "gemv_launches": nl.value, "gemv_algorithmic_bytes": wb.value,
                              "gemv_ms": ms.value}), flush=True)
        return
    use_graph = not a.no_graph
    def make_stepper(state, graph):
        box = {"fn": state.step_graph if graph else state.step, "n": 0}

        def stepper():
            if box["n"] >= seg:                                 # out of positions: rewind to the prompt (device-side, async)
                state.reset(cur, next_pos, S)
                box["n"] = 0
            box["fn"]()
            box["n"] += 1
        return stepper, box

    step, raw_step = make_stepper(st, use_graph)
    with torch.no_grad():
        try:
            for _ in range(a.warmup):
                step()
        except Exception as e:                                  # graph capture unavailable -> eager launches
            if not use_graph:
                raise
            log(f"hipGraph capture failed ({e}); falling back to eager launches")
            use_graph = False
            raw_step["fn"] = st.step
            for _ in range(a.warmup):
                step()
        sync()
        t = time.perf_counter()
        for _ in range(a.steps):
            step()
        sync()
        dt = time.perf_counter() - t
    per_rank_ms = None
    if world > 1:
        # every rank's own clock around the same K steps (they end together: each step holds 2 x layers all-reduces), then MAX
        mine = torch.zeros(world, device=rdev, dtype=torch.float64)
        mine[rank] = dt
        dist.all_reduce(mine, op=dist.ReduceOp.SUM)
        per_rank_ms = [float(v) / a.steps * 1e3 for v in mine.tolist()]
        tt = torch.tensor([dt], device=rdev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    tok_s = a.steps / dt

    # ---- roofline leg: HIP events around every GEMV launch of the SAME steps, replayed eagerly
    check(lib().emu_profile_gemv(1), "emu_profile_gemv")
    n_prof = min(8, a.steps)
    st.reset(cur, next_pos, S)                                  # eager replay from the end of the prompt
    with torch.no_grad():
        for _ in range(n_prof):
            st.step()
    torch.cuda.synchronize()
    ms, wb, nl = C.c_double(), C.c_double(), C.c_long()
    check(lib().emu_profile_gemv_read(C.byref(ms), C.byref(wb), C.byref(nl)), "emu_profile_gemv_read")
    check(lib().emu_profile_gemv(0), "emu_profile_gemv")
    gemv_ms_per_tok = ms.value / n_prof
    bytes_per_launch = wb.value / max(1, nl.value)
    avg_launch_s = ms.value * 1e-3 / max(1, nl.value)
    achieved = bytes_per_launch / avg_launch_s
    ctx_mid = S + a.warmup + a.steps // 2
    kv_bytes = 2 * lcfg.num_hidden_layers * (lm.plan.heads_local * lcfg.head_dim) * 2 * ctx_mid
    ids_host = out_ids[: min(a.warmup + a.steps, total) + 1, 0].tolist()

    # ---- extra leg (never the headline): same decode loop over the fp8 e4m3 weight stream (BASELINE.json configs[5]'s
    # weight-only quantised serving mode); prefill + KV cache stay bf16
    fp8 = None
    if not a.no_fp8 and world == 1:                             # extra legs run on one GPU only: an N > 1 run carries the headline and nothing that could stall it
        try:
            t0 = time.time()
            lm.use_fp8(True)
            torch.cuda.synchronize()
            q_s = time.time() - t0
            # prefill on the fp8 weight set: activations quantised per row ahead of every GEMM, block-scaled fp8 MFMA
            with torch.no_grad():
                lm.use_fp8(True, prefill=True)
                lm.prefill(x.view(1, S, -1), mask, s_max)
                sync(); t = time.perf_counter()
                lm.prefill(x.view(1, S, -1), mask, s_max)
                sync(); pf8 = time.perf_counter() - t
                lm.use_fp8(True)                              # the decode leg below: weight-only fp8 stream, bf16 activations
                # the encoder blocks W8A8 (round 4): same image, tokens against the bf16 encoder's
                vit8 = None
                if world == 1:
                    tok16 = m.visual(img).clone()
                    m.visual.use_fp8(True)
                    try:
                        tok8 = m.visual(img)
                        sync(); t = time.perf_counter(); m.visual(img); sync()
                        vit8 = {"ms": (time.perf_counter() - t) * 1e3,
                                "rel_l2_vs_bf16_tokens": float((tok8.float() - tok16.float()).norm() / tok16.float().norm())}
                    finally:
                        m.visual.use_fp8(False)
                    sync(); t = time.perf_counter(); m.visual(img); sync()
                    vit8["bf16_ms_same_call"] = (time.perf_counter() - t) * 1e3
                    vit8["note"] = ("64 EVA-CLIP-4B blocks with W8A8 GEMMs (per-row e4m3 on weights and activation rows: the LayerNorms emit "
                                    "their rows as fp8, the attention output and the GELU product are quantised by a launch each), "
                                    "LayerNorm / attention / stem arithmetic bf16; random-init weights: the distance "
                                    "is what per-row e4m3 costs on unstructured matrices, not a quality claim")
            out8 = torch.zeros(total + 1, 1, device=dev, dtype=torch.int32)
            out8[0] = cur
            st8 = GreedyState(lm, 1, cur, next_pos, S, kstart, out8)
            step8, _ = make_stepper(st8, use_graph)
            with torch.no_grad():
                for _ in range(a.warmup):
                    step8()
                sync(); t = time.perf_counter()
                for _ in range(a.steps):
                    step8()
                sync(); dt8 = time.perf_counter() - t
                if world > 1:
                    tt = torch.tensor([dt8], device=rdev, dtype=torch.float64)
                    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                    dt8 = float(tt.item())
                check(lib().emu_profile_gemv(1), "emu_profile_gemv")
                st8.reset(cur, next_pos, S)
                for _ in range(n_prof):
                    st8.step()
                torch.cuda.synchronize()
                check(lib().emu_profile_gemv_read(C.byref(ms), C.byref(wb), C.byref(nl)), "emu_profile_gemv_read")
                check(lib().emu_profile_gemv(0), "emu_profile_gemv")
            ids8 = out8[: min(a.warmup + a.steps, total) + 1, 0].tolist()
            agree = 0
            for x0, x1 in zip(ids_host, ids8):
                if x0 != x1:
                    break
                agree += 1
            fp8 = {"value": a.steps / dt8, "unit": "tokens/s", "ms_per_step": dt8 / a.steps * 1e3, "dtype": "fp8 e4m3 weights "
                   "(per-row scale), bf16 activations/KV, fp32 accumulate", "quantise_s": q_s,
                   "weight_bytes_per_token_per_gpu": lm.weight_bytes_per_token(),
                   "gemv_achieved_GBps": wb.value / (ms.value * 1e-3) / 1e9, "gemv_frac_of_hbm_peak": wb.value / (ms.value * 1e-3) / HBM_PEAK,
                   "gemv_ms_per_token": ms.value / n_prof, "tokens_identical_to_bf16_prefix": agree,
                   "prefill_ms": pf8 * 1e3, "vit_encode_fp8": vit8, "prefill_note": "S=%d prefill with W8A8 GEMMs on v_mfma_scale_f32_32x32x64_f8f6f4 "
                   "(per-row e4m3 scales on weights and activations), attention / norms / KV bf16" % S,
                   "note": "extra leg, not the headline metric (which stays bf16 like the reference)"}
            kv8 = 2 * lcfg.num_hidden_layers * lcfg.hidden_size * 2 * (S + a.warmup + a.steps // 2) // world
            fp8["roofline"] = {"bound": "hbm", "kernel": "gemv_fp8* (e4m3 weight-streaming GEMV)", "achieved": wb.value / (ms.value * 1e-3) / 1e9,
                               "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": wb.value / (ms.value * 1e-3) / HBM_PEAK, "traffic": None,
                               "bytes_per_launch": wb.value / max(1, nl.value), "avg_launch_us": ms.value * 1e3 / max(1, nl.value),
                               "token_level_frac": (lm.weight_bytes_per_token() + kv8) * (a.steps / dt8) / HBM_PEAK,
                               "measured": f"HIP events around each fp8 GEMV, eager replay of {n_prof} steps"}
            pfl8 = lcfg.num_hidden_layers * (2 * S * (4 * lcfg.hidden_size ** 2 + 3 * lcfg.hidden_size * lcfg.intermediate_size)
                                             + 2 * S * S * lcfg.hidden_size)
            fp8["prefill_roofline"] = {"bound": "mfma", "achieved": pfl8 / world / pf8 / 1e12, "peak": MFMA_FP8_PEAK / 1e12, "unit": "TFLOP/s",
                                       "frac": pfl8 / world / pf8 / MFMA_FP8_PEAK, "traffic": None,
                                       "note": "W8A8 GEMMs priced against the dense fp8 peak; attention, norms and the row quantisers run bf16 / fp32"}
            if isinstance(vit8, dict) and vit8.get("ms"):
                vit8["roofline"] = {"bound": "mfma", "achieved": VIT_FLOPS_PER_IMAGE / (vit8["ms"] * 1e-3) / 1e12, "peak": MFMA_FP8_PEAK / 1e12,
                                    "unit": "TFLOP/s", "frac": VIT_FLOPS_PER_IMAGE / (vit8["ms"] * 1e-3) / MFMA_FP8_PEAK, "traffic": None}
        except Exception as e:
            fp8 = {"value": None, "note": f"fp8 leg failed: {e}"}
        finally:
            lm.use_fp8(False) if getattr(lm, "_fp8", None) else None

    # ---- extra leg (never the headline): the reference's DEFAULT decoding mode, 5-beam search (emu.py:163-172), whole call
    # (prefill + KV replication + 24 beam steps, 5 activation rows per weight stream); TP=1 only
    beam = None
    if not a.no_beam and world == 1:
        try:
            n_new = 24
            with torch.no_grad():
                lm.beam_search_generate(x.view(1, S, -1), mask, 5, 2, min_len=2)        # warm: cache allocation, lazy loads
                torch.cuda.synchronize(); t = time.perf_counter()
                out = lm.beam_search_generate(x.view(1, S, -1), mask, 5, n_new, min_len=n_new)   # first call of this signature: captures
                torch.cuda.synchronize(); dt_first = time.perf_counter() - t
                torch.cuda.synchronize(); t = time.perf_counter()
                out = lm.beam_search_generate(x.view(1, S, -1), mask, 5, n_new, min_len=n_new)   # the step graph is re-used
                torch.cuda.synchronize(); dtb = time.perf_counter() - t
                lm.beam_graph = False
                torch.cuda.synchronize(); t = time.perf_counter()
                lm.beam_search_generate(x.view(1, S, -1), mask, 5, n_new, min_len=n_new)         # the same launches, eager
                torch.cuda.synchronize(); dt_eager = time.perf_counter() - t
                lm.beam_graph = True
                torch.cuda.synchronize(); t = time.perf_counter()
                lm.prefill(x.view(1, S, -1), mask, lm.kv_capacity(S + n_new))
                torch.cuda.synchronize(); dpf = time.perf_counter() - t
            steps_b = max(1, out.shape[1] - 1)
            beam = {"num_beams": 5, "new_tokens": int(out.shape[1]), "call_ms": dtb * 1e3,
                    "ms_per_beam_step": (dtb - dpf) * 1e3 / steps_b, "tokens_per_s": steps_b / max(1e-9, dtb - dpf),
                    "first_call_ms_incl_graph_capture": dt_first * 1e3,
                    "ms_per_beam_step_eager_launches": (dt_eager - dpf) * 1e3 / steps_b,
                    "note": "5 rows per weight stream (LDS-DMA stages + 16x16x32 MFMA, gemv_thin.hip), the prompt's KV kept once per prompt "
                            "and read by all beams (emu_llama_set_kv_share); round 4: the whole step -- slot / position bookkeeping, re-order of "
                            "the generated KV slots, embedding gather, 60 layers on 5 rows, logits, 2N-best selection and scorer bookkeeping "
                            "(emu_beam_step_bf16) -- reads its step index on the device and is replayed from one hipGraph, the host looks at the "
                            "done flags every 4 steps; the graph is captured once per (prompt length, beams, limits): first_call includes "
                            "that; prefill excluded from the per-step time"}
        except Exception as e:
            beam = {"num_beams": 5, "note": f"beam leg failed: {e}"}

    # ---- second half of the metric: SDXL-style UNet denoise (BASELINE.json configs[3]), replicas only across GPUs
    denoise = None
    if not a.no_denoise:
        # across GPUs this leg runs REPLICAS: every rank times its own loop with no collective inside (a rank that fails
        # must not leave the others in a barrier), then one all_gather_object that every rank reaches
        denoise_leg.no_fp8 = a.no_fp8 or world > 1
        try:
            denoise = denoise_leg(ctx, dev, a.denoise_steps, 1, None, a.unet_fusion)
        except Exception as e:
            denoise = {"value": None, "per_gpu": None, "note": f"denoise leg failed on rank {rank}: {type(e).__name__}: {e}"}
        if world > 1:
            box = [None] * world
            dist.all_gather_object(box, {"per_gpu": denoise.get("per_gpu"), "ms_per_step": denoise.get("ms_per_step")})
            rates = [b.get("per_gpu") for b in box]
            denoise["per_rank_ms_per_step"] = [b.get("ms_per_step") for b in box]
            denoise["value"] = sum(r for r in rates if r) if all(rates) else None
            denoise["scaling"] = "replicas only (independent images per GPU; value = sum over the ranks' own rates)"

    legs = None
    if not a.no_legs and world == 1:
        try:
            config_legs.no_fp8 = a.no_fp8
            legs = config_legs(m, lm, ctx, dev, vcfg, lcfg, img, getattr(denoise_leg, "engine", None))
        except Exception as e:                                  # never lose the headline to an extra leg
            legs = {"note": f"legs failed: {type(e).__name__}: {e}"}

    tp_prefill = None
    if world > 1 and a.tp_prefill_leg >= 1024:
        # opt-in (see --tp-prefill-leg): the same S-row prompt under both prefill schedules, every rank's clock, MAX over ranks
        try:
            Sx = a.tp_prefill_leg
            xx = (torch.randn(1, Sx, lcfg.hidden_size, device=dev) * 0.1).to(torch.bfloat16)
            mm = torch.ones(1, Sx, dtype=torch.long, device=dev)
            keep = lm.tp_overlap_rows
            tp_prefill = {"S": Sx, "allreduce_bytes_per_prefill_per_rank": 2 * lcfg.num_hidden_layers * Sx * lcfg.hidden_size * 2}
            with torch.no_grad():
                capx = lm.kv_capacity(Sx + 8)
                for name, rows in (("serial", 0), ("two_lane", 1024)):
                    lm.set_tp_overlap(rows)
                    n0 = lm.tp_overlap_count()
                    lm.prefill(xx, mm, capx)
                    sync()
                    t0 = time.perf_counter()
                    for _ in range(3):
                        lm.prefill(xx, mm, capx)
                    sync()
                    tt = torch.tensor([(time.perf_counter() - t0) / 3 * 1e3], device=rdev, dtype=torch.float64)
                    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                    tp_prefill[name + "_ms"] = float(tt.item())
                    tp_prefill[name + "_forwards_on_two_lanes"] = lm.tp_overlap_count() - n0
            lm.set_tp_overlap(keep)
            ctx.check_p2p()
        except Exception as e:                                  # never lose the headline to an extra leg
            tp_prefill = {"note": f"leg failed: {type(e).__name__}: {e}"}

    # HBM traffic of the GEMV launches comes from PMC counters (FETCH_SIZE), which a timing run cannot collect itself:
    # the committed pass over THIS command (profiles/r03_gemv_pmc_traffic.json, gfx950-corrected) gives traffic /
    # algorithmic bytes for the same kernels; traffic = that ratio x this run's algorithmic bytes per launch
    traffic, traffic_src = None, "no PMC pass over the GEMV sources committed"
    for rnd in ("r06", "r05", "r03"):                               # the newest committed pass whose kernel sources are these
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", f"{rnd}_gemv_pmc_traffic.json")))
        except Exception:
            continue
        if pmc.get("source_sha256") == gemv_source_hash():
            ratio = float(pmc["all_gemv_launches"]["traffic_over_algorithmic"])
            traffic = ratio * bytes_per_launch
            traffic_src = (f"profiles/{rnd}_gemv_pmc_traffic.json (same kernel sources, sha256 checked): rocprofv3 --pmc FETCH_SIZE "
                           f"pass over `bench.py --pmc-mode`, gfx950-corrected; traffic / algorithmic = {ratio:.4f}")
            break
        traffic_src = f"profiles/{rnd}_gemv_pmc_traffic.json is STALE (taken on other kernel sources: re-run tools/pmc_traffic.sh); traffic not reported"

    prefill_traffic, prefill_traffic_src = None, "no PMC pass over the GEMM sources committed"
    for rnd in ("r06", "r05", "r04", "r03"):                        # the newest committed pass whose kernel sources are these
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", f"{rnd}_prefill_gemm_pmc_traffic.json")))
        except Exception:
            continue
        if pm.get("source_sha256") == gemm_source_hash():
            prefill_traffic = float(pm["traffic_over_algorithmic"]) * prefill_gemm_bytes(lcfg, S)
            prefill_traffic_src = (f"profiles/{rnd}_prefill_gemm_pmc_traffic.json (same kernel sources, sha256 checked): traffic / algorithmic GEMM "
                                   f"bytes = {float(pm['traffic_over_algorithmic']):.3f}")
            break
        prefill_traffic_src = f"profiles/{rnd}_prefill_gemm_pmc_traffic.json is STALE (other kernel sources): traffic not reported"
    if rank == 0:
        for what, val, src in (("roofline.traffic", traffic, traffic_src), ("prefill_roofline.traffic", prefill_traffic, prefill_traffic_src)):
            if val is None:                                     # loud, but never at the price of the line itself
                log(f"WARNING: {what} is not reported -- {src}.  Re-run tools/pmc_traffic.sh / tools/pmc_prefill_traffic.sh on the final sources and "
                    "commit their output under profiles/.")
        prefill_flops = lcfg.num_hidden_layers * (2 * S * (4 * lcfg.hidden_size ** 2 + 3 * lcfg.hidden_size * lcfg.intermediate_size)
                                                   + 2 * S * S * lcfg.hidden_size)
        res = {
            "metric": "decoded text tokens/sec, Emu2-37B greedy decode (LLaMA-33B, KV cache), TP=%d" % world,
            "value": tok_s, "unit": "tokens/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[1]: Emu2-Chat 37B text generate, 1x448x448 image "
                                   f"(256 visual tokens) + {a.prompt_tokens}-token prompt (S={S}), greedy, batch 1",
                       "decoder_layers": lcfg.num_hidden_layers, "vit_layers": vcfg.layers,
                       "parallelism": f"tp{world}" + (" (ranks sharing one GPU: validation only)" if shared else ""), "allreduce": ("p2p one-shot (<=256 KiB) + rccl" if ctx.p2p else "rccl") if world > 1 else None,
                       "tp": ({"ms_per_token_by_rank": per_rank_ms, "per_rank_ms_per_token": per_rank_ms, "allreduces_per_token": 2 * lcfg.num_hidden_layers,
                               "rccl_ranks": 0 if shared else world,
                               "p2p_form": ("fence-free" if getattr(ctx, "p2p_fence_free", False) else "fenced") if ctx.p2p else None,
                               "allreduce_bytes": 2 * lcfg.hidden_size, "weight_bytes_per_token_per_rank": lm.weight_bytes_per_token(),
                               "prefill_overlap_min_rows": lm.tp_overlap_rows, "prefill_overlap_forwards": lm.tp_overlap_count(), "prefill_schedules": tp_prefill,
                               "note": "one process per GPU; decode: o_proj / down_proj partial sums all-reduced in place on the launch stream (one-shot "
                                       "peer-to-peer kernel; its fence-free form only after a soak passed on every rank of this job); prefill: serial "
                                       "all-reduce schedule unless prefill_overlap_min_rows > 0 (opt-in two-lane schedule, EMU_TP_OVERLAP)"}
                              if world > 1 else None),
                       "launch": "hipGraph replay" if use_graph else "eager",
                       "valid": bool(a.layers == 60 and a.vit_layers == 64 and not shared and not a.gemm_tune)},
            "roofline": {"bound": "hbm", "kernel": "gemv_kernel (weight-streaming GEMV, all epilogues)",
                         "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK, "traffic": traffic, "traffic_source": traffic_src,
                         "bytes_per_launch": bytes_per_launch, "avg_launch_us": avg_launch_s * 1e6,
                         "launches_per_token": nl.value / n_prof, "gemv_ms_per_token": gemv_ms_per_tok,
                         "measured": f"HIP events on the launch stream around each GEMV, eager replay of {n_prof} decode steps",
                         "token_level_frac": (lm.weight_bytes_per_token() + kv_bytes) * tok_s / HBM_PEAK},
            "extra": {"vit_encode_ms": min(vit_ms, vit_ms2), "prefill_ms": min(prefill_ms, prefill_ms2),
                      "prefill_roofline": {"bound": "mfma", "achieved": prefill_flops / world / (min(prefill_ms, prefill_ms2) * 1e-3) / 1e12,
                                           "peak": MFMA_BF16_PEAK / 1e12, "unit": "TFLOP/s",
                                           "frac": prefill_flops / world / (min(prefill_ms, prefill_ms2) * 1e-3) / MFMA_BF16_PEAK,
                                           "flops": prefill_flops / world, "traffic": prefill_traffic,
                                           "traffic_source": prefill_traffic_src, "algorithmic_gemm_bytes": prefill_gemm_bytes(lcfg, S)},
                      "vit_roofline": {"bound": "mfma", "achieved": VIT_FLOPS_PER_IMAGE / (min(vit_ms, vit_ms2) * 1e-3) / 1e12,
                                       "peak": MFMA_BF16_PEAK / 1e12, "unit": "TFLOP/s",
                                       "frac": VIT_FLOPS_PER_IMAGE / (min(vit_ms, vit_ms2) * 1e-3) / MFMA_BF16_PEAK,
                                       "flops": VIT_FLOPS_PER_IMAGE, "traffic": None},
                      "prefill_tflops": prefill_flops / world / (min(prefill_ms, prefill_ms2) * 1e-3) / 1e12,
                      "prefill_mfma_frac": prefill_flops / world / (min(prefill_ms, prefill_ms2) * 1e-3) / MFMA_BF16_PEAK,
                      "weight_bytes_per_token_per_gpu": lm.weight_bytes_per_token(), "kv_bytes_per_token_per_gpu": kv_bytes,
                      "first_tokens": ids_host[:8]},
        }
        if fp8 is not None:
            res["decode_fp8_weights"] = fp8
        if beam is not None:
            res["beam_search_5"] = beam
        if denoise is not None:
            res["denoise"] = denoise
        if legs is not None:
            res["legs"] = legs
        if world == 1 and not a.no_cpu_baseline:
            try:
                res["cpu_baseline"] = cpu_baseline(a.cpu_seconds, S, VOCAB_EMU2_CHAT)
            except Exception as e:                              # never lose the GPU line to a host-side problem
                res["cpu_baseline"] = {"value": None, "unit": "tokens/s", "cores": os.cpu_count(), "kind": "port",
                                       "sample": f"failed: {e}"}
            # the other legs of the metric beside their CPU path: bounded samples, same thread count
            try:
                th = int(res["cpu_baseline"].get("cores") or min(os.cpu_count() or 1, 64))
                cl = cpu_baseline_legs(a.cpu_seconds, th, S)
                if denoise is not None:
                    res["denoise"]["cpu_baseline"] = cl["denoise"]
                res["extra"]["vit_cpu_baseline"] = cl["vit"]
                res["extra"]["prefill_cpu_baseline"] = cl["prefill"]
                if legs is not None and "vae_decode" in legs:
                    legs["vae_decode"]["cpu_baseline"] = cl["vae"]
                lay = res["cpu_baseline"].get("layer_ms")
                if legs is not None and "generate_image" in legs and lay:
                    # the KV-cached form on the host: 63 single-row steps of 60 layers (the decode baseline's measured layer; its
                    # 770-position cache over-prices the 21..84-position attention by < 1 %) + the 21-row prefill at >= one step's cost
                    legs["generate_image"]["cpu_baseline"] = {
                        "value": 64 * 60 * lay, "unit": "ms", "cores": th, "kind": res["cpu_baseline"].get("kind", "port"),
                        "higher_is_better": False,
                        "sample": f"64 x 60 x the decode baseline's measured decoder layer ({lay:.1f} ms) = the KV-cached form of emu.py:92-153 "
                                  "(63 cached steps + a 21-row prefill counted as one step); the reference's own uncached algorithm "
                                  "(64 forwards over a growing sequence, 215.9 TFLOP) would cost ~50 x that"}
            except Exception as e:
                res["extra"]["legs_cpu_baseline_note"] = f"failed: {type(e).__name__}: {e}"
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
