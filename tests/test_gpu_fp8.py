"""GPU parity tests of the optional fp8 (OCP e4m3fn, per-row scale) decode weight stream.

Not a reference feature (the reference runs bf16 end to end); the checker is the CPU oracle fed with the exact
de-quantised weights the stream uses, so the tests pin (a) the quantiser bit-exactly against torch's CPU
``float8_e4m3fn`` cast, (b) the fp8 GEMV with its fused RMSNorm / residual / SwiGLU epilogues, and (c) a cached
decode step + greedy generation through the engine with the stream switched on.
"""
import numpy as np
import pytest
import torch

from tests import tiny

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16


def rel_err(got, want):
    got, want = got.float().cpu(), want.float().cpu()
    return float((got - want).norm() / want.norm().clamp_min(1e-12))


def cpu_quant_rows(w: torch.Tensor):
    """The quantiser's definition on CPU: scale = amax/448 (1 for a zero row), q = rne_e4m3fn(w / scale)."""
    w = w.to(BF16).float()
    amax = w.abs().amax(dim=1)
    scale = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax))
    q = (w / scale[:, None]).to(torch.float8_e4m3fn)
    return q, scale


def fake_quant(w: torch.Tensor) -> torch.Tensor:
    q, s = cpu_quant_rows(w)
    return q.float() * s[:, None]


def bfr(x):
    return x.to(BF16).float()


@pytest.mark.parametrize("N,K", [(64, 256), (37, 512), (8, 6656)])
def test_quantiser_bit_exact(N, K):
    from emu_amd import ops
    g = torch.Generator().manual_seed(N * 7 + K)
    w = (torch.randn(N, K, generator=g) * 0.05).to(BF16)
    w[0, :5] = torch.tensor([0.0, -0.0, 1e-6, -3e-5, 2.0]).to(BF16)      # denormal range + the row maximum
    if N > 2:
        w[2] = 0                                                           # all-zero row -> scale 1
    q, sc = ops.quantize_fp8_rows(w.cuda())
    qr, sr = cpu_quant_rows(w)
    assert torch.equal(sc.cpu(), sr)
    got = q.cpu().view(torch.float8_e4m3fn).float()
    assert torch.equal(got, qr.float())                                    # value-exact (+0 / -0 compare equal)


@pytest.mark.parametrize("M", [1, 2])
@pytest.mark.parametrize("N,K", [(512, 256), (1000, 6656), (4100, 512), (32274, 256)])
def test_linear_fp8w_plain_and_resid(M, N, K):
    from emu_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    w = (torch.randn(N, K, generator=g) * 0.03).to(BF16)
    x = (torch.randn(M, K, generator=g)).to(BF16)
    res = (torch.randn(M, N, generator=g)).to(BF16)
    q, sc = ops.quantize_fp8_rows(w.cuda())
    wd = q.cpu().view(torch.float8_e4m3fn).float() * sc.cpu()[:, None]
    want = bfr(x.float() @ wd.T)
    got = ops.linear_fp8w(x.cuda(), q, sc)
    assert rel_err(got, want) < 4e-3
    got = ops.linear_fp8w(x.cuda(), q, sc, res=res.cuda(), epi=ops.EPI_RESID)
    assert rel_err(got, bfr(want + res.float())) < 4e-3


@pytest.mark.parametrize("M", [1, 2])
def test_linear_fp8w_fused_norm_and_swiglu(M):
    from emu_amd import ops
    N, K, eps = 2 * 1120, 6656, 1e-5
    g = torch.Generator().manual_seed(11 + M)
    w = (torch.randn(N, K, generator=g) * 0.02).to(BF16)
    x = (torch.randn(M, K, generator=g) * 2).to(BF16)
    nw = (1 + 0.1 * torch.randn(K, generator=g)).to(BF16)
    q, sc = ops.quantize_fp8_rows(w.cuda())
    wd = q.cpu().view(torch.float8_e4m3fn).float() * sc.cpu()[:, None]
    xf = x.float()
    xn = bfr(nw.float() * bfr(xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)))
    y = bfr(xn @ wd.T)
    got = ops.linear_fp8w(x.cuda(), q, sc, norm_w=nw.cuda(), eps=eps)
    assert rel_err(got, y) < 5e-3
    gate, up = y[:, 0::2], y[:, 1::2]                                      # packed rows interleave gate/up
    want = bfr(bfr(torch.nn.functional.silu(gate)) * up)
    got = ops.linear_fp8w(x.cuda(), q, sc, norm_w=nw.cuda(), eps=eps, epi=ops.EPI_SWIGLU)
    assert got.shape == (M, N // 2)
    assert rel_err(got, want) < 8e-3


def test_fp8_rejects_prefill_rows_and_bad_k():
    from emu_amd import ops
    from emu_amd._lib import EmuHipError
    w = torch.zeros(64, 256, dtype=BF16, device="cuda")
    q, sc = ops.quantize_fp8_rows(w)
    with pytest.raises(EmuHipError):
        ops.linear_fp8w(torch.zeros(3, 256, dtype=BF16, device="cuda"), q, sc)           # > 2 rows: bf16 GEMM territory
    q8 = torch.zeros(64, 72, dtype=torch.uint8, device="cuda")
    with pytest.raises(EmuHipError):
        ops.linear_fp8w(torch.zeros(1, 72, dtype=BF16, device="cuda"), q8, sc)           # K % 16 != 0


@pytest.fixture(scope="module")
def tiny_fp8(golden_dir):
    from emu_amd import EmuModel, TextDecoderCfg
    from oracle import emu2_ref as R
    z = tiny.load(golden_dir, "generate_tiny.npz")
    v, l, vocab, W = tiny.weights_from(z)
    m = EmuModel(v, TextDecoderCfg(instruct=True), llama_cfg=l, device="cuda")
    m.load_state_dict(W, strict=True)
    W = R.bf16_round(W)
    W8 = dict(W)
    for k, t in W.items():                       # per-row scales commute with the row packing (concat / interleave)
        if k.startswith("decoder.lm.") and t.dim() == 2 and "embed_tokens" not in k:
            W8[k] = fake_quant(t)
    return m, W, W8, tiny.oracle_cfg(v, l, vocab)


def test_engine_dequantised_weights_match_cpu_definition(tiny_fp8):
    m, W, W8, cfg = tiny_fp8
    lm = m.decoder.lm
    lm.quantize_fp8()
    assert torch.equal(lm.fp8_dequantized("lm_head").cpu(), W8["decoder.lm.lm_head.weight"])
    wo = lm.fp8_dequantized("0.wo").cpu()
    assert torch.equal(wo, W8["decoder.lm.model.layers.0.self_attn.o_proj.weight"])


def test_fp8_decode_step_matches_oracle(tiny_fp8):
    """bf16 prefill, then ONE cached step through the fp8 stream == oracle step on the de-quantised weights."""
    from oracle import emu2_ref as R
    m, W, W8, cfg = tiny_fp8
    lm = m.decoder.lm
    g = torch.Generator().manual_seed(5)
    S = 24
    x = (torch.randn(2, S + 1, cfg.llama.hidden, generator=g) * 0.5).to(BF16)
    mask = torch.ones(2, S, dtype=torch.long)
    cache = R.KVCache(cfg.llama.layers)
    xf = x.float()
    R.llama_model(xf[:, :S].to(BF16), mask, R.cast_weights(W, BF16), cfg.llama, cache=cache, final_norm=False)
    mask1 = torch.ones(2, S + 1, dtype=torch.long)
    want = R.llama_model(xf[:, S:], mask1, R.cast_weights(W8, torch.float32), cfg.llama, cache=_f32(cache),
                         final_norm=False)[:, 0]
    try:
        lm.use_fp8(True)
        _, kstart, pos = lm.prefill(x[:, :S].contiguous().cuda(), mask)
        got = lm.decode_embeds(x[:, S].contiguous().cuda(), pos, S, kstart)
        assert rel_err(got, want) < 2e-2
        logits = lm.logits(got)
        h = R.rms_norm(want, W["decoder.lm.model.norm.weight"].float(), cfg.llama.rms_eps)
        wl = h @ W8["decoder.lm.lm_head.weight"].T
        assert rel_err(logits, wl) < 2.5e-2
    finally:
        lm.use_fp8(False)


def _f32(cache):
    cache.k = [None if t is None else t.float() for t in cache.k]
    cache.v = [None if t is None else t.float() for t in cache.v]
    return cache


def test_fp8_greedy_graph_equals_eager_and_tracks_oracle(tiny_fp8, golden_dir):
    """Greedy generation with the fp8 stream: hipGraph replay == eager, and ids follow the oracle run on the same
    mixed weights (bf16 prefill, fp8 head + decode) up to the first low-margin step."""
    from oracle import emu2_ref as R
    m, W, W8, cfg = tiny_fp8
    lm = m.decoder.lm
    z = tiny.load(golden_dir, "generate_tiny.npz")
    ids, mask = torch.from_numpy(z["ids2"]), torch.from_numpy(z["mask2"])
    n_new = 6
    # oracle: prefill on W (bf16), every lm_head + later steps on W8
    Wb, W8f = R.cast_weights(W, BF16), R.cast_weights(W8, torch.float32)
    emb = R.embed_tokens(ids, Wb)
    cache = R.KVCache(cfg.llama.layers)
    am = mask.clone()
    pos = (am.long().cumsum(-1) - 1).masked_fill(am == 0, 1)
    h = R.llama_model(emb, am, Wb, cfg.llama, position_ids=pos, cache=cache).float()
    _f32(cache)
    want, margins = [], []
    for step in range(n_new):
        logits = h[:, -1] @ W8f["decoder.lm.lm_head.weight"].T
        if step < 1:
            logits[:, R.EOS_ID] = -float("inf")
        t2 = logits.topk(2, dim=-1).values
        margins.append(float((t2[:, 0] - t2[:, 1]).min()))
        nxt = logits.argmax(-1)
        want.append(nxt)
        am = torch.cat((am, torch.ones(am.shape[0], 1, dtype=am.dtype)), dim=1)
        pos = pos[:, -1:] + 1
        h = R.llama_model(R.embed_tokens(nxt[:, None], W8f), am, W8f, cfg.llama, position_ids=pos, cache=cache)
    want = torch.stack(want, dim=1)
    try:
        lm.use_fp8(True)
        m.use_graph = False
        eager = m.generate_ids(ids, mask, None, max_new_tokens=n_new, stop_on_eos=False)
        m.use_graph = True
        graph = m.generate_ids(ids, mask, None, max_new_tokens=n_new, stop_on_eos=False)
    finally:
        lm.use_fp8(False)
    assert eager.cpu().tolist() == graph.cpu().tolist()
    # ids must follow the oracle; a divergence is only legitimate at a near-tie of the oracle's top-2 logits (after
    # which the two runs see different prefixes and are no longer comparable)
    got = eager.cpu()
    for i in range(n_new):
        if got[:, i].tolist() != want[:, i].tolist():
            assert margins[i] < 0.08, f"diverged from the oracle at step {i} with top-2 margin {margins[i]:.3f}"
            break
    bf16_ids = m.generate_ids(ids, mask, None, max_new_tokens=n_new, stop_on_eos=False)
    assert bf16_ids.cpu().tolist() == z["new2"].tolist()                    # switching back restores the bf16 stream


def test_fp8_weight_stream_acceptance_criterion_on_the_margin_fixture(tiny_fp8, golden_dir):
    """The STATED acceptance criterion of the e4m3 weight-stream decode mode (BASELINE configs[4]; not a reference feature), on the
    real-reference fixture whose prompts were screened for top-2 margins: with the bf16 run's tokens teacher-forced into both streams,
    (i) the logits of every step stay within relative L2 0.08 of the bf16 stream's (measured 0.026 at the first step, 0.056 at the
    eighth), (ii) the arg-max agrees on at least 80 % of the positions (measured 14 of 16), and (iii) no decision flips whose bf16 top-2
    margin exceeds the largest logit change of its row -- i.e. only near-ties move (the two that do have margins 0.25 / 0.125 against
    changes of 0.60 / 0.56).  Free-running, the first tokens of both rows are those of the bf16 stream."""
    m, W, W8, cfg = tiny_fp8
    lm = m.decoder.lm
    z = tiny.load(golden_dir, "generate_tiny.npz")
    ids, mask = torch.from_numpy(z["ids2"]), torch.from_numpy(z["mask2"])
    n_new, B, S = 8, ids.shape[0], ids.shape[1]
    teacher = m.generate_ids(ids, mask, None, max_new_tokens=n_new, stop_on_eos=False).cpu()
    x = m._prompt_embeds(ids, None, m.n_query)

    def run(fp8):
        lm.use_fp8(fp8)
        try:
            hidden, kstart, pos = lm.prefill(x.view(B, S, -1), mask)
            out = [lm.logits(hidden[:, -1, :].contiguous()).float().cpu()]
            for i in range(n_new - 1):
                e = lm.embed_tokens(teacher[:, i:i + 1].cuda()).view(B, -1)
                out.append(lm.logits(lm.decode_embeds(e, pos + i, S + i, kstart)).float().cpu())
        finally:
            lm.use_fp8(False)
        return torch.stack(out, 1)
    lm.quantize_fp8()
    lb, lf = run(False), run(True)
    d = lf - lb
    rel = [float(d[:, i].norm() / lb[:, i].norm()) for i in range(n_new)]
    assert max(rel) < 0.08, rel
    agree = lf.argmax(-1) == lb.argmax(-1)
    assert float(agree.float().mean()) >= 0.8, agree.tolist()
    t2 = lb.topk(2, -1).values
    margin, change = t2[..., 0] - t2[..., 1], d.abs().amax(-1)
    assert bool(((margin <= change) | agree).all()), (margin.tolist(), change.tolist(), agree.tolist())
    try:
        lm.use_fp8(True)
        free = m.generate_ids(ids, mask, None, max_new_tokens=n_new, stop_on_eos=False).cpu()
    finally:
        lm.use_fp8(False)
    assert free[:, :4].tolist() == teacher[:, :4].tolist()


# ------------------------------------------------------------------------------------------------ fp8 x fp8 MFMA GEMM
@pytest.mark.parametrize("M,N,K,epi", [(770, 2560, 6656, 0), (770, 1024, 17920, 1), (1025, 1536, 1792, 4), (2048, 2560, 1280, 5),
                                       (300, 520, 384, 0), (256, 256, 128, 2), (1544, 4096, 6656, 2), (64, 512, 256, 1)])
def test_fp8_mfma_gemm_matches_dequantised_reference(M, N, K, epi):
    """emu_linear_fp8_bf16 (v_mfma_scale_f32_32x32x64_f8f6f4 on the 256x256 ping-pong tile): both operands quantised per row
    by the library's quantiser; the checker is a torch fp32 GEMM on the exactly de-quantised operands (fp8 x fp8 products
    are exact in fp32, so only the accumulation order differs), then the epilogue with the reference's bf16 rounding."""
    import torch.nn.functional as F
    from emu_amd import ops
    g = torch.Generator(device="cuda").manual_seed(7)
    x = (torch.randn(M, K, device="cuda", generator=g)).to(BF16)
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.02).to(BF16)
    bias = (torch.randn(N, device="cuda", generator=g)).to(BF16) if epi in (0, 1, 4) else None
    res = (torch.randn(M, N, device="cuda", generator=g)).to(BF16) if epi == 1 else None
    x8, xs = ops.quantize_fp8_rows(x)
    w8, ws = ops.quantize_fp8_rows(w)
    got = ops.linear_fp8(x8, xs, w8, ws, bias=bias, res=res, epi=epi)
    rep = ops.linear_fp8(x8, xs, w8, ws, bias=bias, res=res, epi=epi)
    assert torch.equal(got, rep)
    xd = x8.view(torch.float8_e4m3fn).float() * xs[:, None]
    wd = w8.view(torch.float8_e4m3fn).float() * ws[:, None]
    y = xd @ wd.t()
    if bias is not None:
        y = y + bias.float()
    y = bfr(y)
    if epi == 1:
        y = bfr(y + res.float())
    elif epi == 2:
        y = bfr(bfr(F.silu(y[:, 0::2])) * y[:, 1::2])
    elif epi == 4:
        y = bfr(F.gelu(y))
    elif epi == 5:
        y = bfr(y[:, 0::2] * bfr(F.gelu(y[:, 1::2])))
    err = (got.float() - y).abs()
    tol = 1e-2 * float(y.abs().max()) + 2e-2 * y.abs()
    assert not bool((err > tol).any()), (int((err > tol).sum()), float(err.max()), float(y.abs().max()))
    # and the quantisation itself stays close to the bf16 product (per-row e4m3 on both operands: ~3 % relative L2)
    if epi == 0:
        full = bfr(x.float() @ w.float().t() + bias.float())
        assert rel_err(got, full) < 6e-2


def _fp8_gemm_check(M, N, K, epi, seed=7):
    import torch.nn.functional as F
    from emu_amd import ops
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = (torch.randn(M, K, device="cuda", generator=g)).to(BF16)
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.02).to(BF16)
    bias = (torch.randn(N, device="cuda", generator=g)).to(BF16) if epi in (0, 1, 4) else None
    res = (torch.randn(M, N, device="cuda", generator=g)).to(BF16) if epi == 1 else None
    x8, xs = ops.quantize_fp8_rows(x)
    w8, ws = ops.quantize_fp8_rows(w)
    got = ops.linear_fp8(x8, xs, w8, ws, bias=bias, res=res, epi=epi)
    rep = ops.linear_fp8(x8, xs, w8, ws, bias=bias, res=res, epi=epi)
    assert torch.equal(got, rep)
    y = (x8.view(torch.float8_e4m3fn).float() * xs[:, None]) @ (w8.view(torch.float8_e4m3fn).float() * ws[:, None]).t()
    if bias is not None:
        y = y + bias.float()
    y = bfr(y)
    if epi == 1:
        y = bfr(y + res.float())
    elif epi == 2:
        y = bfr(bfr(F.silu(y[:, 0::2])) * y[:, 1::2])
    elif epi == 4:
        y = bfr(F.gelu(y))
    elif epi == 5:
        y = bfr(y[:, 0::2] * bfr(F.gelu(y[:, 1::2])))
    err = (got.float() - y).abs()
    tol = 1e-2 * float(y.abs().max()) + 2e-2 * y.abs()
    assert not bool((err > tol).any()), (int((err > tol).sum()), float(err.max()), float(y.abs().max()))
    return got


@pytest.fixture
def force_cfg():
    from emu_amd._lib import lib
    L = lib()
    yield lambda c: L.emu_gemm_force_config(0 if c == "0" else ord(c))
    L.emu_gemm_force_config(0)


# the UNet's M = 2048 transformer-block shapes, the ViT's ragged 1025 rows, the prompt's 770, ragged N, a K of one k tile
FP8_CFG_SHAPES = [(2048, 1280, 1280, 1), (2048, 3840, 1280, 0), (2048, 2560, 1280, 5), (2048, 1280, 5120, 1), (1025, 1792, 1792, 1),
                  (1025, 1536, 1792, 4), (770, 1024, 6656, 2), (300, 520, 384, 0), (65, 136, 128, 1), (8192, 640, 640, 0)]


@pytest.mark.parametrize("cfg", ["K", "B", "C", "S", "P", "0"])
@pytest.mark.parametrize("M,N,K,epi", FP8_CFG_SHAPES)
def test_fp8_gemm_every_tile_configuration(force_cfg, cfg, M, N, K, epi):
    """Round 4: the lock-step tiles (128 x 64 with two k-groups, 128 x 128, 256 x 128, K-sliced 256 x 128) have an fp8 form
    (gemm.hip, F8), the 256 x 256 tile carries the remainder rows of M = 256 j + r (r <= 32) with fp8 operands too; every
    configuration is pinned against the fp32 product of the exactly de-quantised operands, and all configurations agree with
    each other to the accumulation order (same tolerance)."""
    force_cfg(cfg)
    _fp8_gemm_check(M, N, K, epi)


def test_fp8_gemm_identity_every_configuration(force_cfg):
    """A = I against an asymmetric small-integer W (exact in e4m3): out[m, n] == W[n, m] EXACTLY under every tile configuration
    -- catches a row / column / k-half swap in any of the fragment maps."""
    from emu_amd import ops
    K = 256
    x8 = torch.eye(K, device="cuda").to(torch.float8_e4m3fn).view(torch.uint8)
    wi = ((torch.arange(320 * K, device="cuda").reshape(320, K) * 7) % 31 - 15).float()
    w8 = wi.to(torch.float8_e4m3fn).view(torch.uint8)
    ones_m, ones_n = torch.ones(K, device="cuda"), torch.ones(320, device="cuda")
    for cfg in "KBCSPQ":
        force_cfg(cfg)
        got = ops.linear_fp8(x8, ones_m, w8, ones_n)
        assert torch.equal(got.float(), wi.t().contiguous()), cfg


def test_fp8_mfma_gemm_identity_asymmetric():
    """A = I (exactly representable in e4m3, unit scales) against an asymmetric small-integer W: out[m, n] == W[n, m] exactly
    -- catches any row / column / k-half swap in the 32x32x64 fragment maps."""
    from emu_amd import ops
    K = 256
    x8 = torch.eye(K, device="cuda").to(torch.float8_e4m3fn).view(torch.uint8)
    wi = ((torch.arange(320 * K, device="cuda").reshape(320, K) * 7) % 31 - 15).float()       # |w| <= 15: exact in e4m3
    w8 = wi.to(torch.float8_e4m3fn).view(torch.uint8)
    ones_m, ones_n = torch.ones(K, device="cuda"), torch.ones(320, device="cuda")
    got = ops.linear_fp8(x8, ones_m, w8, ones_n)
    assert torch.equal(got.float(), wi.t().contiguous())


def test_fp8_prefill_true_width_layer_tracks_bf16():
    """One LLaMA-33B-shaped decoder layer (6656 / 52 heads / 17920), S = 300 prompt rows: prefill with use_fp8(prefill=True)
    (W8A8 GEMMs on the block-scaled MFMA, activations quantised per row) stays within 0.2 relative L2 of the bf16 prefill on
    random N(0, 0.02) weights (measured 0.13: four chained GEMMs with two e4m3 roundings each, ~5 % per GEMM, nothing to
    average the error down on unstructured weights) -- a plumbing check, not a parity claim: the reference has no fp8 mode and
    the GEMM itself is pinned exactly in test_fp8_mfma_gemm_matches_dequantised_reference -- and is deterministic."""
    from emu_amd import synth
    from emu_amd.conf.emu_conf import LlamaCfg
    from emu_amd.llama import EmuHipContext, LlamaEngine
    l = LlamaCfg(num_hidden_layers=1)
    eng = LlamaEngine(l, 256, EmuHipContext(torch.device("cuda", 0)))
    eng.load_weights(synth.iter_synth(synth.llama_param_shapes(l, 256), device="cuda", dtype=BF16))
    S = 300
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1, S, l.hidden_size, generator=g).to(BF16).cuda()
    mask = torch.ones(1, S, dtype=torch.long)
    s_max = eng.kv_capacity(S + 8)
    ref, _, _ = eng.prefill(x.clone(), mask, s_max)
    ref = ref.clone()
    eng.use_fp8(True, prefill=True)
    try:
        a, _, _ = eng.prefill(x.clone(), mask, s_max)
        a = a.clone()
        b, _, _ = eng.prefill(x.clone(), mask, s_max)
        assert torch.equal(a, b)
    finally:
        eng.use_fp8(False)
    assert bool(torch.isfinite(a.float()).all())
    assert rel_err(a, ref) < 0.2, rel_err(a, ref)
    assert rel_err(a, ref) > 1e-4                    # the fp8 path really ran


# ------------------------------------------------------------------------------------------------ W8A8 ViT blocks
@pytest.mark.parametrize("postnorm,layers,image", [(True, 2, 448), (False, 1, 224)])
def test_fp8_vit_true_width_blocks_track_bf16(postnorm, layers, image):
    """EVA-CLIP-4B-shaped encoder blocks (1792 wide, 16 heads of 112, MLP 15360) on a 448-pixel image (1025 tokens: the 256x256
    tile's remainder row with fp8 operands; qkv / proj on the lock-step tiles' fp8 form): ``use_fp8`` quantises the four
    matrices of every block per row, quantises the activation rows ahead of every GEMM and must (i) be deterministic, (ii) give
    back the bf16 engine bit for bit when switched off, (iii) stay within 0.15 relative L2 of the bf16 blocks on random
    N(0, 0.02) weights (measured ~0.05 per block) -- a plumbing check: the reference has no fp8 mode, the GEMM itself is pinned
    exactly in test_fp8_gemm_every_tile_configuration -- and (iv) really differ from bf16."""
    from emu_amd import CLIPVisionCfg, synth
    from emu_amd.llama import EmuHipContext
    from emu_amd.vit import VitEngine
    v = CLIPVisionCfg(layers=layers, image_size=image, postnorm=postnorm)
    ctx = EmuHipContext(torch.device("cuda", 0))
    vit = VitEngine(v, ctx)
    vit.load_weights(synth.iter_synth(synth.vit_param_shapes(v), seed=11, device="cuda", dtype=BF16))
    assert vit.ready
    g = torch.Generator().manual_seed(4)
    img = torch.randn(1, 3, image, image, generator=g).to(BF16).cuda()
    ref = vit(img).clone()
    vit.use_fp8(True)
    try:
        a = vit(img).clone()
        b = vit(img).clone()
        assert torch.equal(a, b)
        tok = vit.run_blocks(ref, 0, layers)             # the parity hook takes the same switch
        assert bool(torch.isfinite(tok.float()).all())
    finally:
        vit.use_fp8(False)
    assert torch.equal(vit(img), ref)
    assert bool(torch.isfinite(a.float()).all())
    e = rel_err(a, ref)
    assert 1e-4 < e < 0.15, e
    # the quantised matrices are the quantiser's definition of the PACKED bf16 matrices
    q, sc = cpu_quant_rows(vit._keep["0.fc1w"].cpu())
    assert torch.equal(vit.fp8_dequantized("0.fc1w").cpu(), q.float() * sc[:, None])


def test_fp8_vit_refuses_widths_off_the_k_tile():
    from emu_amd import CLIPVisionCfg, synth
    from emu_amd._lib import EmuHipError
    from emu_amd.llama import EmuHipContext
    from emu_amd.vit import VitEngine
    v = CLIPVisionCfg(image_size=56, patch_size=14, width=192, layers=1, head_width=64, mlp_ratio=2.0)
    vit = VitEngine(v, EmuHipContext(torch.device("cuda", 0)))
    vit.load_weights(synth.iter_synth(synth.vit_param_shapes(v), seed=1, device="cuda", dtype=BF16))
    with pytest.raises(EmuHipError):
        vit.use_fp8(True)
