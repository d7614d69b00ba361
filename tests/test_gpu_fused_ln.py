"""The fused epilogues of the UNet transformer GEMMs (emu_linear_fused_bf16 / GemmArgs::row_stats_out, ln_*, vt_*), which
replace the LayerNorm and V-transpose launches of diffusers' BasicTransformerBlock (norm1/2/3 -> Linear, attn1 V):

  * producer side: per-row (sum, sum of squares) of the bf16 outputs per 128-column slot, from every tile configuration's
    epilogue and from both K-slice reduce kernels -- compared with torch sums of the kernel's own output, and the output
    itself must be BIT-identical to the plain launch of the same configuration;
  * consumer side: LayerNorm folded into the GEMM (W * gamma, mean / rstd correction in the epilogue) -- against
    F.linear(F.layer_norm(x)) in fp32 and against the unfused HIP sequence (layernorm kernel + GEMM);
  * V^T store: the V columns of a fused qkv projection land key-contiguous, bit-identical to transpose(plain output);
  * the chain producer -> consumer as the engine runs it.
Every tile configuration is pinned in turn (emu_gemm_force_config) at the UNet's true shapes.  Run with `-m gpu`."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16
CFGS = ["B", "C", "S", "K", "P", "Q", "H", "0", "W"]     # "W": 'P' with the four-wave tile taken everywhere (emu_gemm_tune bit 22)


@pytest.fixture()
def force():
    from emu_amd._lib import lib
    L = lib()
    sk = torch.zeros(256 * 288 * 256, dtype=torch.float32, device="cuda")
    L.emu_set_splitk_scratch(sk.data_ptr(), sk.numel() * 4)
    def set_cfg(c):
        L.emu_gemm_tune((1 << 22) if c == "W" else 0)
        L.emu_gemm_force_config(0 if c == "0" else ord("P" if c == "W" else c))
    yield set_cfg
    L.emu_gemm_force_config(0)
    L.emu_gemm_tune(0)
    L.emu_set_splitk_scratch(0, 0)


def rnd(*shape, seed, scale=1.0, shift=0.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return (torch.randn(*shape, device="cuda", generator=g) * scale + shift).to(BF16)


def rel(got, want):
    got, want = got.float(), want.float()
    return float((got - want).norm() / want.norm().clamp_min(1e-12))


def slot_stats(y):
    """[N/128, M, 2] fp32: (sum, sum of squares) of every 128-column slot of every row of the bf16 tensor y."""
    M, N = y.shape
    v = y.float().view(M, N // 128, 128)
    return torch.stack((v.sum(-1), (v * v).sum(-1)), dim=-1).permute(1, 0, 2).contiguous()


# (M, N, K): attn out / to_q at 32^2, ff-out (K-sliced), proj at 64^2, ff-out at 64^2, a toy latent, and a shape whose forced
# 'H' launch really splits (320 tiles of 256 x 256: four tile columns on the ping-pong kernel + 256 remainder columns as a second
# GEMM whose statistics slots start at slot 8 -- the remainder once wrote them at twice that index)
PRODUCER_SHAPES = [(2048, 1280, 1280), (2048, 1280, 5120), (8192, 640, 640), (8192, 640, 2560), (128, 1280, 1280), (512, 384, 320),
                   (16384, 1280, 1280)]


@pytest.mark.parametrize("M,N,K", PRODUCER_SHAPES)
@pytest.mark.parametrize("epi", [0, 1])
def test_row_stats_from_every_epilogue(force, M, N, K, epi):
    from emu_amd import ops
    x, w, bias = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3, scale=0.1)
    res = rnd(M, N, seed=4, shift=0.3) if epi == 1 else None
    for c in CFGS:
        force(c)
        plain = ops.linear(x, w, bias=bias, res=res, epi=epi)
        st = torch.full((N // 128, M, 2), float("nan"), device="cuda", dtype=torch.float32)
        out = ops.linear_fused(x, w, bias=bias, res=res, epi=epi, stats_out=st)
        torch.cuda.synchronize()
        assert torch.equal(out, plain), f"cfg {c}: output changed by the statistics epilogue"
        want = slot_stats(out)
        assert bool(torch.isfinite(st).all()), f"cfg {c}: {int((~torch.isfinite(st)).sum())} slots never written"
        err = (st - want).abs()
        tol = 1e-4 * want.abs() + 2e-3
        assert bool((err <= tol).all()), f"cfg {c} {M}x{N}x{K} epi {epi}: max err {float(err.max())}"


# (M, N, K, epi): qkv / to_q / GEGLU at 32^2 and 64^2, toy latents
CONSUMER_SHAPES = [(2048, 3840, 1280, 0), (2048, 1280, 1280, 0), (2048, 10240, 1280, 5), (8192, 1920, 640, 0), (8192, 5120, 640, 5),
                   (128, 3840, 1280, 0), (512, 2560, 384, 5)]


def fold(w, gamma, beta, bias):
    wln = (w.float() * gamma.float()[None, :]).to(BF16)
    d = (w.float() * beta.float()[None, :]).sum(1) + (bias.float() if bias is not None else 0.0)
    return wln, wln.float().sum(1).contiguous(), d.contiguous()


@pytest.mark.parametrize("M,N,K,epi", CONSUMER_SHAPES)
def test_layernorm_folded_into_the_gemm(force, M, N, K, epi):
    from emu_amd import ops
    eps = 1e-5
    x = rnd(M, K, seed=5, scale=1.5, shift=0.7)                      # a mean that matters: the correction term is exercised
    gamma, beta = rnd(K, seed=6, scale=0.2, shift=1.0), rnd(K, seed=7, scale=0.2)
    w = rnd(N, K, seed=8, scale=K ** -0.5)
    bias = rnd(N, seed=9, scale=0.1) if epi == 5 else None
    wln, c, d = fold(w, gamma, beta, bias)
    st = slot_stats(x)
    # fp32 yardstick: the reference op sequence
    y = F.linear(F.layer_norm(x.float(), (K,), gamma.float(), beta.float(), eps), w.float(), None if bias is None else bias.float())
    if epi == 5:
        y = y[:, 0::2] * F.gelu(y[:, 1::2])
    # the unfused HIP sequence: layernorm kernel -> GEMM
    ln = ops.layernorm(x, gamma, beta, eps)
    for cfg in CFGS:
        force(cfg)
        unfused = ops.linear(ln, w, bias=bias, epi=epi)
        got = ops.linear_fused(x, wln, epi=epi, ln=(c, d, st, eps))
        torch.cuda.synchronize()
        e_f, e_u = rel(got, y), rel(unfused, y)
        assert e_f < max(6e-3, 1.5 * e_u), f"cfg {cfg} {M}x{N}x{K} epi {epi}: fused {e_f:.4f} vs unfused {e_u:.4f}"
        bad = (got.float() - y).abs() > 2e-2 * float(y.abs().max()) + 3e-2 * y.abs()
        assert not bool(bad.any()), f"cfg {cfg}: {int(bad.sum())}/{bad.numel()} elements off"


@pytest.mark.parametrize("B,S,C", [(2, 1024, 1280), (2, 4096, 640), (2, 64, 1280), (2, 256, 320)])
def test_v_transposed_store_of_the_qkv_projection(force, B, S, C):
    from emu_amd import ops
    M = B * S
    x, w = rnd(M, C, seed=10), rnd(3 * C, C, seed=11, scale=C ** -0.5)
    for cfg in CFGS:
        force(cfg)
        plain = ops.linear(x, w)
        out = torch.full((M, 3 * C), float("nan"), device="cuda", dtype=BF16)
        vt = torch.full((B, C, S), float("nan"), device="cuda", dtype=BF16)
        ops.linear_fused(x, w, out=out, vt=(vt, 2 * C, S))
        torch.cuda.synchronize()
        assert torch.equal(out[:, : 2 * C], plain[:, : 2 * C]), f"cfg {cfg}: q / k columns changed"
        want = plain[:, 2 * C:].reshape(B, S, C).permute(0, 2, 1).contiguous()       # [B, (h, d), s]
        assert torch.equal(vt, want), f"cfg {cfg}: V^T differs ({int((vt != want).sum())} elements)"
        assert bool(torch.isnan(out[:, 2 * C:].float()).all()), f"cfg {cfg}: V columns were also written row-major"


@pytest.mark.parametrize("M,C", [(2048, 1280), (8192, 640)])
def test_producer_consumer_chain_equals_unfused_sequence(force, M, C):
    """attn out-projection (+ residual, statistics) -> [LayerNorm] -> to_q, then GEGLU: what run_transformer launches."""
    from emu_amd import ops
    eps = 1e-5
    att, res = rnd(M, C, seed=20), rnd(M, C, seed=21, shift=0.2)
    wo, bo = rnd(C, C, seed=22, scale=C ** -0.5), rnd(C, seed=23, scale=0.1)
    gamma, beta = rnd(C, seed=24, scale=0.2, shift=1.0), rnd(C, seed=25, scale=0.2)
    wg, bg = rnd(8 * C, C, seed=26, scale=C ** -0.5), rnd(8 * C, seed=27, scale=0.1)
    wln, c, d = fold(wg, gamma, beta, bg)
    force("0")
    st = torch.zeros(C // 128, M, 2, device="cuda", dtype=torch.float32)
    h = ops.linear_fused(att, wo, bias=bo, res=res, epi=1, stats_out=st)
    fused = ops.linear_fused(h, wln, epi=5, ln=(c, d, st, eps))
    h2 = ops.linear(att, wo, bias=bo, res=res, epi=1)
    unfused = ops.linear(ops.layernorm(h2, gamma, beta, eps), wg, bias=bg, epi=5)
    torch.cuda.synchronize()
    assert torch.equal(h, h2)
    y = F.linear(F.layer_norm(h.float(), (C,), gamma.float(), beta.float(), eps), wg.float(), bg.float())
    y = y[:, 0::2] * F.gelu(y[:, 1::2])
    assert rel(fused, y) < max(6e-3, 1.5 * rel(unfused, y)), (rel(fused, y), rel(unfused, y))


@pytest.mark.parametrize("B,S,C,n", [(2, 1024, 1280, 64), (2, 4096, 640, 64), (2, 64, 256, 64), (2, 256, 128, 40), (1, 128, 64, 7)])
@pytest.mark.parametrize("with_ln", [False, True])
def test_cross_attention_in_the_to_q_epilogue(force, B, S, C, n, with_ln):
    """attn2 of a BasicTransformerBlock over the fixed prompt tokens: to_q GEMM + attention over n <= 64 cached keys + softmax + PV
    in ONE launch (emu_linear_fx::cross_*), against the two-launch sequence (GEMM, flash attention kernel) and against fp32
    torch; optionally with the LayerNorm ahead of to_q folded in as well.  n < 64 exercises the key mask / clamped K rows."""
    from emu_amd import ops
    H, D, M, eps, scale = C // 64, 64, B * S, 1e-5, 0.125
    x = rnd(M, C, seed=30, scale=1.2, shift=0.4)
    wq = rnd(C, C, seed=31, scale=C ** -0.5)
    kv = rnd(B * n, 2 * C, seed=32)                                    # K | V rows of the prompt, as emu_unet_set_context packs them
    vt = ops.transpose_v(kv[:, C:], B, H, n, D, n * 2 * C, D, 2 * C)   # [B, H, 64, n_pad], zero beyond n
    gamma, beta = rnd(C, seed=33, scale=0.2, shift=1.0), rnd(C, seed=34, scale=0.2)
    xin = ops.layernorm(x, gamma, beta, eps) if with_ln else x
    force("0")
    q = ops.linear(xin, wq)
    o = ops.flash_attn(q.view(B, S, H, D), kv[:, :C].view(B, n, H, D), kv[:, C:].view(B, n, H, D), False, scale).reshape(M, C)
    if with_ln:
        wln, c, d = fold(wq, gamma, beta, None)
        got = ops.linear_fused(x, wln, ln=(c, d, slot_stats(x), eps), cross=(kv, vt, n, S, scale)) if C % 128 == 0 else None
        if got is None:
            pytest.skip("statistics slots are 128 columns wide")
    else:
        got = ops.linear_fused(xin, wq, cross=(kv, vt, n, S, scale))
    torch.cuda.synchronize()
    qf = F.linear((F.layer_norm(x.float(), (C,), gamma.float(), beta.float(), eps) if with_ln else x.float()), wq.float())
    qf = qf.to(BF16).float().view(B, S, H, D).transpose(1, 2)
    kf, vf = kv[:, :C].float().view(B, n, H, D).transpose(1, 2), kv[:, C:].float().view(B, n, H, D).transpose(1, 2)
    y = (torch.softmax(qf @ kf.transpose(-1, -2) * scale, dim=-1) @ vf).transpose(1, 2).reshape(M, C)
    assert bool(torch.isfinite(got.float()).all())
    e_f, e_u = rel(got, y), rel(o, y)
    assert e_f < max(6e-3, 1.5 * e_u), (e_f, e_u)
    if not with_ln:
        assert rel(got, o) < 4e-3, rel(got, o)                          # same rounding points, another summation order
