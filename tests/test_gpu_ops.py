"""GPU parity tests of the primitive C-ABI operators (through emu_amd.ops -> libemu_hip.so) against CPU fp32
references that keep the reference's bf16 rounding points.  Run on an MI355X with `-m gpu`."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16


def _ops():
    from emu_amd import ops
    return ops


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(BF16)


def bfr(t):
    return t.to(BF16).float()


def ref_linear(x, w, bias=None, res=None, norm_w=None, eps=0.0, epi=0):
    """CPU fp32 math on bf16-valued inputs with the reference's rounding points."""
    x, w = x.float(), w.float()
    if norm_w is not None:
        x = bfr(norm_w.float() * bfr(x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps)))
    y = x @ w.t()
    if bias is not None:
        y = y + bias.float()
    y = bfr(y)
    if epi == 1:
        y = bfr(y + res.float())
    elif epi == 2:
        y = bfr(bfr(F.silu(y[:, 0::2])) * y[:, 1::2])
    elif epi == 3:
        y = bfr(F.silu(y))
    elif epi == 4:
        y = bfr(F.gelu(y))
    elif epi == 5:
        y = bfr(y[:, 0::2] * bfr(F.gelu(y[:, 1::2])))
    return y


def close(got, want, rtol=2e-2, atol=None, what=""):
    got, want = got.float().cpu(), want.float().cpu()
    if atol is None:
        atol = 1e-2 * float(want.abs().max().clamp_min(1e-6))
    bad = (got - want).abs() > atol + rtol * want.abs()
    assert not bool(bad.any()), (f"{what}: {int(bad.sum())}/{bad.numel()} mismatches, max abs err "
                                 f"{float((got - want).abs().max())}, ref max {float(want.abs().max())}")


@pytest.mark.parametrize("M", [1, 2, 3, 5, 8])
@pytest.mark.parametrize("N,K", [(64, 256), (1002, 896), (520, 6656), (96, 17920)])
@pytest.mark.parametrize("epi", [0, 1, 2, 3])
def test_gemv(M, N, K, epi):
    ops = _ops()
    x, w = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.05)
    bias = rnd(N, seed=3) if epi in (0, 3) else None
    res = rnd(M, N, seed=4) if epi == 1 else None
    got = ops.linear(x.cuda(), w.cuda(), bias=None if bias is None else bias.cuda(),
                     res=None if res is None else res.cuda(), epi=epi)
    close(got, ref_linear(x, w, bias, res, epi=epi), what=f"gemv M{M} N{N} K{K} epi{epi}")


@pytest.mark.parametrize("M", [2, 5, 9, 13, 16])
@pytest.mark.parametrize("N,K", [(50, 64), (1002, 896), (264, 6656), (96, 17920)])
@pytest.mark.parametrize("epi", [0, 1, 2, 3, 4])
def test_skinny_mfma_stream(M, N, K, epi):
    """2..16 activation rows (beam search, CFG pairs) stream the weights through the 16x16x32 MFMA kernel: ragged N
    (partial 16-row groups, odd store alignment), every epilogue, bias."""
    ops = _ops()
    x, w = rnd(M, K, seed=31), rnd(N, K, seed=32, scale=0.05)
    bias = rnd(N, seed=33) if epi in (0, 2, 3) else None
    res = rnd(M, N, seed=34) if epi == 1 else None
    got = ops.linear(x.cuda(), w.cuda(), bias=None if bias is None else bias.cuda(),
                     res=None if res is None else res.cuda(), epi=epi)
    close(got, ref_linear(x, w, bias, res, epi=epi), what=f"skinny M{M} N{N} K{K} epi{epi}")


@pytest.mark.parametrize("M", [4, 5, 7, 8, 9, 12, 16])
@pytest.mark.parametrize("N,K", [(50, 256), (1002, 512), (264, 768), (37, 1024), (16, 256)])
@pytest.mark.parametrize("epi", [0, 1, 2, 3, 4])
def test_thin_stream_rows_4_to_16(M, N, K, epi):
    """4..16 activation rows with K % 256 == 0 (5 beams: the reference's default decoding mode) stream the weights through LDS-DMA
    stages into v_mfma_f32_16x16x32_bf16 (gemv_thin.hip): one to four k stages (shorter than the ring is deep), ragged N (partial
    16-row workgroups, fewer rows than one workgroup, odd store alignment), both activation-row paddings (8 / 16), every
    epilogue, bias; the same call with the stream switched off (emu_gemm_tune bit 2) must agree to accumulation-order noise."""
    ops = _ops()
    from emu_amd._lib import lib
    N += int(epi == 2 and N % 2)                                   # gate / up pairs
    x, w = rnd(M, K, seed=61), rnd(N, K, seed=62, scale=0.05)
    bias = rnd(N, seed=63) if epi in (0, 2, 3) else None
    res = rnd(M, N, seed=64) if epi == 1 else None
    args = dict(bias=None if bias is None else bias.cuda(), res=None if res is None else res.cuda(), epi=epi)
    got = ops.linear(x.cuda(), w.cuda(), **args)
    assert torch.equal(got, ops.linear(x.cuda(), w.cuda(), **args))
    close(got, ref_linear(x, w, bias, res, epi=epi), what=f"thin M{M} N{N} K{K} epi{epi}")
    lib().emu_gemm_tune(4)
    try:
        old = ops.linear(x.cuda(), w.cuda(), **args)
    finally:
        lib().emu_gemm_tune(0)
    assert float((got.float() - old.float()).abs().max()) <= 2 ** -6 * float(old.float().abs().max().clamp_min(1e-3))


def test_thin_stream_identity_asymmetric_and_strides():
    """x = the first 16 rows of I (K = 512: two k stages) against an asymmetric, exactly representable W: out[m, n] must equal
    W[n, m] exactly -- catches any row / column / k-slot swap between the DMA swizzle, the ds_read of the fragments and the
    16x16x32 operand maps.  Strided x rows, strided weight rows and strided out rows; nothing is written beyond N."""
    ops = _ops()
    K, N, M = 512, 83, 16
    x = torch.eye(K)[:M].to(BF16)
    w = ((torch.arange(N * K).reshape(N, K) * 7) % 251 - 125).float().to(BF16)
    xb = torch.zeros(M, 2 * K, dtype=BF16)
    xb[:, :K] = x
    wb = torch.zeros(N, K + 64, dtype=BF16)
    wb[:, :K] = w
    out = torch.zeros(M, N + 8, dtype=BF16, device="cuda")
    ops.linear(xb.cuda()[:, :K], wb.cuda()[:, :K], out=out[:, :N])
    assert torch.equal(out[:, :N].float().cpu(), w.float()[:, :M].t().contiguous())
    assert float(out[:, N:].abs().max()) == 0.0
    # k values far into the row: x = rows 300..304 of I
    x2 = torch.eye(K)[300:305].to(BF16)
    got = ops.linear(x2.cuda(), w.cuda())
    assert torch.equal(got.float().cpu(), w.float()[:, 300:305].t().contiguous())


def test_skinny_mfma_identity_asymmetric():
    """x = the first 16 rows of I with an asymmetric, exactly representable W: out[m, n] must equal W[n, m] exactly
    (catches any row/col or k-lane swap in the 16x16x32 fragment maps); strided x and out rows."""
    ops = _ops()
    K, N, M = 96, 80, 16
    x = torch.eye(K)[:M].to(BF16)
    w = ((torch.arange(N * K).reshape(N, K) * 7) % 251 - 125).float().to(BF16)
    xb = torch.zeros(M, 2 * K, dtype=BF16)
    xb[:, :K] = x
    out = torch.zeros(M, N + 8, dtype=BF16, device="cuda")
    ops.linear(xb.cuda()[:, :K], w.cuda(), out=out[:, :N])
    assert torch.equal(out[:, :N].float().cpu(), w.float()[:, :M].t().contiguous())
    assert float(out[:, N:].abs().max()) == 0.0


@pytest.mark.parametrize("N,K", [(6656, 896), (2050, 2240), (1030, 64), (4099, 2560), (1024, 520)])
@pytest.mark.parametrize("epi", [0, 1])
def test_gemv_short_rows_wave_kernel(N, K, epi):
    """Short rows (K <= 2560, e.g. the o_proj / down_proj of a TP = 8 shard): one wave per 4 rows, no LDS; ragged N, K not a
    multiple of 512 (partially filled last load), bias, residual."""
    ops = _ops()
    x, w = rnd(1, K, seed=51), rnd(N, K, seed=52, scale=0.05)
    bias = rnd(N, seed=53) if epi == 0 else None
    res = rnd(1, N, seed=54) if epi == 1 else None
    got = ops.linear(x.cuda(), w.cuda(), bias=None if bias is None else bias.cuda(),
                     res=None if res is None else res.cuda(), epi=epi)
    close(got, ref_linear(x, w, bias, res, epi=epi), what=f"wave gemv N{N} K{K} epi{epi}")


@pytest.mark.parametrize("M", [1, 4])
@pytest.mark.parametrize("epi", [0, 2])
def test_gemv_fused_rmsnorm(M, epi):
    ops = _ops()
    K, N = 6656, 264
    x, w, g = rnd(M, K, seed=1, scale=3.0), rnd(N, K, seed=2, scale=0.05), (1 + 0.1 * rnd(K, seed=5).float()).to(BF16)
    got = ops.linear(x.cuda(), w.cuda(), norm_w=g.cuda(), eps=1e-6, epi=epi)
    close(got, ref_linear(x, w, norm_w=g, eps=1e-6, epi=epi), what="gemv+rmsnorm")


@pytest.mark.parametrize("N,K", [(4096, 2048), (2050, 6656), (2304, 17920), (5000, 5120), (2048, 13824)])
@pytest.mark.parametrize("epi,norm", [(0, False), (0, True), (1, False), (2, True), (2, False)])
def test_gemv_large_decode_matrices(N, K, epi, norm):
    """Decode-size matrices (M = 1, >= 16 MiB of weights, every dispatch branch of launch_gemv): ragged N (N % 16 != 0),
    short / long rows, every epilogue, fused RMSNorm; repeated launches are deterministic."""
    ops = _ops()
    x, w = rnd(1, K, seed=1, scale=2.0 if norm else 1.0), rnd(N, K, seed=2, scale=0.05)
    g = (1 + 0.1 * rnd(K, seed=5).float()).to(BF16) if norm else None
    bias = rnd(N, seed=3) if epi == 0 else None
    res = rnd(1, N, seed=4) if epi == 1 else None
    wd = w.cuda()
    outs = [ops.linear(x.cuda(), wd, bias=None if bias is None else bias.cuda(), res=None if res is None else res.cuda(),
                       norm_w=None if g is None else g.cuda(), eps=1e-6, epi=epi) for _ in range(3)]
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    close(outs[0], ref_linear(x, w, bias, res, norm_w=g, eps=1e-6, epi=epi), what=f"gemv N{N} K{K} epi{epi} norm{norm}")


def test_gemv_one_row_matches_two_rows():
    """Same row through the M = 1 and the M = 2 kernels agrees to accumulation-order noise."""
    ops = _ops()
    N, K = 6656, 6656
    x, w, r = rnd(1, K, seed=21).cuda(), rnd(N, K, seed=22, scale=0.03).cuda(), rnd(1, N, seed=23).cuda()
    a = ops.linear(x, w, res=r, epi=1)
    b = ops.linear(torch.cat([x, x]), w, res=torch.cat([r, r]), epi=1)
    assert float((a.float() - b[:1].float()).abs().max()) <= 2 ** -6 * float(b.float().abs().max())


def test_gemv_strided_rows():
    ops = _ops()
    big = rnd(4, 3 * 512, seed=7).cuda()
    x = big[:, 512:1024]                       # ldx = 1536, K = 512
    w = rnd(40, 512, seed=8, scale=0.05)
    close(ops.linear(x, w.cuda()), ref_linear(x.cpu(), w), what="gemv strided")


@pytest.mark.parametrize("M,N,K", [(9, 128, 64), (128, 128, 128), (300, 320, 640), (770, 384, 1792), (257, 1000, 200),
                                   (130, 6656, 6656)])
@pytest.mark.parametrize("epi", [0, 1, 2, 3, 4, 5])
def test_gemm(M, N, K, epi):
    ops = _ops()
    if N * K > 4e6 and epi not in (0, 2):
        pytest.skip("large shape: two epilogues are enough")
    x, w = rnd(M, K, seed=11), rnd(N, K, seed=12, scale=0.05)
    bias = rnd(N, seed=13) if epi in (0, 1, 4) else None
    res = rnd(M, N, seed=14) if epi == 1 else None
    got = ops.linear(x.cuda(), w.cuda(), bias=None if bias is None else bias.cuda(),
                     res=None if res is None else res.cuda(), epi=epi)
    close(got, ref_linear(x, w, bias, res, epi=epi), what=f"gemm M{M} N{N} K{K} epi{epi}")


def test_gemm_identity_asymmetric():
    """A = I with an asymmetric W catches any row/col swap in the MFMA fragment maps exactly."""
    ops = _ops()
    K = 256
    x = torch.eye(K).to(BF16)
    w = (torch.arange(192 * K).reshape(192, K) % 251 - 125).float().to(BF16)      # exactly representable
    got = ops.linear(x.cuda(), w.cuda())
    assert torch.equal(got.float().cpu(), w.float().t().contiguous())


def test_gemm_matches_gemv_rows():
    """Same weights through both kernels: the M<=8 GEMV and the MFMA GEMM agree to accumulation-order noise."""
    ops = _ops()
    x, w = rnd(16, 6656, seed=21).cuda(), rnd(512, 6656, seed=22, scale=0.05).cuda()
    a = ops.linear(x, w)                       # GEMM (M = 16)
    b = torch.cat([ops.linear(x[:8], w), ops.linear(x[8:], w)])
    close(a, b.float().cpu(), rtol=1e-2, what="gemm vs gemv")


@pytest.mark.parametrize("rows,cols", [(1, 6656), (5, 256), (300, 1792)])
def test_rmsnorm_layernorm(rows, cols):
    ops = _ops()
    x = rnd(rows, cols, seed=31, scale=2.0)
    w = (1 + 0.1 * rnd(cols, seed=32).float()).to(BF16)
    b = rnd(cols, seed=33, scale=0.1)
    res = rnd(rows, cols, seed=34)
    xf = x.float()
    want = bfr(w.float() * bfr(xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6)))
    close(ops.rmsnorm(x.cuda(), w.cuda(), 1e-6), want, what="rmsnorm")
    ln = bfr(F.layer_norm(xf, (cols,), w.float(), b.float(), 1e-6))
    close(ops.layernorm(x.cuda(), w.cuda(), b.cuda(), 1e-6), ln, what="layernorm")
    close(ops.layernorm(x.cuda(), w.cuda(), b.cuda(), 1e-6, res=res.cuda()), bfr(res.float() + ln), what="layernorm+res")
    # in-place residual form used by the ViT engine (y aliases res)
    r = res.cuda().clone()
    ops.layernorm(x.cuda(), w.cuda(), b.cuda(), 1e-6, res=r, out=r)
    close(r, bfr(res.float() + ln), what="layernorm in place")


def test_embed_scatter_argmax_avgpool():
    ops = _ops()
    table = rnd(1000, 256, seed=41)
    ids = torch.tensor([0, 999, 5, 5, 123], dtype=torch.int32)
    e = ops.embed_gather(ids.cuda(), table.cuda())
    assert torch.equal(e.cpu(), table[ids.long()])
    src = rnd(2, 256, seed=42)
    ops.scatter_rows(src.cuda(), torch.tensor([3, 1], dtype=torch.int32).cuda(), e)
    want = table[ids.long()].clone(); want[3] = src[0]; want[1] = src[1]
    assert torch.equal(e.cpu(), want)
    logits = rnd(3, 32274, seed=43)
    logits[0, 777] = 50.0; logits[0, 30000] = 50.0          # tie -> first index
    logits[1, 2] = 60.0                                      # suppressed
    got = ops.argmax(logits.cuda(), suppress_id=2).cpu()
    lf = logits.float().clone(); lf[:, 2] = -float("inf")
    assert got.tolist() == lf.argmax(-1).tolist() and got[0] == 777
    x = rnd(2, 17, 224, seed=44)
    for s in (1, 2, 4):
        ref = F.avg_pool2d(x[:, 1:].float().permute(0, 2, 1).reshape(2, 224, 4, 4), s, s).reshape(2, 224, -1).permute(0, 2, 1)
        close(ops.avgpool_tokens(x.cuda(), 4, s), bfr(ref), what=f"avgpool s{s}")


@pytest.mark.parametrize("vocab", [7, 64, 1000, 32274])
def test_argmax_alignment_and_ties(vocab):
    """Rows that start at every 2-byte phase of a 16-byte line (ld = vocab, odd sizes), first-index tie-break, suppressed
    id; compared with torch.argmax on the same bf16 values (torch also returns the first maximum on CPU)."""
    ops = _ops()
    rows = 9
    g = torch.Generator().manual_seed(vocab)
    x = torch.randn(rows, vocab, generator=g).to(BF16)
    x[1, vocab // 2] = 50.0; x[1, vocab - 1] = 50.0            # tie: the first index wins
    x[2, 0] = 60.0
    x[3, vocab - 1] = 60.0
    want = [int(torch.argmax(x[r].float())) for r in range(rows)]
    got = ops.argmax(x.cuda()).cpu().tolist()
    assert got == want
    sup = want[4]
    y = x[4:5].float().clone(); y[0, sup] = -float("inf")
    assert ops.argmax(x[4:5].cuda().contiguous(), suppress_id=sup).cpu().tolist() == [int(torch.argmax(y[0]))]


@pytest.mark.parametrize("D", [64, 128])
def test_rope_kv_append(D):
    from oracle import emu2_ref as R
    ops = _ops()
    B, T, H, S_max = 2, 5, 3, 16
    from emu_amd.llama import rope_tables
    cos, sin = rope_tables(D, 64, 10000.0, "cuda")
    qkv = rnd(B * T, 3 * H * D, seed=51)
    pos = torch.tensor([[3, 4, 5, 6, 7], [0, 1, 2, 3, 4]], dtype=torch.int32)
    slot = torch.tensor([[2, 3, 4, 5, 6], [2, 3, 4, 5, 6]], dtype=torch.int32)
    kc = torch.zeros(B, H, S_max, D, dtype=BF16, device="cuda"); vc = torch.zeros_like(kc)
    dq = qkv.cuda().clone()
    ops.rope_kv_append(dq, cos, sin, pos.reshape(-1).cuda(), slot.reshape(-1).cuda(), kc, vc, B, T, H, D)
    q = qkv.view(B, T, 3, H, D)
    c, s = R.rope_cos_sin(pos.long(), D, 10000.0, BF16)
    qr, kr = R.apply_rope(q[:, :, 0].transpose(1, 2), q[:, :, 1].transpose(1, 2), c, s)     # bf16 ops, [B,H,T,D]
    got = dq.view(B, T, 3, H, D).cpu()
    assert torch.equal(got[:, :, 0].transpose(1, 2), qr)
    assert torch.equal(got[:, :, 1].transpose(1, 2), kr)
    assert torch.equal(got[:, :, 2], q[:, :, 2])
    assert torch.equal(kc[:, :, 2:7].cpu(), kr) and torch.equal(vc[:, :, 2:7].cpu(), q[:, :, 2].transpose(1, 2))
    assert float(kc[:, :, :2].abs().max()) == 0 and float(kc[:, :, 7:].abs().max()) == 0


def ref_attention(q, k, v, causal, scale, kstart=None):
    """q [B,Sq,H,D], k/v [B,Sk,H,D] fp32 softmax reference."""
    q, k, v = (t.float().transpose(1, 2) for t in (q, k, v))
    Sq, Sk = q.shape[2], k.shape[2]
    s = q @ k.transpose(-1, -2) * scale
    allowed = torch.ones(q.shape[0], 1, Sq, Sk, dtype=torch.bool)
    if causal:
        allowed &= (torch.arange(Sk)[None, :] <= torch.arange(Sq)[:, None] + (Sk - Sq))[None, None]
    if kstart is not None:
        allowed &= (torch.arange(Sk)[None, :] >= kstart[:, None])[:, None, None, :]
    s = s.masked_fill(~allowed, -float("inf"))
    p = torch.softmax(s, -1)
    p = torch.nan_to_num(p, nan=0.0)
    return (p @ v).transpose(1, 2)


@pytest.mark.parametrize("D", [64, 128])
@pytest.mark.parametrize("Sq,Sk,causal", [(17, 17, False), (200, 200, True), (130, 333, True), (64, 64, True),
                                           (1025, 1025, False), (5, 77, False)])
def test_flash_attn(D, Sq, Sk, causal):
    ops = _ops()
    B, H = 2, 3
    q, k, v = rnd(B, Sq, H, D, seed=61), rnd(B, Sk, H, D, seed=62), rnd(B, Sk, H, D, seed=63)
    scale = D ** -0.5
    got = ops.flash_attn(q.cuda(), k.cuda(), v.cuda(), causal, scale)
    close(got, ref_attention(q, k, v, causal, scale), atol=2e-2, what=f"flash D{D} {Sq}x{Sk} causal{causal}")


@pytest.mark.parametrize("Sq,Sk,causal,H", [(1024, 1024, False, 20), (300, 300, True, 3), (257, 700, False, 2), (1024, 64, False, 4)])
def test_flash_attn_variants_agree_bitwise(Sq, Sk, causal, H):
    """Round 4 forms of the D = 64 attention kernel -- 256-query workgroups (8 waves sharing every K / V tile), the XCD-aware
    (head, query block) order, O staged through LDS -- change neither the arithmetic nor its order: every combination gives the
    bits of the round-3 form (emu_gemm_tune: bit 6 straight order, bit 7 direct O stores, bits 12-13 = 1 / 2: always 4 / 8 waves),
    and all of them the reference."""
    from emu_amd._lib import lib
    ops = _ops()
    B, D = 2, 64
    q, k, v = rnd(B, Sq, H, D, seed=161), rnd(B, Sk, H, D, seed=162), rnd(B, Sk, H, D, seed=163)
    scale = D ** -0.5
    outs = []
    try:
        for tune in (64 | 128 | 4096, 4096, 8192, 8192 | 64 | 128, 0):
            lib().emu_gemm_tune(tune)
            outs.append(ops.flash_attn(q.cuda(), k.cuda(), v.cuda(), causal, scale).cpu())
    finally:
        lib().emu_gemm_tune(0)
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    close(outs[0], ref_attention(q, k, v, causal, scale), atol=2e-2, what=f"flash variants {Sq}x{Sk}")


def test_flash_attn_left_padding_and_strided_qkv():
    ops = _ops()
    B, S, H, D = 2, 150, 2, 128
    qkv = rnd(B, S, 3, H, D, seed=64).cuda()                  # packed projection output, like the engines use
    kstart = torch.tensor([0, 37], dtype=torch.int32)
    got = ops.flash_attn(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], True, D ** -0.5, kstart=kstart.cuda())
    want = ref_attention(qkv[:, :, 0].cpu(), qkv[:, :, 1].cpu(), qkv[:, :, 2].cpu(), True, D ** -0.5, kstart.long())
    ok_rows = torch.ones(B, S, dtype=torch.bool); ok_rows[1, :37] = False     # padded query rows are don't-care
    close(got.cpu()[ok_rows], want[ok_rows], atol=2e-2, what="flash left pad")
    assert bool(torch.isfinite(got.float()).all())


def test_flash_attn_spiked_scores():
    """Force the online-softmax rescale path: one key dominates late in the sequence."""
    ops = _ops()
    B, S, H, D = 1, 300, 1, 128
    q, k, v = rnd(B, S, H, D, seed=65), rnd(B, S, H, D, seed=66), rnd(B, S, H, D, seed=67)
    k[0, 250, 0] = q[0, 280, 0] * 4
    got = ops.flash_attn(q.cuda(), k.cuda(), v.cuda(), True, D ** -0.5)
    close(got, ref_attention(q, k, v, True, D ** -0.5), atol=2e-2, what="flash spike")


@pytest.mark.parametrize("D", [64, 128])
@pytest.mark.parametrize("ctx", [1, 255, 256, 257, 700])
def test_decode_attn(D, ctx):
    ops = _ops()
    B, H, S_max = 2, 3, 1024
    q = rnd(B, H, D, seed=71)
    kc, vc = rnd(B, H, S_max, D, seed=72), rnd(B, H, S_max, D, seed=73)
    kstart = torch.tensor([0, min(5, ctx - 1)], dtype=torch.int32)
    want = ref_attention(q[:, None], kc[:, :, :ctx].transpose(1, 2), vc[:, :, :ctx].transpose(1, 2), False, D ** -0.5,
                         kstart.long())[:, 0]
    got = ops.decode_attn(q.cuda(), kc.cuda(), vc.cuda(), ctx, D ** -0.5, kstart=kstart.cuda())
    close(got, want, atol=2e-2, what=f"decode attn ctx{ctx}")
    # graph-replay form: live context read from device memory, launch sized for S_max
    ctx_dev = torch.tensor([ctx], dtype=torch.int32, device="cuda")
    got2 = ops.decode_attn(q.cuda(), kc.cuda(), vc.cuda(), S_max, D ** -0.5, kstart=kstart.cuda(), ctx_dev=ctx_dev,
                           ctx_max=S_max)
    assert torch.equal(got.cpu(), got2.cpu())
