"""Every GEMM / conv tile configuration at the shapes that carry the benchmark (LLaMA prefill S = 770 / 1544, ViT N = 1025,
the UNet's 32^2 / 64^2 / 128^2 levels), pinned one by one with emu_gemm_force_config and compared with a torch fp32
reference on the GPU (the reference op itself: F.linear / F.conv2d, then the reference's bf16 rounding points).
The shape heuristic only ever picks a subset of (configuration, shape) pairs; this walks the full product so a
tile-specific addressing bug cannot hide behind the dispatch.  Run on an MI355X with `-m gpu`."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16
CFGS = ["B", "C", "S", "K", "P", "Q", "H", "0", "W"]     # "W": 'P' with the four-wave tile taken everywhere (emu_gemm_tune bit 22)


def bfr(t):
    return t.to(BF16).float()


def ref_linear(x, w, bias, res, epi):
    y = x.float() @ w.float().t()
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


@pytest.fixture()
def force():
    from emu_amd._lib import lib
    L = lib()
    sk = torch.zeros(256 * 288 * 256, dtype=torch.float32, device="cuda")     # split-K scratch, as the engines carry
    L.emu_set_splitk_scratch(sk.data_ptr(), sk.numel() * 4)
    def set_cfg(c):
        L.emu_gemm_tune((1 << 22) if c == "W" else 0)
        L.emu_gemm_force_config(0 if c == "0" else ord("P" if c == "W" else c))
    yield set_cfg
    L.emu_gemm_force_config(0)
    L.emu_gemm_tune(0)
    L.emu_set_splitk_scratch(0, 0)


def check(got, want, what):
    got, want = got.float(), want.float()
    tol = 1e-2 * float(want.abs().max()) + 2e-2 * want.abs()
    bad = (got - want).abs() > tol
    assert not bool(bad.any()), f"{what}: {int(bad.sum())}/{bad.numel()} mismatches, max err {float((got - want).abs().max())}"


def rnd(*shape, seed, scale=1.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return (torch.randn(*shape, device="cuda", generator=g) * scale).to(BF16)


# (M, N, K, epilogues): the bench's true shapes (N cut where it only multiplies identical tiles)
SHAPES = [
    (770, 19968 // 4, 6656, (0,)), (770, 6656, 6656, (1,)), (770, 35840 // 4, 6656, (2,)), (770, 6656, 17920, (1,)),
    (1544, 6656, 6656, (0, 1, 2, 3, 4, 5)), (1025, 6144, 1792, (0,)), (1025, 1792, 2048, (1,)), (1025, 15360, 1792, (4,)),
    (1025, 1792, 15360, (1,)), (2048, 1280, 1280, (0, 1)), (2048, 10240, 1280, (5,)), (2048, 1280, 5120, (1,)),
    (8192, 1920, 640, (0,)), (8192, 5120, 640, (5,)), (8192, 640, 2560, (1,)), (800, 512, 1536, (0, 1, 2, 3, 4, 5)),
    (2048, 10240, 640, (0, 1, 2)),                      # 320 tiles of 256x256: one whole round + 2048 columns ('H')
]


@pytest.mark.parametrize("cfg", CFGS)
@pytest.mark.parametrize("M,N,K,epis", SHAPES)
def test_gemm_true_shapes_every_config(force, cfg, M, N, K, epis):
    from emu_amd import ops
    x, w = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.02)
    for epi in epis:
        bias = rnd(N, seed=3) if epi in (0, 1, 4) else None
        res = rnd(M, N, seed=4) if epi == 1 else None
        want = ref_linear(x, w, bias, res, epi)
        force(cfg)
        got = ops.linear(x, w, bias=bias, res=res, epi=epi)
        rep = ops.linear(x, w, bias=bias, res=res, epi=epi)
        assert torch.equal(got, rep), f"cfg {cfg} M{M} N{N} K{K} epi{epi}: repeat launch differs"
        check(got, want, f"cfg {cfg} M{M} N{N} K{K} epi{epi}")


def test_unsplit_configs_are_bit_identical(force):
    """128x128, 256x128 and the 256x256 ping-pong tile accumulate every output in the same k order: equal bits."""
    from emu_amd import ops
    x, w, bias = rnd(770, 6656, seed=5), rnd(1536, 6656, seed=6, scale=0.02), rnd(1536, seed=7)
    outs = []
    for cfg in "BCQ":
        force(cfg)
        outs.append(ops.linear(x, w, bias=bias))
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


@pytest.mark.parametrize("cfg", CFGS)
@pytest.mark.parametrize("B,H,Cin,Cout,mode", [(2, 128, 320, 320, 1), (2, 32, 1280, 1280, 1), (2, 32, 2560, 1280, 1),
                                              (2, 64, 640, 640, 2), (2, 32, 1280, 1280, 3), (1, 24, 64, 96, 1),
                                              (3, 17, 128, 200, 2), (1, 1100, 64, 64, 1)])
def test_conv_true_shapes_every_config(force, cfg, B, H, Cin, Cout, mode):
    from emu_amd import ops
    x, w = rnd(B, H, H, Cin, seed=11), rnd(Cout, 3, 3, Cin, seed=12, scale=0.02)
    bias, b2 = rnd(Cout, seed=13), rnd(B, Cout, seed=14)
    xi = x.float().permute(0, 3, 1, 2)
    if mode == 3:
        xi = F.interpolate(xi, scale_factor=2.0, mode="nearest")
    y = F.conv2d(xi, w.float().permute(0, 3, 1, 2), bias.float(), stride=2 if mode == 2 else 1, padding=1)
    want = bfr(bfr(y) + b2.float()[:, :, None, None]).permute(0, 2, 3, 1)
    force(cfg)
    got = ops.conv3x3_nhwc(x, w, bias=bias, bias2=b2, mode=mode)
    check(got.reshape(want.shape), want, f"conv cfg {cfg} {H}^2 {Cin}->{Cout} mode {mode}")


def test_logits_rows_with_unaligned_stride(force):
    """More than 16 rows onto a [M, 32274] buffer (stride % 4 == 2: Emu2-Chat's vocabulary; 4 prompts x 5 beams): the
    GEMM epilogue falls back to scalar stores instead of rejecting the launch."""
    from emu_amd import ops
    M, N, K = 20, 32274, 512
    x, w = rnd(M, K, seed=21), rnd(N, K, seed=22, scale=0.05)
    for cfg in "0BK":
        force(cfg)
        out = torch.zeros(M, N, dtype=BF16, device="cuda")
        ops.linear(x, w, out=out)
        check(out, ref_linear(x, w, None, None, 0), f"logits cfg {cfg}")


@pytest.mark.parametrize("M,N,K,epis", [(2048, 1280, 1280, (0, 1, 3, 4)), (2048, 2560, 1280, (2, 5)), (770, 1536, 6656, (0, 1, 2)),
                                        (1025, 1792, 2048, (1, 4)), (8192, 640, 640, (1,)), (300, 384, 512, (0, 1, 5)),
                                        (2048, 1288, 1280, (0, 1))])
def test_staged_epilogue_is_bit_identical_to_the_direct_one(force, M, N, K, epis):
    """Round 4: the results of a whole tile leave through LDS as row-contiguous 16-byte stores (and the residual tile comes in by
    LDS-DMA) instead of 8 bytes per lane to 32 different rows (gemm_tile.h::EpiStage).  Same arithmetic, same rounding points: every
    tile configuration must give the bits of the direct epilogue (emu_gemm_tune bit 3 switches the staged form off), on full tiles,
    ragged rows (M = 770, 1025, 300), ragged column tiles (N = 1288, 384: the last tile column keeps the direct form) and with an
    output / residual row stride wider than N."""
    from emu_amd import ops
    from emu_amd._lib import lib
    L = lib()
    x, w = rnd(M, K, seed=31), rnd(N, K, seed=32, scale=K ** -0.5)
    try:
        for epi in epis:
            bias = rnd(N, seed=33) if epi in (0, 1, 4, 5) else None
            wide = N + 64
            res_full = rnd(M, wide, seed=34) if epi == 1 else None
            res = res_full[:, :N] if epi == 1 else None                       # row stride != N
            nout = N // 2 if epi in (2, 5) else N
            for cfg in CFGS:
                force(cfg)
                outs = []
                for tune in (8, 0):
                    L.emu_gemm_tune(tune | ((1 << 22) if cfg == "W" else 0))
                    buf = torch.full((M, nout + 8), float("nan"), dtype=BF16, device="cuda")
                    out = buf[:, :nout]
                    ops.linear(x, w, bias=bias, res=res, epi=epi, out=out)
                    torch.cuda.synchronize()
                    assert bool(torch.isnan(buf[:, nout:].float()).all()), f"cfg {cfg} epi {epi} tune {tune}: wrote beyond N"
                    outs.append(out.clone())
                assert torch.equal(outs[0], outs[1]), (f"cfg {cfg} M{M} N{N} K{K} epi {epi}: staged != direct, "
                                                       f"{int((outs[0] != outs[1]).sum())} elements")
            check(outs[1], ref_linear(x, w, bias, res, epi), f"staged M{M} N{N} K{K} epi{epi}")
    finally:
        L.emu_gemm_tune(0)


@pytest.mark.parametrize("B,H,Cin,Cout,mode", [(2, 32, 1280, 1280, 1), (2, 64, 640, 640, 2), (2, 32, 640, 640, 3)])
def test_staged_conv_epilogue_is_bit_identical_to_the_direct_one(force, B, H, Cin, Cout, mode):
    from emu_amd import ops
    from emu_amd._lib import lib
    L = lib()
    x, w = rnd(B, H, H, Cin, seed=41), rnd(Cout, 3, 3, Cin, seed=42, scale=0.02)
    bias, b2 = rnd(Cout, seed=43), rnd(B, Cout, seed=44)
    Ho = H // 2 if mode == 2 else (2 * H if mode == 3 else H)
    res = rnd(B, Ho, Ho, Cout, seed=45)
    try:
        for cfg in CFGS:
            force(cfg)
            outs = []
            for tune in (8, 0):
                L.emu_gemm_tune(tune | ((1 << 22) if cfg == "W" else 0))
                outs.append(ops.conv3x3_nhwc(x, w, bias=bias, bias2=b2, res=res, mode=mode).clone())
            assert torch.equal(outs[0], outs[1]), f"conv cfg {cfg} mode {mode}: staged != direct"
    finally:
        L.emu_gemm_tune(0)


@pytest.mark.parametrize("M,N,K,epi", [(770, 1536, 6656, 1), (770, 1536, 17920, 1), (1544, 1536, 17920, 0), (1025, 1792, 15360, 1),
                                       (544, 768, 4096, 1), (2048, 1280, 5120, 1)])
def test_four_wave_tile_slices_through_lds_equal_direct_slices_and_the_ping_pong_tile(M, N, K, epi):
    """Round 6 (gemm_w4.hip): the fp32 K-slices of the four-wave 256 x 256 tile leave through LDS, 128 columns at a time, as whole
    512-byte row segments (emu_gemm_tune bit 23 = direct 16-byte stores from the accumulators instead).  Same values, same slab
    layout, same reduce launch: the bits of the direct form -- and of the eight-wave ping-pong tile (bit 21), whose k order and
    rounding points the new tile keeps -- on the S = 770 / 1544 o_proj / down_proj shapes (remainder rows included), the ViT's fc2 and
    the UNet's ff-out."""
    from emu_amd import ops
    from emu_amd._lib import lib
    L = lib()
    sk = torch.zeros(256 * 288 * 256, dtype=torch.float32, device="cuda")
    L.emu_set_splitk_scratch(sk.data_ptr(), sk.numel() * 4)
    x, w = rnd(M, K, seed=51), rnd(N, K, seed=52, scale=K ** -0.5)
    bias, res = rnd(N, seed=53), (rnd(M, N, seed=54) if epi == 1 else None)
    try:
        L.emu_gemm_force_config(ord("P"))
        outs = []
        for tune in ((1 << 22), (1 << 22) | (1 << 23), (1 << 21)):
            L.emu_gemm_tune(tune)
            outs.append(ops.linear(x, w, bias=bias, res=res, epi=epi).clone())
        assert torch.equal(outs[0], outs[1]), f"M{M} N{N} K{K}: slices through LDS != direct slices ({int((outs[0] != outs[1]).sum())} elements)"
        assert torch.equal(outs[0], outs[2]), f"M{M} N{N} K{K}: four-wave tile != ping-pong tile ({int((outs[0] != outs[2]).sum())} elements)"
        check(outs[0], ref_linear(x, w, bias, res, epi), f"four-wave sliced M{M} N{N} K{K}")
    finally:
        L.emu_gemm_tune(0)
        L.emu_gemm_force_config(0)
        L.emu_set_splitk_scratch(0, 0)


def test_four_wave_tile_equals_the_ping_pong_tile_on_random_shapes():
    """48 seeded random problems through both 256 x 256 tiles (emu_gemm_tune bit 22 / bit 21 under config 'P'): ragged and remainder
    rows (M mod 256 in 1 .. 16 and beyond), columns that end inside a tile, 1 .. 40 k tiles (fewer tiles than the ring has slots,
    odd counts, counts that are not multiples of the five ring positions), every epilogue, with and without K-slices.  Equal bits."""
    import random
    from emu_amd import ops
    from emu_amd._lib import lib
    L = lib()
    sk = torch.zeros(256 * 288 * 256, dtype=torch.float32, device="cuda")
    L.emu_set_splitk_scratch(sk.data_ptr(), sk.numel() * 4)
    rng = random.Random(20260930)
    try:
        L.emu_gemm_force_config(ord("P"))
        for case in range(48):
            M = rng.choice([256, 512, 768, 257, 258, 264, 272, 273, 300, 511, 770, 1025, 200, 1300])
            N = rng.choice([256, 512, 768, 1024, 1280, 320, 384, 1000, 1288, 2048])
            K = 64 * rng.choice([1, 2, 3, 4, 5, 6, 7, 9, 10, 11, 16, 21, 32, 40, 64, 96])
            epi = rng.choice([0, 0, 1, 1, 2, 3, 4, 5])
            if epi in (2, 5) and N % 2:
                N += 1
            x, w = rnd(M, K, seed=100 + case), rnd(N, K, seed=200 + case, scale=K ** -0.5)
            bias = rnd(N, seed=300 + case) if epi in (0, 1, 4) else None
            res = rnd(M, N, seed=400 + case) if epi == 1 else None
            outs = []
            for tune in ((1 << 22), (1 << 21)):
                L.emu_gemm_tune(tune)
                outs.append(ops.linear(x, w, bias=bias, res=res, epi=epi).clone())
            assert torch.equal(outs[0], outs[1]), (f"case {case} M{M} N{N} K{K} epi{epi}: four-wave != ping-pong, "
                                                   f"{int((outs[0] != outs[1]).sum())} elements")
            check(outs[0], ref_linear(x, w, bias, res, epi), f"random case {case} M{M} N{N} K{K} epi{epi}")
    finally:
        L.emu_gemm_tune(0)
        L.emu_gemm_force_config(0)
        L.emu_set_splitk_scratch(0, 0)
