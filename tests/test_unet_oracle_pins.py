"""What of the UNet / Euler / VAE restatement (oracle/unet_ref.py, oracle/vae_ref.py) can be pinned WITHOUT diffusers.

diffusers==0.24.0 (Emu2/requirements.txt:13) cannot be installed here, so the restatement cannot be run against the
package itself (that part stays "parity unpinned").  These CPU tests pin it against things that do not come from the
restatement's own code path:

* independent torch.nn module twins of the diffusers building blocks (ResnetBlock2D, Attention + AttnProcessor2_0 on
  F.scaled_dot_product_attention, GEGLU / FeedForward, BasicTransformerBlock, Transformer2DModel) that carry diffusers'
  attribute names -- the oracle's weight dicts must load into them with strict=True (names + shapes) and produce the same
  outputs (norm eps defaults, GEGLU half order, attention scale, residual placement);
* published constants of the SDXL / Stable Diffusion family: parameter counts (SDXL-base UNet 2,567,463,684; SD
  AutoencoderKL decoder 49,490,179), the scaled-linear noise schedule's sigma range (k-diffusion: 0.0292 .. 14.6146), and
  a float64 closed-form recomputation of the Euler tables the reference's scheduler_config.json selects.
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from oracle import unet_ref as U
from oracle import vae_ref as V


# ----------------------------------------------------------------------------------------------- torch.nn twins
class ResnetBlock2D(nn.Module):
    def __init__(self, cin, cout, temb, groups=32, eps=1e-5):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb, cout)
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.nonlinearity = nn.SiLU()
        if cin != cout:
            self.conv_shortcut = nn.Conv2d(cin, cout, 1)

    def forward(self, x, temb):
        h = self.conv1(self.nonlinearity(self.norm1(x)))
        h = h + self.time_emb_proj(self.nonlinearity(temb))[:, :, None, None]
        h = self.conv2(self.nonlinearity(self.norm2(h)))
        if hasattr(self, "conv_shortcut"):
            x = self.conv_shortcut(x)
        return x + h


class Attention(nn.Module):
    def __init__(self, c, kv_dim, heads):
        super().__init__()
        self.heads = heads
        self.to_q = nn.Linear(c, c, bias=False)
        self.to_k = nn.Linear(kv_dim, c, bias=False)
        self.to_v = nn.Linear(kv_dim, c, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(c, c), nn.Dropout(0.0)])

    def forward(self, x, ctx=None):
        ctx = x if ctx is None else ctx
        B, N, C = x.shape
        split = lambda t: t.view(B, -1, self.heads, C // self.heads).transpose(1, 2)
        o = F.scaled_dot_product_attention(split(self.to_q(x)), split(self.to_k(ctx)), split(self.to_v(ctx)))
        return self.to_out[1](self.to_out[0](o.transpose(1, 2).reshape(B, N, C)))


class GEGLU(nn.Module):
    def __init__(self, c, inner):
        super().__init__()
        self.proj = nn.Linear(c, inner * 2)

    def forward(self, x):
        hidden, gate = self.proj(x).chunk(2, dim=-1)
        return hidden * F.gelu(gate)


class FeedForward(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(c, 4 * c), nn.Dropout(0.0), nn.Linear(4 * c, c)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


class BasicTransformerBlock(nn.Module):
    def __init__(self, c, cross, heads):
        super().__init__()
        self.norm1, self.norm2, self.norm3 = nn.LayerNorm(c), nn.LayerNorm(c), nn.LayerNorm(c)
        self.attn1 = Attention(c, c, heads)
        self.attn2 = Attention(c, cross, heads)
        self.ff = FeedForward(c)

    def forward(self, h, ctx):
        h = self.attn1(self.norm1(h)) + h
        h = self.attn2(self.norm2(h), ctx) + h
        return self.ff(self.norm3(h)) + h


class Transformer2DModel(nn.Module):
    def __init__(self, c, cross, heads, depth, groups=32):
        super().__init__()
        self.norm = nn.GroupNorm(groups, c, eps=1e-6)
        self.proj_in = nn.Linear(c, c)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(c, cross, heads) for _ in range(depth)])
        self.proj_out = nn.Linear(c, c)

    def forward(self, x, ctx):
        B, C, H, W = x.shape
        h = self.norm(x).permute(0, 2, 3, 1).reshape(B, H * W, C)
        h = self.proj_in(h)
        for blk in self.transformer_blocks:
            h = blk(h, ctx)
        h = self.proj_out(h)
        return h.reshape(B, H, W, C).permute(0, 3, 1, 2).contiguous() + x


def close(got, want, tol=1e-4):
    return float((got - want).abs().max()) <= tol * max(1.0, float(want.abs().max()))


def _weights(shapes, prefix, seed):
    g = torch.Generator().manual_seed(seed)
    W = {}
    for k, s in shapes.items():
        if k.startswith(prefix):
            t = torch.randn(*s, generator=g) * (0.2 if len(s) > 1 else 0.3)
            W[k] = t + (1.0 if k.endswith("norm.weight") or ".norm" in k and k.endswith("weight") and len(s) == 1 else 0.0)
    return W


def _sub(W, prefix):
    return {k[len(prefix):]: v for k, v in W.items() if k.startswith(prefix)}


CFG = U.UNetCfg(block_out_channels=(64, 128, 256), transformer_layers=(1, 1, 2), heads=(1, 2, 4), cross_dim=96,
                proj_class_in=96 + 6 * 256)


def test_resnet_block_twin_matches_and_keys_load_strict():
    shapes = U.unet_param_shapes(CFG)
    for p, cin, cout in (("down_blocks.1.resnets.0.", 64, 128), ("up_blocks.0.resnets.2.", 384, 256), ("mid_block.resnets.0.", 256, 256)):
        W = _weights(shapes, p, seed=1)
        twin = ResnetBlock2D(cin, cout, CFG.temb_dim)
        twin.load_state_dict(_sub(W, p), strict=True)                  # diffusers attribute names, exact shapes
        x, temb = torch.randn(2, cin, 6, 5), torch.randn(2, CFG.temb_dim)
        with torch.no_grad():
            want = twin(x, temb)
        got = U.resnet_block(x, temb, W, p, CFG)
        assert close(got, want, 1e-5), float((got - want).abs().max())


def test_transformer_2d_twin_matches_and_keys_load_strict():
    shapes = U.unet_param_shapes(CFG)
    for p, c, heads, depth in (("down_blocks.1.attentions.0.", 128, 2, 1), ("mid_block.attentions.0.", 256, 4, 2)):
        W = _weights(shapes, p, seed=2)
        twin = Transformer2DModel(c, CFG.cross_dim, heads, depth)
        twin.load_state_dict(_sub(W, p), strict=True)
        x, ctx = torch.randn(2, c, 4, 6), torch.randn(2, 7, CFG.cross_dim)
        with torch.no_grad():
            want = twin(x, ctx)
        got = U.transformer_2d(x, ctx, W, p, depth, heads, CFG)
        assert close(got, want), float((got - want).abs().max())


def test_twins_at_the_true_width_one_block_of_each_kind():
    """The same twins at the reference's TRUE configuration (unet/config.json: 1280 channels, 20 heads x 64, cross-attention dim
    1792, GEGLU 1280 -> 10240, ten BasicTransformerBlocks per Transformer2DModel at the 32 x 32 level; 640 channels, 10 heads, depth 2
    at 64 x 64): one ResnetBlock2D of each shape class (plain, channel-changing with the 1x1 shortcut, the widest skip-concat input)
    and one Transformer2DModel of each level, on a small spatial extent (the arithmetic per token does not depend on it).  The oracle's
    weight dict must load strict=True under the diffusers key names of the TRUE shapes and give the twin's outputs."""
    cfg = U.UNetCfg()
    shapes = U.unet_param_shapes(cfg)
    for p, cin, cout in (("mid_block.resnets.0.", 1280, 1280), ("down_blocks.2.resnets.0.", 640, 1280), ("up_blocks.0.resnets.0.", 2560, 1280),
                         ("up_blocks.2.resnets.2.", 640, 320)):
        W = _weights(shapes, p, seed=31)
        W = {k: (v * (1.0 / max(1, v[0].numel()) ** 0.5 / 0.2) if v.dim() > 1 else v) for k, v in W.items()}   # fan-in scale
        twin = ResnetBlock2D(cin, cout, cfg.temb_dim)
        twin.load_state_dict(_sub(W, p), strict=True)
        x, temb = torch.randn(2, cin, 3, 4), torch.randn(2, cfg.temb_dim)
        with torch.no_grad():
            want = twin(x, temb)
        got = U.resnet_block(x, temb, W, p, cfg)
        assert close(got, want, 1e-4), (p, float((got - want).abs().max()))
    for p, c, heads, depth in (("down_blocks.1.attentions.1.", 640, 10, 2), ("mid_block.attentions.0.", 1280, 20, 10)):
        W = _weights(shapes, p, seed=32)
        W = {k: (v * (1.0 / max(1, v[0].numel()) ** 0.5 / 0.2) if v.dim() > 1 else v) for k, v in W.items()}   # fan-in scale: O(1) through 10 blocks
        twin = Transformer2DModel(c, cfg.cross_dim, heads, depth)
        twin.load_state_dict(_sub(W, p), strict=True)
        x, ctx = torch.randn(2, c, 3, 4), torch.randn(2, 64, cfg.cross_dim)
        with torch.no_grad():
            want = twin(x, ctx)
        got = U.transformer_2d(x, ctx, W, p, depth, heads, cfg)
        assert got.shape == want.shape == (2, c, 3, 4)
        assert close(got, want, 2e-4), (p, float((got - want).abs().max()))


def test_attention_twin_against_nn_multihead_attention():
    """The restated attention also equals torch.nn.MultiheadAttention (a third implementation) on shared weights."""
    c, heads = 128, 2
    W = {f"a.to_{n}.weight": torch.randn(c, c) * 0.1 for n in "qkv"}
    W["a.to_out.0.weight"], W["a.to_out.0.bias"] = torch.randn(c, c) * 0.1, torch.randn(c) * 0.1
    mha = nn.MultiheadAttention(c, heads, bias=True, batch_first=True)
    with torch.no_grad():
        mha.in_proj_weight.copy_(torch.cat([W["a.to_q.weight"], W["a.to_k.weight"], W["a.to_v.weight"]]))
        mha.in_proj_bias.zero_()
        mha.out_proj.weight.copy_(W["a.to_out.0.weight"])
        mha.out_proj.bias.copy_(W["a.to_out.0.bias"])
        x = torch.randn(2, 9, c)
        want, _ = mha(x, x, x, need_weights=False)
    got = U.attention(x, x, W, "a", heads)
    assert close(got, want), float((got - want).abs().max())


def test_geglu_half_order_and_gelu_form():
    """hidden = first half, gate = second half of ff.net.0.proj; exact (erf) GELU, not the tanh approximation."""
    c = 64
    twin = FeedForward(c)
    W = {"b.ff.net.0.proj.weight": twin.net[0].proj.weight.detach(), "b.ff.net.0.proj.bias": twin.net[0].proj.bias.detach(),
         "b.ff.net.2.weight": twin.net[2].weight.detach(), "b.ff.net.2.bias": twin.net[2].bias.detach()}
    x = torch.randn(3, 5, c) * 3
    hid, gate = U._lin(x, W, "b.ff.net.0.proj").chunk(2, dim=-1)
    got = U._lin(hid * F.gelu(gate), W, "b.ff.net.2")
    with torch.no_grad():
        want = twin(x)
    assert torch.allclose(got, want, rtol=1e-5, atol=1e-5)
    swapped = U._lin(gate * F.gelu(hid), W, "b.ff.net.2")
    assert not torch.allclose(swapped, want, rtol=1e-3, atol=1e-3)      # the check can tell the two orders apart
    tanh = U._lin(hid * F.gelu(gate, approximate="tanh"), W, "b.ff.net.2")
    assert float((tanh - want).abs().max()) > float((got - want).abs().max())


def test_timestep_embedding_layout():
    """flip_sin_to_cos=True, downscale_freq_shift=0: [cos | sin], frequencies exp(-ln(1e4) * i / half)."""
    t = torch.tensor([981.0, 1.0])
    e = U.timestep_embedding(t, 320)
    i = torch.arange(160, dtype=torch.float64)
    f = torch.exp(-math.log(10000.0) * i / 160)
    want = torch.cat([torch.cos(t.double()[:, None] * f), torch.sin(t.double()[:, None] * f)], dim=-1)
    assert torch.allclose(e.double(), want, atol=2e-4)
    assert abs(float(e[1, 0]) - math.cos(1.0)) < 1e-6 and abs(float(e[1, 160]) - math.sin(1.0)) < 1e-6


# ----------------------------------------------------------------------------------------------- published constants
def test_parameter_counts_match_the_published_sdxl_family_numbers():
    """SDXL-base UNet2DConditionModel has 2,567,463,684 parameters.  The reference's unet/config.json differs from SDXL-base
    in two fields only: cross_attention_dim 1792 instead of 2048 (to_k / to_v of every cross-attention: 10 blocks at 640
    channels, 60 at 1280) and projection_class_embeddings_input_dim 3328 instead of 2816 (add_embedding.linear_1)."""
    n = sum(int(np.prod(s)) for s in U.unet_param_shapes(U.UNetCfg()).values())
    sdxl = 2_567_463_684
    kv = 2 * (2048 - 1792) * (10 * 640 + 60 * 1280)
    add = (3328 - 2816) * 1280
    assert n == sdxl - kv + add == 2_525_520_644
    # Stable Diffusion's AutoencoderKL: 83,653,863 parameters, of which encoder 34,163,592 + quant_conv 72 and
    # decoder 49,490,179 + post_quant_conv 20
    nv = sum(int(np.prod(s)) for k, s in V.vae_decoder_param_shapes(V.VaeCfg()).items())
    assert nv == 49_490_179 + 20


def test_up_block_channel_plan_of_the_reference_config():
    """SURVEY Appendix B (from unet/config.json): up0 resnets take 2560, 2560, 1920 -> 1280; up1 1920, 1280, 960 -> 640;
    up2 960, 640, 640 -> 320."""
    plan = U.up_block_plan(U.UNetCfg())
    assert [(o, ins) for o, ins, *_ in plan] == [(1280, [2560, 2560, 1920]), (640, [1920, 1280, 960]), (320, [960, 640, 640])]
    assert [(a, d, h, u) for _, _, a, d, h, u in plan] == [(True, 10, 20, True), (True, 2, 10, True), (False, 1, 5, False)]


def test_euler_tables_against_closed_form_and_published_sigma_range():
    """scheduler_config.json: scaled_linear betas 0.00085 .. 0.012, 1000 train steps, leading spacing, steps_offset 1,
    linear interpolation.  k-diffusion / Stable Diffusion quote sigma_min = 0.0292, sigma_max = 14.6146 for this schedule."""
    sch = U.EulerSchedule()
    betas = np.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=np.float64) ** 2
    ac = np.cumprod(1.0 - betas)
    sig64 = np.sqrt((1 - ac) / ac)
    assert abs(sig64[0] - 0.0292) < 5e-5 and abs(sig64[-1] - 14.6146) < 5e-4
    assert np.allclose(sch.train_sigmas, sig64, rtol=2e-5)
    for n in (50, 30, 4):
        s = U.EulerSchedule().set_timesteps(n)
        ts = (np.arange(n) * (1000 // n))[::-1] + 1
        assert s.timesteps.tolist() == ts.astype(np.float32).tolist()
        want = np.concatenate([sig64[ts], [0.0]])                      # integer timesteps: interpolation = table lookup
        assert np.allclose(s.sigmas.numpy(), want, rtol=2e-5, atol=1e-7)
        assert abs(s.init_noise_sigma - math.sqrt(sig64[ts[0]] ** 2 + 1)) < 1e-4
    s50 = U.EulerSchedule().set_timesteps(50)
    assert s50.timesteps[0] == 981 and s50.timesteps[-1] == 1 and float(s50.timesteps[0] - s50.timesteps[1]) == 20
    assert abs(float(s50.sigmas[0]) - 13.1204) < 1e-3                   # sigma(t = 981)
    assert abs(s50.init_noise_sigma - 13.1585) < 1e-3                   # sqrt(sigma^2 + 1): the latents' initial scale


def test_euler_step_is_the_published_update():
    """epsilon prediction, no churn: x_{i+1} = x_i + eps * (sigma_{i+1} - sigma_i); the model input is x / sqrt(sigma^2 + 1)."""
    s = U.EulerSchedule().set_timesteps(10)
    x, eps = torch.randn(1, 4, 8, 8), torch.randn(1, 4, 8, 8)
    for i in (0, 5, 9):
        want = x + eps * (s.sigmas[i + 1] - s.sigmas[i])
        assert torch.allclose(s.step(eps, i, x), want, rtol=1e-5, atol=1e-5)
        assert torch.allclose(s.scale_model_input(x, i), x / math.sqrt(float(s.sigmas[i]) ** 2 + 1), rtol=1e-6, atol=1e-6)


def test_product_schedule_equals_oracle_schedule():
    """emu_amd/unet.py builds its own tables on the host (they ride to the device once): same numbers as the oracle's."""
    from emu_amd.unet import EulerDiscreteSchedule as P
    for n in (50, 20, 3):
        a, b = P().set_timesteps(n), U.EulerSchedule().set_timesteps(n)
        assert a.timesteps.tolist() == b.timesteps.tolist()
        assert np.allclose(np.asarray(a.sigmas), np.asarray(b.sigmas), rtol=1e-6, atol=1e-9)
        assert abs(a.init_noise_sigma - b.init_noise_sigma) < 1e-6
