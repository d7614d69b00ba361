"""CPU restatement of the Emu2 visual decoder's denoising path (TEST INFRASTRUCTURE ONLY).

**PARITY UNPINNED.**  The arithmetic lives in the third-party package ``diffusers`` (pinned ==0.24.0 by
``Emu2/requirements.txt:13``), which is neither vendored in ``/root/reference`` nor installed here and cannot be
fetched (no network).  The reference ships no test, golden vector or fixture for this path.  This file restates the
published diffusers-0.24 algorithm of the modules the reference instantiates, anchored on the reference's own call
sites and configs:

* loop, CFG (cond FIRST), ``time_ids``/``text_embeds`` construction: ``Emu2/emu/diffusion.py:77-166``
* UNet topology: ``Emu2/emu/conf/diffusion_config/unet/config.json:1-72`` (SDXL-shaped ``UNet2DConditionModel``:
  channels [320,640,1280], 2 resnets/block, transformer depth [1(unused),2,10], heads [5,10,20] x 64,
  cross_attention_dim 1792, ``text_time`` additional embedding 3328 = 1792 + 6*256, linear projections)
* scheduler: ``.../scheduler/scheduler_config.json:1-18`` (EulerDiscreteScheduler: scaled_linear betas
  0.00085..0.012, 1000 train steps, "leading" spacing, steps_offset 1, epsilon prediction, linear sigma interpolation)

State-dict keys follow diffusers' module names (``conv_in``, ``time_embedding.linear_1``, ``down_blocks.N.resnets.M``,
``...attentions.M.transformer_blocks.K.attn1.to_q`` ...), i.e. what the reference's checkpoint stores under ``unet.``.
Until a diffusers wheel is available to pin it, this restatement IS the spec the HIP path is tested against.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Weights = Dict[str, Tensor]


@dataclass
class UNetCfg:
    """unet/config.json, with the fields the forward pass uses."""
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280)
    layers_per_block: int = 2
    transformer_layers: Tuple[int, ...] = (1, 2, 10)       # per down block; block 0 has no attention
    heads: Tuple[int, ...] = (5, 10, 20)                   # "attention_head_dim" in the json = number of heads
    down_attn: Tuple[bool, ...] = (False, True, True)      # DownBlock2D, CrossAttnDownBlock2D x2
    cross_dim: int = 1792
    groups: int = 32
    gn_eps: float = 1e-5
    addition_time_embed_dim: int = 256
    proj_class_in: int = 3328

    @property
    def temb_dim(self) -> int:
        return self.block_out_channels[0] * 4

    @property
    def head_dim(self) -> int:
        return self.block_out_channels[0] // self.heads[0]


# --------------------------------------------------------------------------- parameter inventory
def _resnet(shapes, p, cin, cout, temb):
    shapes[p + "norm1.weight"] = (cin,); shapes[p + "norm1.bias"] = (cin,)
    shapes[p + "conv1.weight"] = (cout, cin, 3, 3); shapes[p + "conv1.bias"] = (cout,)
    shapes[p + "time_emb_proj.weight"] = (cout, temb); shapes[p + "time_emb_proj.bias"] = (cout,)
    shapes[p + "norm2.weight"] = (cout,); shapes[p + "norm2.bias"] = (cout,)
    shapes[p + "conv2.weight"] = (cout, cout, 3, 3); shapes[p + "conv2.bias"] = (cout,)
    if cin != cout:
        shapes[p + "conv_shortcut.weight"] = (cout, cin, 1, 1); shapes[p + "conv_shortcut.bias"] = (cout,)


def _transformer(shapes, p, c, depth, cross):
    shapes[p + "norm.weight"] = (c,); shapes[p + "norm.bias"] = (c,)
    shapes[p + "proj_in.weight"] = (c, c); shapes[p + "proj_in.bias"] = (c,)
    for k in range(depth):
        b = p + f"transformer_blocks.{k}."
        for n in ("norm1", "norm2", "norm3"):
            shapes[b + n + ".weight"] = (c,); shapes[b + n + ".bias"] = (c,)
        for a, kd in (("attn1", c), ("attn2", cross)):
            shapes[b + a + ".to_q.weight"] = (c, c)
            shapes[b + a + ".to_k.weight"] = (c, kd)
            shapes[b + a + ".to_v.weight"] = (c, kd)
            shapes[b + a + ".to_out.0.weight"] = (c, c); shapes[b + a + ".to_out.0.bias"] = (c,)
        shapes[b + "ff.net.0.proj.weight"] = (8 * c, c); shapes[b + "ff.net.0.proj.bias"] = (8 * c,)
        shapes[b + "ff.net.2.weight"] = (c, 4 * c); shapes[b + "ff.net.2.bias"] = (c,)
    shapes[p + "proj_out.weight"] = (c, c); shapes[p + "proj_out.bias"] = (c,)


def up_block_plan(cfg: UNetCfg):
    """(out_channels, [resnet input channels], has_attn, depth, heads, has_upsampler) per up block, diffusers'
    UNet2DConditionModel.__init__ / get_up_block channel bookkeeping."""
    rev = list(reversed(cfg.block_out_channels))
    rev_depth = list(reversed(cfg.transformer_layers))
    rev_heads = list(reversed(cfg.heads))
    rev_attn = list(reversed(cfg.down_attn))
    plan = []
    out = rev[0]
    n = len(rev)
    for i in range(n):
        prev, out = out, rev[i]
        inp = rev[min(i + 1, n - 1)]
        ins = []
        for j in range(cfg.layers_per_block + 1):
            skip = inp if j == cfg.layers_per_block else out
            first = prev if j == 0 else out
            ins.append(first + skip)
        plan.append((out, ins, rev_attn[i], rev_depth[i], rev_heads[i], i < n - 1))
    return plan


def unet_param_shapes(cfg: UNetCfg = UNetCfg()) -> "OrderedDict[str, Tuple[int, ...]]":
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    ch, T = cfg.block_out_channels, cfg.temb_dim
    s["conv_in.weight"] = (ch[0], cfg.in_channels, 3, 3); s["conv_in.bias"] = (ch[0],)
    s["time_embedding.linear_1.weight"] = (T, ch[0]); s["time_embedding.linear_1.bias"] = (T,)
    s["time_embedding.linear_2.weight"] = (T, T); s["time_embedding.linear_2.bias"] = (T,)
    s["add_embedding.linear_1.weight"] = (T, cfg.proj_class_in); s["add_embedding.linear_1.bias"] = (T,)
    s["add_embedding.linear_2.weight"] = (T, T); s["add_embedding.linear_2.bias"] = (T,)
    cin = ch[0]
    for i, cout in enumerate(ch):
        for j in range(cfg.layers_per_block):
            _resnet(s, f"down_blocks.{i}.resnets.{j}.", cin if j == 0 else cout, cout, T)
            if cfg.down_attn[i]:
                _transformer(s, f"down_blocks.{i}.attentions.{j}.", cout, cfg.transformer_layers[i], cfg.cross_dim)
        if i < len(ch) - 1:
            s[f"down_blocks.{i}.downsamplers.0.conv.weight"] = (cout, cout, 3, 3)
            s[f"down_blocks.{i}.downsamplers.0.conv.bias"] = (cout,)
        cin = cout
    _resnet(s, "mid_block.resnets.0.", ch[-1], ch[-1], T)
    _transformer(s, "mid_block.attentions.0.", ch[-1], cfg.transformer_layers[-1], cfg.cross_dim)
    _resnet(s, "mid_block.resnets.1.", ch[-1], ch[-1], T)
    for i, (out, ins, attn, depth, heads, ups) in enumerate(up_block_plan(cfg)):
        for j, c_in in enumerate(ins):
            _resnet(s, f"up_blocks.{i}.resnets.{j}.", c_in, out, T)
            if attn:
                _transformer(s, f"up_blocks.{i}.attentions.{j}.", out, depth, cfg.cross_dim)
        if ups:
            s[f"up_blocks.{i}.upsamplers.0.conv.weight"] = (out, out, 3, 3)
            s[f"up_blocks.{i}.upsamplers.0.conv.bias"] = (out,)
    s["conv_norm_out.weight"] = (ch[0],); s["conv_norm_out.bias"] = (ch[0],)
    s["conv_out.weight"] = (cfg.out_channels, ch[0], 3, 3); s["conv_out.bias"] = (cfg.out_channels,)
    return s


# --------------------------------------------------------------------------- modules
def timestep_embedding(timesteps: Tensor, dim: int) -> Tensor:
    """diffusers get_timestep_embedding(flip_sin_to_cos=True, downscale_freq_shift=0, scale=1): fp32 [N, dim] = [cos | sin]."""
    half = dim // 2
    exponent = -math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=timesteps.device) / half
    emb = timesteps.to(torch.float32)[:, None] * torch.exp(exponent)[None, :]
    return torch.cat([torch.cos(emb), torch.sin(emb)], dim=-1)


def _lin(x, W, p):
    return F.linear(x, W[p + ".weight"], W.get(p + ".bias"))


def resnet_block(x: Tensor, temb: Tensor, W: Weights, p: str, cfg: UNetCfg) -> Tensor:
    """diffusers ResnetBlock2D (time_embedding_norm='default', output_scale_factor 1)."""
    h = F.silu(F.group_norm(x, cfg.groups, W[p + "norm1.weight"], W[p + "norm1.bias"], cfg.gn_eps))
    h = F.conv2d(h, W[p + "conv1.weight"], W[p + "conv1.bias"], padding=1)
    h = h + _lin(F.silu(temb), W, p + "time_emb_proj")[:, :, None, None]
    h = F.silu(F.group_norm(h, cfg.groups, W[p + "norm2.weight"], W[p + "norm2.bias"], cfg.gn_eps))
    h = F.conv2d(h, W[p + "conv2.weight"], W[p + "conv2.bias"], padding=1)
    if p + "conv_shortcut.weight" in W:
        x = F.conv2d(x, W[p + "conv_shortcut.weight"], W[p + "conv_shortcut.bias"])
    return x + h


def attention(x: Tensor, ctx: Tensor, W: Weights, p: str, heads: int) -> Tensor:
    """diffusers Attention with AttnProcessor2_0 (scaled_dot_product_attention), bias-free q/k/v, biased to_out."""
    B, N, C = x.shape
    q = F.linear(x, W[p + ".to_q.weight"]).view(B, N, heads, -1).transpose(1, 2)
    k = F.linear(ctx, W[p + ".to_k.weight"]).view(B, ctx.shape[1], heads, -1).transpose(1, 2)
    v = F.linear(ctx, W[p + ".to_v.weight"]).view(B, ctx.shape[1], heads, -1).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) * (q.shape[-1] ** -0.5)
    o = (torch.softmax(s.float(), dim=-1).to(q.dtype) @ v).transpose(1, 2).reshape(B, N, C)
    return _lin(o, W, p + ".to_out.0")


def transformer_block(h: Tensor, ctx: Tensor, W: Weights, p: str, heads: int) -> Tensor:
    """diffusers BasicTransformerBlock (layer_norm eps 1e-5, GEGLU feed-forward)."""
    C = h.shape[-1]
    n = F.layer_norm(h, (C,), W[p + "norm1.weight"], W[p + "norm1.bias"], 1e-5)
    h = attention(n, n, W, p + "attn1", heads) + h
    n = F.layer_norm(h, (C,), W[p + "norm2.weight"], W[p + "norm2.bias"], 1e-5)
    h = attention(n, ctx, W, p + "attn2", heads) + h
    n = F.layer_norm(h, (C,), W[p + "norm3.weight"], W[p + "norm3.bias"], 1e-5)
    hid, gate = _lin(n, W, p + "ff.net.0.proj").chunk(2, dim=-1)
    return _lin(hid * F.gelu(gate), W, p + "ff.net.2") + h


def transformer_2d(x: Tensor, ctx: Tensor, W: Weights, p: str, depth: int, heads: int, cfg: UNetCfg) -> Tensor:
    """diffusers Transformer2DModel with use_linear_projection=True (GroupNorm eps 1e-6)."""
    B, C, H, Wd = x.shape
    res = x
    h = F.group_norm(x, cfg.groups, W[p + "norm.weight"], W[p + "norm.bias"], 1e-6)
    h = h.permute(0, 2, 3, 1).reshape(B, H * Wd, C)
    h = _lin(h, W, p + "proj_in")
    for k in range(depth):
        h = transformer_block(h, ctx, W, p + f"transformer_blocks.{k}.", heads)
    h = _lin(h, W, p + "proj_out")
    return h.reshape(B, H, Wd, C).permute(0, 3, 1, 2) + res


def unet_embeddings(timestep: Tensor, text_embeds: Tensor, time_ids: Tensor, W: Weights, cfg: UNetCfg, dtype) -> Tensor:
    """time embedding + 'text_time' additional embedding of UNet2DConditionModel.forward.  ``time_ids`` may be the
    reference's 1-D length-12 tensor (diffusion.py:108-110): it is flattened then reshaped to [B, -1]."""
    B = text_embeds.shape[0]
    t = timestep_embedding(timestep.reshape(-1).expand(B), cfg.block_out_channels[0]).to(dtype)
    emb = _lin(F.silu(_lin(t, W, "time_embedding.linear_1")), W, "time_embedding.linear_2")
    tid = timestep_embedding(time_ids.flatten(), cfg.addition_time_embed_dim).reshape(B, -1)
    add = torch.cat([text_embeds, tid.to(text_embeds.dtype)], dim=-1).to(dtype)
    aug = _lin(F.silu(_lin(add, W, "add_embedding.linear_1")), W, "add_embedding.linear_2")
    return emb + aug


def unet_forward(sample: Tensor, timestep: Tensor, ctx: Tensor, text_embeds: Tensor, time_ids: Tensor, W: Weights,
                 cfg: UNetCfg = UNetCfg()) -> Tensor:
    """UNet2DConditionModel.forward as called at Emu2/emu/diffusion.py:136-141."""
    emb = unet_embeddings(timestep, text_embeds, time_ids, W, cfg, sample.dtype)
    h = F.conv2d(sample, W["conv_in.weight"], W["conv_in.bias"], padding=1)
    skips = [h]
    nb = len(cfg.block_out_channels)
    for i in range(nb):
        for j in range(cfg.layers_per_block):
            h = resnet_block(h, emb, W, f"down_blocks.{i}.resnets.{j}.", cfg)
            if cfg.down_attn[i]:
                h = transformer_2d(h, ctx, W, f"down_blocks.{i}.attentions.{j}.", cfg.transformer_layers[i], cfg.heads[i], cfg)
            skips.append(h)
        if i < nb - 1:
            p = f"down_blocks.{i}.downsamplers.0.conv"
            h = F.conv2d(h, W[p + ".weight"], W[p + ".bias"], stride=2, padding=1)
            skips.append(h)
    h = resnet_block(h, emb, W, "mid_block.resnets.0.", cfg)
    h = transformer_2d(h, ctx, W, "mid_block.attentions.0.", cfg.transformer_layers[-1], cfg.heads[-1], cfg)
    h = resnet_block(h, emb, W, "mid_block.resnets.1.", cfg)
    for i, (out, ins, attn, depth, heads, ups) in enumerate(up_block_plan(cfg)):
        for j in range(len(ins)):
            h = torch.cat([h, skips.pop()], dim=1)
            h = resnet_block(h, emb, W, f"up_blocks.{i}.resnets.{j}.", cfg)
            if attn:
                h = transformer_2d(h, ctx, W, f"up_blocks.{i}.attentions.{j}.", depth, heads, cfg)
        if ups:
            p = f"up_blocks.{i}.upsamplers.0.conv"
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = F.conv2d(h, W[p + ".weight"], W[p + ".bias"], padding=1)
    h = F.silu(F.group_norm(h, cfg.groups, W["conv_norm_out.weight"], W["conv_norm_out.bias"], cfg.gn_eps))
    return F.conv2d(h, W["conv_out.weight"], W["conv_out.bias"], padding=1)


# --------------------------------------------------------------------------- scheduler
class EulerSchedule:
    """diffusers EulerDiscreteScheduler as configured by scheduler_config.json (no churn, epsilon prediction)."""

    def __init__(self, num_train: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.012, steps_offset: int = 1):
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train, dtype=torch.float32) ** 2
        ac = torch.cumprod(1.0 - betas, dim=0)
        self.train_sigmas = (((1 - ac) / ac) ** 0.5).numpy()
        self.num_train, self.offset = num_train, steps_offset

    def set_timesteps(self, n: int):
        ratio = self.num_train // n                                            # "leading" spacing
        ts = (np.arange(0, n) * ratio).round()[::-1].copy().astype(np.float32) + self.offset
        sig = np.interp(ts, np.arange(0, len(self.train_sigmas)), self.train_sigmas)
        self.timesteps = torch.from_numpy(ts)
        self.sigmas = torch.from_numpy(np.concatenate([sig, [0.0]]).astype(np.float32))
        return self

    @property
    def init_noise_sigma(self) -> float:
        return float((self.sigmas.max() ** 2 + 1) ** 0.5)                     # spacing "leading"

    def scale_model_input(self, x: Tensor, i: int) -> Tensor:
        return x / ((self.sigmas[i] ** 2 + 1) ** 0.5)

    def step(self, eps: Tensor, i: int, x: Tensor) -> Tensor:
        sigma = self.sigmas[i]
        pred_x0 = x - sigma * eps
        derivative = (x - pred_x0) / sigma
        return x + derivative * (self.sigmas[i + 1] - sigma)


def denoise(latents: Tensor, prompt_embeds: Tensor, W: Weights, steps: int = 50, guidance: float = 3.0,
            height: int = 1024, width: int = 1024, cfg: UNetCfg = UNetCfg(), original_size=(1024, 1024),
            crop=(0, 0), return_all: bool = False):
    """The loop of EmuVisualGeneration.forward, Emu2/emu/diffusion.py:104-149, for batch 1 with CFG:
    prompt_embeds [2, n, 1792] = (cond, uncond) -- cond FIRST (:202,210,145); latents [1,4,h,w] already
    multiplied by init_noise_sigma (:126-127)."""
    sch = EulerSchedule().set_timesteps(steps)
    time_ids = torch.tensor(list(original_size) + list(crop) + [height, width], dtype=torch.long)
    time_ids = torch.cat([time_ids, time_ids], dim=0).to(latents.device)       # 1-D, length 12 (:108-110)
    text_embeds = prompt_embeds.mean(dim=1)                                    # (:113)
    x = latents
    hist = []
    for i, t in enumerate(sch.timesteps):
        inp = sch.scale_model_input(torch.cat([x] * 2), i)
        eps = unet_forward(inp, t.to(latents.device), prompt_embeds, text_embeds, time_ids, W, cfg)
        e_c, e_u = eps.chunk(2)
        eps = e_u + guidance * (e_c - e_u)
        x = sch.step(eps, i, x)
        if return_all:
            hist.append(x)
    return (x, hist) if return_all else x
