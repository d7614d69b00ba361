"""CPU restatement of the VAE decode step of the Emu2 visual decoder (TEST INFRASTRUCTURE ONLY).

**PARITY UNPINNED** for the same reason as oracle/unet_ref.py: ``diffusers==0.24.0`` (``AutoencoderKL``) is third-party,
not vendored, not installed.  Anchors in the reference: ``EmuVisualGeneration.decode_latents``
(``Emu2/emu/diffusion.py:214-219``: latents / scaling_factor -> vae.decode -> (x/2+0.5).clamp(0,1)) and
``Emu2/emu/conf/diffusion_config/vae/config.json:1-32`` (block_out_channels [128,256,512,512], 2 layers per block,
4 latent channels, GroupNorm 32, SiLU, scaling_factor 0.13025).  State-dict keys follow diffusers' decoder module names.
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


@dataclass
class VaeCfg:
    latent_channels: int = 4
    out_channels: int = 3
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    groups: int = 32
    eps: float = 1e-6
    scaling_factor: float = 0.13025


def decoder_plan(cfg: VaeCfg):
    """(in, out, has_upsampler) per up block of diffusers' Decoder."""
    rev = list(reversed(cfg.block_out_channels))
    plan, out = [], rev[0]
    for i, c in enumerate(rev):
        prev, out = out, c
        plan.append((prev, out, i < len(rev) - 1))
    return plan


def _res_shapes(s, p, cin, cout):
    s[p + "norm1.weight"] = (cin,); s[p + "norm1.bias"] = (cin,)
    s[p + "conv1.weight"] = (cout, cin, 3, 3); s[p + "conv1.bias"] = (cout,)
    s[p + "norm2.weight"] = (cout,); s[p + "norm2.bias"] = (cout,)
    s[p + "conv2.weight"] = (cout, cout, 3, 3); s[p + "conv2.bias"] = (cout,)
    if cin != cout:
        s[p + "conv_shortcut.weight"] = (cout, cin, 1, 1); s[p + "conv_shortcut.bias"] = (cout,)


def vae_decoder_param_shapes(cfg: VaeCfg = VaeCfg()) -> "OrderedDict[str, Tuple[int, ...]]":
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    L, top = cfg.latent_channels, cfg.block_out_channels[-1]
    s["post_quant_conv.weight"] = (L, L, 1, 1); s["post_quant_conv.bias"] = (L,)
    s["decoder.conv_in.weight"] = (top, L, 3, 3); s["decoder.conv_in.bias"] = (top,)
    _res_shapes(s, "decoder.mid_block.resnets.0.", top, top)
    a = "decoder.mid_block.attentions.0."
    s[a + "group_norm.weight"] = (top,); s[a + "group_norm.bias"] = (top,)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        s[a + n + ".weight"] = (top, top); s[a + n + ".bias"] = (top,)
    _res_shapes(s, "decoder.mid_block.resnets.1.", top, top)
    for i, (cin, cout, ups) in enumerate(decoder_plan(cfg)):
        for j in range(cfg.layers_per_block + 1):
            _res_shapes(s, f"decoder.up_blocks.{i}.resnets.{j}.", cin if j == 0 else cout, cout)
        if ups:
            s[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"] = (cout, cout, 3, 3)
            s[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"] = (cout,)
    c0 = cfg.block_out_channels[0]
    s["decoder.conv_norm_out.weight"] = (c0,); s["decoder.conv_norm_out.bias"] = (c0,)
    s["decoder.conv_out.weight"] = (cfg.out_channels, c0, 3, 3); s["decoder.conv_out.bias"] = (cfg.out_channels,)
    return s


def _resnet(x, W, p, cfg):
    h = F.silu(F.group_norm(x, cfg.groups, W[p + "norm1.weight"], W[p + "norm1.bias"], cfg.eps))
    h = F.conv2d(h, W[p + "conv1.weight"], W[p + "conv1.bias"], padding=1)
    h = F.silu(F.group_norm(h, cfg.groups, W[p + "norm2.weight"], W[p + "norm2.bias"], cfg.eps))
    h = F.conv2d(h, W[p + "conv2.weight"], W[p + "conv2.bias"], padding=1)
    if p + "conv_shortcut.weight" in W:
        x = F.conv2d(x, W[p + "conv_shortcut.weight"], W[p + "conv_shortcut.bias"])
    return x + h


def _mid_attention(x, W, p, cfg):
    """diffusers Attention built from the deprecated AttentionBlock: 1 head of C dims, GroupNorm, biased q/k/v/out, residual."""
    B, C, H, Wd = x.shape
    h = F.group_norm(x, cfg.groups, W[p + "group_norm.weight"], W[p + "group_norm.bias"], cfg.eps)
    h = h.view(B, C, H * Wd).transpose(1, 2)
    q = F.linear(h, W[p + "to_q.weight"], W[p + "to_q.bias"])
    k = F.linear(h, W[p + "to_k.weight"], W[p + "to_k.bias"])
    v = F.linear(h, W[p + "to_v.weight"], W[p + "to_v.bias"])
    a = torch.softmax((q @ k.transpose(1, 2)).float() * (C ** -0.5), dim=-1).to(q.dtype) @ v
    a = F.linear(a, W[p + "to_out.0.weight"], W[p + "to_out.0.bias"])
    return a.transpose(1, 2).reshape(B, C, H, Wd) + x


def vae_decode(z: Tensor, W: Dict[str, Tensor], cfg: VaeCfg = VaeCfg()) -> Tensor:
    """AutoencoderKL.decode(z).sample: post_quant_conv -> Decoder (conv_in, mid block, 4 up blocks, GN+SiLU, conv_out)."""
    h = F.conv2d(z, W["post_quant_conv.weight"], W["post_quant_conv.bias"])
    h = F.conv2d(h, W["decoder.conv_in.weight"], W["decoder.conv_in.bias"], padding=1)
    h = _resnet(h, W, "decoder.mid_block.resnets.0.", cfg)
    h = _mid_attention(h, W, "decoder.mid_block.attentions.0.", cfg)
    h = _resnet(h, W, "decoder.mid_block.resnets.1.", cfg)
    for i, (cin, cout, ups) in enumerate(decoder_plan(cfg)):
        for j in range(cfg.layers_per_block + 1):
            h = _resnet(h, W, f"decoder.up_blocks.{i}.resnets.{j}.", cfg)
        if ups:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            p = f"decoder.up_blocks.{i}.upsamplers.0.conv"
            h = F.conv2d(h, W[p + ".weight"], W[p + ".bias"], padding=1)
    h = F.silu(F.group_norm(h, cfg.groups, W["decoder.conv_norm_out.weight"], W["decoder.conv_norm_out.bias"], cfg.eps))
    return F.conv2d(h, W["decoder.conv_out.weight"], W["decoder.conv_out.bias"], padding=1)


def decode_latents(latents: Tensor, W: Dict[str, Tensor], cfg: VaeCfg = VaeCfg()) -> Tensor:
    """EmuVisualGeneration.decode_latents, Emu2/emu/diffusion.py:214-219, up to the clamp (NCHW in [0, 1])."""
    image = vae_decode(latents / cfg.scaling_factor, W, cfg)
    return (image / 2 + 0.5).clamp(0, 1)
