"""VAE decoder of the Emu2 visual decoder (diffusers ``AutoencoderKL.decode``; reference call site
``EmuVisualGeneration.decode_latents``, Emu2/emu/diffusion.py:214-219; config conf/diffusion_config/vae/config.json).

Runs once per image, so the host walks the decoder layer by layer over the primitive C-ABI operators (NHWC activations):
GroupNorm+SiLU, implicit-GEMM 3x3 convs (nearest-x2 upsample folded into the gather), 1x1 convs as GEMMs, and the
single-head (D = 512) mid-block attention as two MFMA GEMMs around an in-place row softmax.  Small-channel edges are
zero-padded to the kernels' granularity (conv_in 4 -> 64 input channels, conv_out 3 -> 4 output channels).
PARITY UNPINNED (see oracle/vae_ref.py).
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, Tuple

import torch

from . import ops

BF16 = torch.bfloat16


@dataclass
class VaeCfg:
    latent_channels: int = 4
    out_channels: int = 3
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    eps: float = 1e-6
    scaling_factor: float = 0.13025


def _plan(cfg: VaeCfg):
    rev = list(reversed(cfg.block_out_channels))
    plan, out = [], rev[0]
    for i, c in enumerate(rev):
        prev, out = out, c
        plan.append((prev, out, i < len(rev) - 1))
    return plan


def vae_decoder_param_shapes(cfg: VaeCfg = VaeCfg()) -> "OrderedDict[str, Tuple[int, ...]]":
    """diffusers AutoencoderKL keys used by decode() (encoder / quant_conv are not needed for generation)."""
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    L, top = cfg.latent_channels, cfg.block_out_channels[-1]

    def res(p, cin, cout):
        s[p + "norm1.weight"] = (cin,); s[p + "norm1.bias"] = (cin,)
        s[p + "conv1.weight"] = (cout, cin, 3, 3); s[p + "conv1.bias"] = (cout,)
        s[p + "norm2.weight"] = (cout,); s[p + "norm2.bias"] = (cout,)
        s[p + "conv2.weight"] = (cout, cout, 3, 3); s[p + "conv2.bias"] = (cout,)
        if cin != cout:
            s[p + "conv_shortcut.weight"] = (cout, cin, 1, 1); s[p + "conv_shortcut.bias"] = (cout,)
    s["post_quant_conv.weight"] = (L, L, 1, 1); s["post_quant_conv.bias"] = (L,)
    s["decoder.conv_in.weight"] = (top, L, 3, 3); s["decoder.conv_in.bias"] = (top,)
    res("decoder.mid_block.resnets.0.", top, top)
    a = "decoder.mid_block.attentions.0."
    s[a + "group_norm.weight"] = (top,); s[a + "group_norm.bias"] = (top,)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        s[a + n + ".weight"] = (top, top); s[a + n + ".bias"] = (top,)
    res("decoder.mid_block.resnets.1.", top, top)
    for i, (cin, cout, ups) in enumerate(_plan(cfg)):
        for j in range(cfg.layers_per_block + 1):
            res(f"decoder.up_blocks.{i}.resnets.{j}.", cin if j == 0 else cout, cout)
        if ups:
            s[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"] = (cout, cout, 3, 3)
            s[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"] = (cout,)
    c0 = cfg.block_out_channels[0]
    s["decoder.conv_norm_out.weight"] = (c0,); s["decoder.conv_norm_out.bias"] = (c0,)
    s["decoder.conv_out.weight"] = (cfg.out_channels, c0, 3, 3); s["decoder.conv_out.bias"] = (cfg.out_channels,)
    return s


class VaeDecoder:
    CIN_PAD = 64          # implicit-GEMM convs need Cin % 64 == 0

    def __init__(self, cfg: VaeCfg, ctx):
        self.cfg, self.ctx, self.device = cfg, ctx, ctx.device
        self.w: Dict[str, torch.Tensor] = {}
        self.ready = False

    def load_state_dict(self, sd, prefix: str = "", strict: bool = True):
        want = vae_decoder_param_shapes(self.cfg)
        items = sd.items() if hasattr(sd, "items") else sd
        seen = set()
        dev = lambda t: t.to(self.device, BF16)
        for name, t in items:
            if prefix:
                if not name.startswith(prefix):
                    continue
                name = name[len(prefix):]
            if name not in want:
                continue                                   # encoder.*, quant_conv.* are legitimately unused here
            seen.add(name)
            t = dev(t)
            if name == "decoder.conv_in.weight":           # [C, 4, 3, 3] -> [C, 3, 3, 64] zero padded
                w = torch.zeros(t.shape[0], 3, 3, self.CIN_PAD, device=self.device, dtype=BF16)
                w[..., : t.shape[1]] = t.permute(0, 2, 3, 1)
                t = w
            elif name == "decoder.conv_out.weight":        # [3, C, 3, 3] -> [4, 3, 3, C]
                w = torch.zeros(4, 3, 3, t.shape[1], device=self.device, dtype=BF16)
                w[: t.shape[0]] = t.permute(0, 2, 3, 1)
                t = w
            elif name == "decoder.conv_out.bias":
                b = torch.zeros(4, device=self.device, dtype=BF16)
                b[: t.shape[0]] = t
                t = b
            elif name == "post_quant_conv.weight":         # 1x1: [4, 4] -> [4, 8] (K padded to the 16-byte granule)
                w = torch.zeros(t.shape[0], 8, device=self.device, dtype=BF16)
                w[:, : t.shape[1]] = t.reshape(t.shape[0], t.shape[1])
                t = w
            elif t.dim() == 4 and t.shape[-1] == 3:
                t = t.permute(0, 2, 3, 1)
            elif t.dim() == 4:
                t = t.reshape(t.shape[0], t.shape[1])      # 1x1 shortcut
            self.w[name] = t.contiguous()
        missing = sorted(set(want) - seen)
        if missing and strict:
            raise RuntimeError(f"missing VAE decoder tensors: {missing[:5]}{'...' if len(missing) > 5 else ''}")
        self.ready = not missing
        return missing

    # ------------------------------------------------------------------ blocks (x: [1, H, W, C] NHWC bf16)
    def _gn(self, x, p, silu=True):
        B, H, W, C = x.shape
        return ops.groupnorm_nhwc(x.view(B, H * W, C), self.w[p + ".weight"], self.w[p + ".bias"], self.cfg.norm_num_groups,
                                  self.cfg.eps, silu).view(B, H, W, C)

    def _resnet(self, x, p):
        w = self.w
        h = self._gn(x, p + "norm1")
        h = ops.conv3x3_nhwc(h, w[p + "conv1.weight"], bias=w[p + "conv1.bias"])
        h = self._gn(h, p + "norm2")
        res = x
        if p + "conv_shortcut.weight" in w:
            B, H, W, C = x.shape
            res = ops.linear(x.view(-1, C), w[p + "conv_shortcut.weight"], bias=w[p + "conv_shortcut.bias"]).view(B, H, W, -1)
        return ops.conv3x3_nhwc(h, w[p + "conv2.weight"], bias=w[p + "conv2.bias"], res=res)

    def _attention(self, x, p):
        w = self.w
        B, H, W, C = x.shape
        assert B == 1
        t = self._gn(x, p + "group_norm", silu=False).view(H * W, C)
        q = ops.linear(t, w[p + "to_q.weight"], bias=w[p + "to_q.bias"])
        k = ops.linear(t, w[p + "to_k.weight"], bias=w[p + "to_k.bias"])
        v = ops.linear(t, w[p + "to_v.weight"], bias=w[p + "to_v.bias"])
        s = ops.linear(q, k)                                   # [HW, HW] scores
        ops.softmax_rows_(s, C ** -0.5)
        o = ops.linear(s, v.t().contiguous())                  # P @ V as an NT GEMM against V^T
        return ops.linear(o, w[p + "to_out.0.weight"], bias=w[p + "to_out.0.bias"], res=x.view(H * W, C),
                          epi=ops.EPI_RESID).view(B, H, W, C)

    @torch.no_grad()
    def decode(self, z: torch.Tensor) -> torch.Tensor:
        """z [1, 4, h, w] (already divided by scaling_factor) -> image [1, 3, 8h, 8w] bf16 (AutoencoderKL.decode().sample)."""
        assert self.ready and z.shape[0] == 1
        w, cfg = self.w, self.cfg
        _, L, H, W = z.shape
        x = torch.zeros(H * W, 8, device=self.device, dtype=BF16)
        x[:, :L] = z.to(self.device, BF16)[0].permute(1, 2, 0).reshape(H * W, L)
        x = ops.linear(x, w["post_quant_conv.weight"], bias=w["post_quant_conv.bias"])          # [HW, 4]
        xin = torch.zeros(1, H, W, self.CIN_PAD, device=self.device, dtype=BF16)
        xin[0, :, :, :L] = x.view(H, W, L)
        h = ops.conv3x3_nhwc(xin, w["decoder.conv_in.weight"], bias=w["decoder.conv_in.bias"])
        h = self._resnet(h, "decoder.mid_block.resnets.0.")
        h = self._attention(h, "decoder.mid_block.attentions.0.")
        h = self._resnet(h, "decoder.mid_block.resnets.1.")
        for i, (cin, cout, ups) in enumerate(_plan(cfg)):
            for j in range(cfg.layers_per_block + 1):
                h = self._resnet(h, f"decoder.up_blocks.{i}.resnets.{j}.")
            if ups:
                p = f"decoder.up_blocks.{i}.upsamplers.0.conv"
                h = ops.conv3x3_nhwc(h, w[p + ".weight"], bias=w[p + ".bias"], mode=ops.CONV_3X3_UP2)
        h = self._gn(h, "decoder.conv_norm_out")
        out = ops.conv3x3_nhwc(h, w["decoder.conv_out.weight"], bias=w["decoder.conv_out.bias"])  # [1, 8h, 8w, 4]
        return out[..., : cfg.out_channels].permute(0, 3, 1, 2).contiguous()

    def decode_latents(self, latents: torch.Tensor) -> torch.Tensor:
        """diffusion.py:214-219 up to the clamp: NCHW image in [0, 1] (bf16 ops like the reference's bf16 pipeline)."""
        image = self.decode((latents.to(BF16) * (1.0 / self.cfg.scaling_factor)).to(BF16))
        return (image / 2 + 0.5).clamp(0, 1)
