"""EVA-CLIP ViT engine: host side of the ``emu_vit_*`` C ABI.

Plays the role of ``EVAVisionTransformer`` (reference Emu2/emu/eva_vit.py:338-445) as configured by
``CLIPVisionCfg`` (post-norm blocks, fused qkv with q/v bias, naive attention, GELU MLP, abs pos-embed).
Weights are re-packed once at load time into MFMA-friendly shapes:
  * heads are zero-padded from head_width (112) to 128 so QK^T tiles are whole MFMA k-steps;
  * the patch-embed conv becomes a [C, Kpad] GEMM weight (K = 3*p*p = 588 padded to a multiple of 64).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Iterable, Optional, Tuple

import torch

from . import ops
from ._lib import VitCfgC, check, lib
from .conf.emu_conf import CLIPVisionCfg

BF16 = torch.bfloat16
DP = 128
_BLOCK_KEYS = ("norm1.weight", "norm1.bias", "attn.q_bias", "attn.v_bias", "attn.qkv.weight", "attn.proj.weight",
               "attn.proj.bias", "norm2.weight", "norm2.bias", "mlp.fc1.weight", "mlp.fc1.bias", "mlp.fc2.weight",
               "mlp.fc2.bias")


class VitEngine:
    def __init__(self, cfg: CLIPVisionCfg, ctx):
        """``cfg.postnorm`` True = Emu2's EVA-CLIP-4B blocks; False = Emu1's EVA-CLIP-g pre-norm blocks."""
        if cfg.rope or cfg.naiveswiglu or cfg.subln or cfg.init_value:
            raise NotImplementedError("only the EVA-CLIP configurations Emu uses (GELU MLP, no rope / sub-LN / layer scale) are built")
        self.cfg, self.ctx, self.device = cfg, ctx, ctx.device
        self.kpad = (3 * cfg.patch_size * cfg.patch_size + 63) // 64 * 64
        c = VitCfgC(cfg.image_size, cfg.patch_size, cfg.width, cfg.layers, cfg.heads, cfg.head_width,
                    cfg.mlp_hidden, self.kpad, 1e-6, 0 if cfg.postnorm else 1)
        h = C.c_void_p()
        check(lib().emu_vit_create(ctx.handle, C.byref(c), C.byref(h)), "emu_vit_create", ctx.handle)
        self.handle = h
        self._keep: Dict[str, torch.Tensor] = {}
        self._pending: Dict[int, Dict[str, torch.Tensor]] = {}
        self._stem: Dict[str, torch.Tensor] = {}
        self.blocks_loaded = set()          # indices of the packed blocks (a reload must not count twice)
        self._ws = None

    def _dev(self, t):
        return t.to(device=self.device, dtype=BF16).contiguous()

    def load_tensor(self, name: str, t: torch.Tensor) -> bool:
        """Consume one tensor named relative to ``visual.`` (reference state-dict keys)."""
        if name in ("cls_token", "pos_embed", "patch_embed.proj.weight", "patch_embed.proj.bias"):
            self._stem[name] = t
            if len(self._stem) == 4:
                self._pack_stem()
            return True
        if name.startswith("blocks."):
            idx, key = name[len("blocks."):].split(".", 1)
            if key not in _BLOCK_KEYS:
                return False
            d = self._pending.setdefault(int(idx), {})
            d[key] = t
            if len(d) == len(_BLOCK_KEYS):
                self._pack_block(int(idx), self._pending.pop(int(idx)))
            return True
        return False

    def load_weights(self, items: Iterable[Tuple[str, torch.Tensor]], prefix: str = "visual.") -> None:
        for name, t in items:
            if name.startswith(prefix):
                self.load_tensor(name[len(prefix):], t)

    @property
    def ready(self) -> bool:
        return len(self.blocks_loaded) == self.cfg.layers and "wpatch" in self._keep

    def _pack_stem(self):
        c = self.cfg
        w = self._stem["patch_embed.proj.weight"].to(self.device, BF16).reshape(c.width, -1)
        wp = torch.zeros(c.width, self.kpad, device=self.device, dtype=BF16)
        wp[:, : w.shape[1]] = w
        k = self._keep
        k["wpatch"] = wp
        k["bpatch"] = self._dev(self._stem["patch_embed.proj.bias"])
        k["cls"] = self._dev(self._stem["cls_token"].reshape(-1))
        k["pos"] = self._dev(self._stem["pos_embed"].reshape(c.tokens, c.width))
        check(lib().emu_vit_set_stem(self.handle, k["wpatch"].data_ptr(), k["bpatch"].data_ptr(), k["cls"].data_ptr(),
                                     k["pos"].data_ptr()), "emu_vit_set_stem")
        self._stem = {}

    def _pack_block(self, i: int, d: Dict[str, torch.Tensor]):
        c = self.cfg
        Hh, hw, Cw = c.heads, c.head_width, c.width
        g = lambda k: d[k].to(self.device, BF16)
        # qkv [3*Hh*hw, C] -> [3, Hh, 128, C] zero padded; bias = cat(q_bias, 0, v_bias) (eva_vit.py:194-198)
        wqkv = torch.zeros(3, Hh, DP, Cw, device=self.device, dtype=BF16)
        wqkv[:, :, :hw] = g("attn.qkv.weight").view(3, Hh, hw, Cw)
        bqkv = torch.zeros(3, Hh, DP, device=self.device, dtype=BF16)
        bqkv[0, :, :hw] = g("attn.q_bias").view(Hh, hw)
        bqkv[2, :, :hw] = g("attn.v_bias").view(Hh, hw)
        wproj = torch.zeros(Cw, Hh, DP, device=self.device, dtype=BF16)
        wproj[:, :, :hw] = g("attn.proj.weight").view(Cw, Hh, hw)
        p = {"wqkv": wqkv.view(3 * Hh * DP, Cw), "bqkv": bqkv.view(-1), "wproj": wproj.view(Cw, Hh * DP),
             "bproj": self._dev(d["attn.proj.bias"]), "ln1w": self._dev(d["norm1.weight"]),
             "ln1b": self._dev(d["norm1.bias"]), "fc1w": self._dev(d["mlp.fc1.weight"]),
             "fc1b": self._dev(d["mlp.fc1.bias"]), "fc2w": self._dev(d["mlp.fc2.weight"]),
             "fc2b": self._dev(d["mlp.fc2.bias"]), "ln2w": self._dev(d["norm2.weight"]),
             "ln2b": self._dev(d["norm2.bias"])}
        for k, v in p.items():
            self._keep[f"{i}.{k}"] = v
        order = ("wqkv", "bqkv", "wproj", "bproj", "ln1w", "ln1b", "fc1w", "fc1b", "fc2w", "fc2b", "ln2w", "ln2b")
        check(lib().emu_vit_set_block(self.handle, i, *[p[k].data_ptr() for k in order]), "emu_vit_set_block")
        self.blocks_loaded.add(i)

    @torch.no_grad()
    def forward(self, image: torch.Tensor) -> torch.Tensor:
        """image [B,3,H,W] (fp32 or bf16, already CLIP-normalised) -> tokens [B, 1+g*g, C] bf16."""
        c = self.cfg
        if image.dim() != 4 or image.shape[1] != 3 or image.shape[2] != c.image_size or image.shape[3] != c.image_size:
            raise AssertionError(f"Input image size ({image.shape[2]}*{image.shape[3]}) doesn't match model "
                                 f"({c.image_size}*{c.image_size}).")
        if not self.ready:
            raise RuntimeError("ViT weights not fully loaded")
        img = image.to(self.device)
        if img.dtype not in (torch.float32, BF16):
            img = img.float()
        img = img.contiguous()
        B = img.shape[0]
        need = lib().emu_vit_workspace_bytes(self.handle, B)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, device=self.device, dtype=torch.uint8)
        out = torch.empty(B, c.tokens, c.width, device=self.device, dtype=BF16)
        check(lib().emu_vit_forward(self.handle, img.data_ptr(), int(img.dtype == torch.float32), B, out.data_ptr(),
                                    self._ws.data_ptr(), self._ws.numel(), ops.stream(self.device)), "emu_vit_forward",
              self.ctx.handle)
        return out

    __call__ = forward

    # ------------------------------------------------------------------ optional W8A8 mode (BASELINE configs[4]: fp8 MFMA)
    def quantize_fp8(self) -> None:
        """Per-row-scaled e4m3fn copies of the four packed matrices of every block (the bf16 set stays resident: +4.3 GB for
        EVA-CLIP-4B).  Not a reference feature (the reference is bf16 end to end); off unless ``use_fp8``."""
        if not self.ready:
            raise RuntimeError("quantize_fp8: load all weights first")
        if getattr(self, "_fp8", None):
            return
        self._fp8 = {}
        for i in range(self.cfg.layers):
            a = []
            for k in ("wqkv", "wproj", "fc1w", "fc2w"):
                q, sc = ops.quantize_fp8_rows(self._keep[f"{i}.{k}"])
                self._fp8[f"{i}.{k}"] = (q, sc)
                a += [q.data_ptr(), sc.data_ptr()]
            check(lib().emu_vit_set_block_fp8(self.handle, i, *a), "emu_vit_set_block_fp8", self.ctx.handle)

    def set_fusion(self, mask: int) -> None:
        """Launch fusions of the blocks (bit 0: V^T from the qkv projection's epilogue for a single image, bit 1: fc2's K-slice sum
        applies bias + LayerNorm + residual); 0 = unfused.  All on by default."""
        check(lib().emu_vit_set_fusion(self.handle, int(mask)), "emu_vit_set_fusion", self.ctx.handle)

    def use_fp8(self, enable: bool = True) -> None:
        """Run the blocks' GEMMs W8A8 on the block-scaled fp8 MFMA (activation rows quantised per row ahead of every GEMM);
        LayerNorm, attention and the patch embedding stay bf16."""
        if enable:
            self.quantize_fp8()
        check(lib().emu_vit_use_fp8(self.handle, 1 if enable else 0), "emu_vit_use_fp8", self.ctx.handle)
        self.fp8 = bool(enable)

    def fp8_dequantized(self, key: str) -> torch.Tensor:
        """fp32 value of a registered fp8 matrix, e.g. ``"3.fc1w"`` (tests: the checker runs on the exact weights the GEMMs use)."""
        q, sc = self._fp8[key]
        return q.view(torch.float8_e4m3fn).to(torch.float32) * sc[:, None]

    @torch.no_grad()
    def run_blocks(self, tokens: torch.Tensor, l0: int, l1: int) -> torch.Tensor:
        """Parity hook (include/emu_hip.h: emu_vit_blocks): blocks [l0, l1) on a COPY of tokens [B, 1+g*g, C] bf16."""
        x = tokens.to(device=self.device, dtype=BF16).contiguous().clone()
        B = x.shape[0]
        need = lib().emu_vit_workspace_bytes(self.handle, B)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, device=self.device, dtype=torch.uint8)
        check(lib().emu_vit_blocks(self.handle, x.data_ptr(), B, int(l0), int(l1), self._ws.data_ptr(), self._ws.numel(),
                                   ops.stream(self.device)), "emu_vit_blocks", self.ctx.handle)
        return x
