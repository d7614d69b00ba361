"""CPU restatement of the Emu1 caption path (BASELINE.json configs[0])  -- TEST INFRASTRUCTURE ONLY.

**Parity pinned** against the REAL ``Emu1/models`` classes, imported on CPU by oracle/make_golden_emu1.py behind import shims
(timm / peft stubs, two symbols and one mixin method newer transformers dropped, ``T5Config.from_pretrained("t5-base")``
answered offline with a t5-base-kind config at tiny width, a temp ``./models/llama_config`` with the reference tokenizer):
* EVA-CLIP-g ``forward_features`` -> ``ln_visual`` -> ``CausalFormer``: equal to fp32 round-off (tests/golden/emu1_tiny.npz);
* ``Emu.generate`` of the real class run as inference.py runs it (whole model bf16): greedy ids and the default 5-beam
  search (length_penalty 0) reproduced exactly, in fp32 and in bf16 arithmetic (tests/golden/emu1_generate_tiny.npz).
Checked by tests/test_oracle_golden.py on any machine.  Sources restated:

* ``Emu.generate``                 Emu1/models/modeling_emu.py:100-185  (ViT-g -> ln_visual -> CausalFormer -> scatter at
                                   the 32 <image> slots -> LLaMA generate, default num_beams=5, length_penalty=0)
* EVA-CLIP-g ``forward_features``  Emu1/models/eva_vit_model.py:636-665; pre-norm blocks :409-416; config
                                   Emu1/models/Emu-14B.json (40 layers, width 1408, 16 heads x 88, mlp 6144)
* ``CausalFormer``                 Emu1/models/causal_former.py:15-62: 32 learned queries through a T5-base DECODER
                                   (12 blocks: causal self-attention with the bucketed relative-position bias of block 0
                                   shared by all blocks, cross-attention whose k/v read the 1408-wide image features, ReLU
                                   FFN; no 1/sqrt(d) scaling) + final T5LayerNorm + Linear(768 -> 5120)
                                   (Emu1/models/modeling_t5.py:309-331, 407-689, 766-905, 1235-1341)
The LLaMA decoder is the same arithmetic as oracle/emu2_ref.py (LLaMA-13B shape).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from . import emu2_ref as R

Tensor = torch.Tensor
Weights = Dict[str, Tensor]


@dataclass
class T5Cfg:
    """t5-base decoder constants: d_model 768, 12 layers, 12 heads, d_kv 64, d_ff 3072, relu."""
    d_model: int = 768
    layers: int = 12
    heads: int = 12
    d_kv: int = 64
    d_ff: int = 3072
    buckets: int = 32
    max_distance: int = 128
    eps: float = 1e-6
    n_causal: int = 32


@dataclass
class Emu1Cfg:
    vit: R.VitCfg = field(default_factory=lambda: R.VitCfg(image_size=224, patch_size=14, width=1408, layers=40,
                                                           head_width=88, mlp_hidden=6144))
    t5: T5Cfg = field(default_factory=T5Cfg)
    llama: R.LlamaCfg = field(default_factory=lambda: R.LlamaCfg(hidden=5120, heads=40, layers=40, ffn=13824, vocab=32006))


def vit_g_forward(image: Tensor, W: Weights, cfg: R.VitCfg) -> Tensor:
    """eva_vit_model.py:636-665 with pre-norm blocks (:409-416): x += attn(LN1(x)); x += mlp(LN2(x)); no final norm."""
    x = R.vit_patch_embed(image, W)
    cls = W["visual.cls_token"].expand(x.shape[0], -1, -1)
    x = torch.cat((cls, x), dim=1) + W["visual.pos_embed"]
    C = x.shape[-1]
    for i in range(cfg.layers):
        p = f"visual.blocks.{i}."
        h = F.layer_norm(x, (C,), W[p + "norm1.weight"], W[p + "norm1.bias"], cfg.ln_eps)
        x = x + R.vit_attention(h, W, i, cfg)            # xformers attention == softmax(q k^T * d^-0.5) v
        h = F.layer_norm(x, (C,), W[p + "norm2.weight"], W[p + "norm2.bias"], cfg.ln_eps)
        x = x + R.vit_mlp(h, W, i)
    return x


def t5_norm(x: Tensor, w: Tensor, eps: float) -> Tensor:
    """T5LayerNorm, modeling_t5.py:309-331 (RMS norm, fp32 variance, cast to the weight dtype before the gain)."""
    v = x.to(torch.float32).pow(2).mean(-1, keepdim=True)
    h = x * torch.rsqrt(v + eps)
    return w * h.to(w.dtype)


def relative_position_bucket(rel: Tensor, num_buckets: int, max_distance: int) -> Tensor:
    """T5Attention._relative_position_bucket with bidirectional=False (decoder), modeling_t5.py:456-510."""
    rel = -torch.min(rel, torch.zeros_like(rel))
    max_exact = num_buckets // 2
    is_small = rel < max_exact
    large = max_exact + (torch.log(rel.float() / max_exact) / math.log(max_distance / max_exact)
                         * (num_buckets - max_exact)).to(torch.long)
    large = torch.min(large, torch.full_like(large, num_buckets - 1))
    return torch.where(is_small, rel, large)


def self_position_bias(n: int, table: Tensor, cfg: T5Cfg) -> Tensor:
    """compute_bias (modeling_t5.py:512-535) + the causal decoder mask: [heads, n, n] in the table's dtype."""
    ctx = torch.arange(n)[:, None]
    mem = torch.arange(n)[None, :]
    bucket = relative_position_bucket(mem - ctx, cfg.buckets, cfg.max_distance)
    bias = table[bucket].permute(2, 0, 1)                                        # [heads, n, n]
    causal = torch.where(mem <= ctx, 0.0, torch.finfo(table.dtype).min).to(table.dtype)
    return bias + causal[None]


def t5_attention(x: Tensor, kv: Tensor, W: Weights, p: str, cfg: T5Cfg, bias: Optional[Tensor]) -> Tensor:
    """T5Attention.forward, modeling_t5.py:537-689: no 1/sqrt(d) scaling; softmax in fp32, cast back."""
    B, N, _ = x.shape
    H, D = cfg.heads, cfg.d_kv
    q = F.linear(x, W[p + "q.weight"]).view(B, N, H, D).transpose(1, 2)
    k = F.linear(kv, W[p + "k.weight"]).view(B, kv.shape[1], H, D).transpose(1, 2)
    v = F.linear(kv, W[p + "v.weight"]).view(B, kv.shape[1], H, D).transpose(1, 2)
    s = q @ k.transpose(2, 3)
    if bias is not None:
        s = s + bias[None]
    a = torch.softmax(s.float(), dim=-1).type_as(s) @ v
    return F.linear(a.transpose(1, 2).reshape(B, N, H * D), W[p + "o.weight"])


def causal_former(img_feats: Tensor, W: Weights, cfg: T5Cfg) -> Tensor:
    """CausalFormer.forward, causal_former.py:43-62 over T5Stack(decoder), modeling_t5.py:1235-1341."""
    B = img_feats.shape[0]
    x = W["cformer.causal_tokens"].expand(B, -1, -1).to(img_feats.dtype)
    n = x.shape[1]
    bias = self_position_bias(n, W["cformer.cformer.block.0.layer.0.SelfAttention.relative_attention_bias.weight"], cfg)
    for i in range(cfg.layers):
        b = f"cformer.cformer.block.{i}.layer."
        h = t5_norm(x, W[b + "0.layer_norm.weight"], cfg.eps)
        x = x + t5_attention(h, h, W, b + "0.SelfAttention.", cfg, bias)
        h = t5_norm(x, W[b + "1.layer_norm.weight"], cfg.eps)
        x = x + t5_attention(h, img_feats, W, b + "1.EncDecAttention.", cfg, None)
        h = t5_norm(x, W[b + "2.layer_norm.weight"], cfg.eps)
        x = x + F.linear(F.relu(F.linear(h, W[b + "2.DenseReluDense.wi.weight"])), W[b + "2.DenseReluDense.wo.weight"])
    x = t5_norm(x, W["cformer.cformer.final_layer_norm.weight"], cfg.eps)
    return F.linear(x, W["cformer.projection.weight"], W["cformer.projection.bias"])


def encode_image(image: Tensor, W: Weights, cfg: Emu1Cfg) -> Tensor:
    """modeling_emu.py:124-126: ViT-g features -> ln_visual -> CausalFormer -> [B, 32, llama hidden]."""
    f = vit_g_forward(image, W, cfg.vit)
    f = F.layer_norm(f, (f.shape[-1],), W["ln_visual.weight"], W["ln_visual.bias"], 1e-6)
    return causal_former(f, W, cfg.t5)


def emu1_generate(input_ids: Tensor, attention_mask: Tensor, image: Optional[Tensor], W: Weights, cfg: Emu1Cfg,
                  max_new_tokens: int, num_beams: int = 1, min_len: int = 1, length_penalty: float = 0.0,
                  return_margins: bool = False):
    """Emu.generate at the token-id level (modeling_emu.py:134-181): scatter the 32 visual tokens at the <image>
    (id 32003) slots, then LLaMA generate (greedy or beam)."""
    x = R.embed_tokens(input_ids, W)
    if image is not None:
        e = encode_image(image, W, cfg)
        x = R.scatter_image_embeds(x, input_ids, e.reshape(-1, e.shape[-1]))
    if num_beams > 1:
        return R.beam_search_generate(x, attention_mask, W, cfg.llama, num_beams, max_new_tokens, min_len, length_penalty,
                                      return_margin=return_margins)
    return R.greedy_generate(x, attention_mask, W, cfg.llama, max_new_tokens, min_len, return_margins)
