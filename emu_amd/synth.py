"""Deterministic synthetic Emu2 weights (there is no network for real checkpoints).

Parameter names and shapes are exactly the reference's state-dict keys
(``visual.*``, ``decoder.lm.model.*``, ``decoder.lm.lm_head.weight``, ``project_up.weight``,
``project_down.weight``; SURVEY section 5 "Checkpoint / resume"), so a synthetic state dict
loads into the reference ``EmuModel`` with ``load_state_dict(strict=True)`` and into
``emu_amd.EmuModel`` alike.

Each tensor is drawn from its own generator seeded with ``crc32(name) ^ seed`` so any
subset can be regenerated independently, on CPU (bit-reproducible; used for parity tests and
golden fixtures) or directly on the GPU (used for the full 37B bench model, where generating
65 GB on the host would not fit).
"""
from __future__ import annotations

import zlib
from collections import OrderedDict
from typing import Dict, Iterator, Tuple

import torch

from .conf.emu_conf import CLIPVisionCfg, LlamaCfg


def vit_param_shapes(v: CLIPVisionCfg) -> "OrderedDict[str, Tuple[int, ...]]":
    C, F, p = v.width, v.mlp_hidden, v.patch_size
    out: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    out["visual.cls_token"] = (1, 1, C)
    out["visual.pos_embed"] = (1, v.tokens, C)
    out["visual.patch_embed.proj.weight"] = (C, 3, p, p)
    out["visual.patch_embed.proj.bias"] = (C,)
    for i in range(v.layers):
        b = f"visual.blocks.{i}."
        out[b + "norm1.weight"] = (C,)
        out[b + "norm1.bias"] = (C,)
        out[b + "attn.q_bias"] = (C,)
        out[b + "attn.v_bias"] = (C,)
        out[b + "attn.qkv.weight"] = (3 * C, C)
        out[b + "attn.proj.weight"] = (C, C)
        out[b + "attn.proj.bias"] = (C,)
        out[b + "norm2.weight"] = (C,)
        out[b + "norm2.bias"] = (C,)
        out[b + "mlp.fc1.weight"] = (F, C)
        out[b + "mlp.fc1.bias"] = (F,)
        out[b + "mlp.fc2.weight"] = (C, F)
        out[b + "mlp.fc2.bias"] = (C,)
    return out


def llama_param_shapes(l: LlamaCfg, vocab: int) -> "OrderedDict[str, Tuple[int, ...]]":
    H, F = l.hidden_size, l.intermediate_size
    out: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    out["decoder.lm.model.embed_tokens.weight"] = (vocab, H)
    for i in range(l.num_hidden_layers):
        b = f"decoder.lm.model.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "o_proj"):
            out[b + f"self_attn.{n}.weight"] = (H, H)
        out[b + "mlp.gate_proj.weight"] = (F, H)
        out[b + "mlp.up_proj.weight"] = (F, H)
        out[b + "mlp.down_proj.weight"] = (H, F)
        out[b + "input_layernorm.weight"] = (H,)
        out[b + "post_attention_layernorm.weight"] = (H,)
    out["decoder.lm.model.norm.weight"] = (H,)
    out["decoder.lm.lm_head.weight"] = (vocab, H)
    return out


def emu_param_shapes(v: CLIPVisionCfg, l: LlamaCfg, vocab: int) -> "OrderedDict[str, Tuple[int, ...]]":
    out = vit_param_shapes(v)
    out.update(llama_param_shapes(l, vocab))
    out["project_up.weight"] = (l.hidden_size, v.width)
    out["project_down.weight"] = (v.width, l.hidden_size)
    return out


def _is_norm_weight(name: str) -> bool:
    return name.endswith(("norm1.weight", "norm2.weight", "layernorm.weight", "model.norm.weight"))


def synth_tensor(name: str, shape: Tuple[int, ...], seed: int = 0, device="cpu",
                 dtype=torch.float32, std: float = 0.02, lm_head_scale: float = 1.0) -> torch.Tensor:
    """One synthetic parameter: N(0, std^2) for linears/convs/embeddings/biases (matches
    ``initializer_range`` 0.02), 1 + N(0, 0.1^2) for norm gains so the gain multiply is tested."""
    g = torch.Generator(device=device)
    g.manual_seed((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    t = torch.empty(shape, device=device, dtype=torch.float32 if torch.device(device).type == "cpu" else dtype)
    t.normal_(0.0, 1.0, generator=g)
    if _is_norm_weight(name):
        t.mul_(0.1).add_(1.0)
    else:
        t.mul_(std * (lm_head_scale if name.endswith("lm_head.weight") else 1.0))
    return t.to(dtype)


def synth_state_dict(shapes: Dict[str, Tuple[int, ...]], seed: int = 0, device="cpu",
                     dtype=torch.float32, lm_head_scale: float = 1.0) -> "OrderedDict[str, torch.Tensor]":
    return OrderedDict((n, synth_tensor(n, s, seed, device, dtype, lm_head_scale=lm_head_scale))
                       for n, s in shapes.items())


def iter_synth(shapes: Dict[str, Tuple[int, ...]], seed: int = 0, device="cpu", dtype=torch.float32,
               lm_head_scale: float = 1.0) -> Iterator[Tuple[str, torch.Tensor]]:
    """Streaming variant for models too large to hold twice (packers consume and drop)."""
    for n, s in shapes.items():
        yield n, synth_tensor(n, s, seed, device, dtype, lm_head_scale=lm_head_scale)
