"""Shared tiny-model helpers for the tests: rebuild configs + synthetic weights from the
``cfg_*`` entries stored in a golden fixture."""
import numpy as np
import torch

from emu_amd import synth
from emu_amd.conf.emu_conf import CLIPVisionCfg, LlamaCfg
from oracle import emu2_ref as R


def load(golden_dir, name):
    z = np.load(f"{golden_dir}/{name}")
    return {k: z[k] for k in z.files}


def cfgs_from(z):
    g = lambda k: z["cfg_" + k].item()
    v = CLIPVisionCfg(image_size=g("image_size"), patch_size=g("patch_size"), width=g("width"),
                      layers=g("layers"), head_width=g("head_width"), mlp_ratio=g("mlp_ratio"),
                      n_query=g("n_query"), v_query=g("v_query"))
    l = LlamaCfg(hidden_size=g("hidden"), intermediate_size=g("ffn"), num_attention_heads=g("heads"),
                 num_hidden_layers=g("llayers"))
    vocab = 32274 if g("instruct") else 32272
    return v, l, vocab, g("seed"), g("lm_head_scale")


def oracle_cfg(v: CLIPVisionCfg, l: LlamaCfg, vocab: int) -> R.EmuCfg:
    return R.EmuCfg(
        vit=R.VitCfg(image_size=v.image_size, patch_size=v.patch_size, width=v.width, layers=v.layers,
                     head_width=v.head_width, mlp_hidden=v.mlp_hidden),
        llama=R.LlamaCfg(hidden=l.hidden_size, heads=l.num_attention_heads, layers=l.num_hidden_layers,
                         ffn=l.intermediate_size, vocab=vocab, rms_eps=l.rms_norm_eps,
                         rope_theta=l.rope_theta, max_pos=l.max_position_embeddings),
        n_query=v.n_query, v_query=v.v_query)


def weights_from(z, dtype=torch.float32):
    v, l, vocab, seed, lmh = cfgs_from(z)
    W = synth.synth_state_dict(synth.emu_param_shapes(v, l, vocab), seed=seed, lm_head_scale=lmh)
    return v, l, vocab, {k: t.to(dtype) for k, t in W.items()}
