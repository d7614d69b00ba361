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


def emu1_from(z):
    """Configs + synthetic weights of the Emu1 fixture (tests/golden/emu1_tiny.npz, oracle/make_golden_emu1.py)."""
    from emu_amd.emu1 import T5DecoderCfg, cformer_param_shapes, emu1_vision_cfg
    from oracle import emu1_ref as E
    g = lambda k: z["cfg_" + k].item()
    v = emu1_vision_cfg(image_size=g("image_size"), width=g("width"), layers=g("layers"), head_width=g("head_width"),
                        mlp_ratio=g("mlp_ratio"))
    t5 = T5DecoderCfg(d_model=g("d_model"), num_layers=g("t5_layers"), num_heads=g("t5_heads"), d_ff=g("d_ff"),
                      n_causal=g("n_causal"))
    shapes = synth.vit_param_shapes(v)
    shapes["ln_visual.weight"] = (v.width,)
    shapes["ln_visual.bias"] = (v.width,)
    shapes.update(cformer_param_shapes(t5, v.width, g("out_dim")))
    W = {k: t.float() for k, t in synth.synth_state_dict(shapes, seed=g("seed")).items()}
    ocfg = E.Emu1Cfg(vit=R.VitCfg(image_size=g("image_size"), patch_size=14, width=g("width"), layers=g("layers"),
                                  head_width=g("head_width"), mlp_hidden=v.mlp_hidden),
                     t5=E.T5Cfg(d_model=g("d_model"), layers=g("t5_layers"), heads=g("t5_heads"), d_ff=g("d_ff"),
                                n_causal=g("n_causal")),
                     llama=R.LlamaCfg(hidden=g("out_dim"), heads=2, layers=1, ffn=64, vocab=64))
    return v, t5, g("out_dim"), W, ocfg


def emu1_generate_from(z):
    """Adds the tiny LLaMA of tests/golden/emu1_generate_tiny.npz to the Emu1 fixture weights."""
    from emu_amd.emu1 import emu1_llama_cfg
    g = lambda k: z["cfg_" + k].item()
    v, t5, out_dim, W, ocfg = emu1_from(z)
    l = emu1_llama_cfg(hidden_size=g("lhidden"), intermediate_size=g("lffn"), num_attention_heads=g("lheads"),
                       num_hidden_layers=g("llayers"))
    W.update({k: t.float() for k, t in synth.synth_state_dict(synth.llama_param_shapes(l, g("vocab")), seed=g("seed"),
                                                             lm_head_scale=8.0).items()})
    ocfg.llama = R.LlamaCfg(hidden=g("lhidden"), heads=g("lheads"), layers=g("llayers"), ffn=g("lffn"), vocab=g("vocab"))
    return v, t5, l, g("vocab"), W, ocfg
