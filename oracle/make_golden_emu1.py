"""Freeze outputs of the REAL Emu1 reference modules into tests/golden/emu1_tiny.npz  -- TEST INFRASTRUCTURE ONLY.

Run in the build container:  python -m oracle.make_golden_emu1

``/root/reference/Emu1/models`` is imported on CPU with four small shims (nothing of the model arithmetic is touched):
``timm.models.layers`` (three helpers eva_vit_model.py:11 imports), two head-pruning symbols that newer transformers
removed from ``transformers.pytorch_utils`` and ``transformers.utils.model_parallel_utils`` (modeling_t5.py:37-51, unused
in a forward), ``peft`` (prediction_mixin.py:1), and ``T5Config.from_pretrained("t5-base")`` (causal_former.py:25, a
network fetch) answered with a T5Config of the t5-base *kind* (relu FFN, 32 buckets, max distance 128, eps 1e-6, d_kv 64)
at tiny width.  The EVA-CLIP-g tower (``_build_vision_tower``, Emu-14B.json flags, pre-norm) and ``CausalFormer`` are
built at the tiny sizes of tests/test_gpu_emu1.py, loaded with ``emu_amd.synth`` weights under the reference's own
parameter names (strict), and ``visual.forward_features`` / ``ln_visual`` / ``cformer`` outputs are stored in fp32.
The fixture pins oracle/emu1_ref.py (tests/test_oracle_golden.py) and the HIP engines (tests/test_gpu_emu1.py).
"""
import importlib.machinery
import json
import os
import sys
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emu_amd import synth  # noqa: E402
from emu_amd.emu1 import T5DecoderCfg, cformer_param_shapes, emu1_vision_cfg  # noqa: E402

REF = "/root/reference/Emu1"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
TINY = dict(image_size=56, width=176, layers=2, head_width=88, mlp_ratio=2.0, d_model=128, t5_layers=2, t5_heads=2,
            d_ff=256, n_causal=8, out_dim=256, seed=4)


def install_shims(t5_kwargs):
    import transformers  # noqa: F401
    import transformers.pytorch_utils as pu
    from transformers.models.t5.configuration_t5 import T5Config

    def _unused(*a, **k):
        raise NotImplementedError("head pruning is not part of the forward pass")
    for name in ("find_pruneable_heads_and_indices", "prune_linear_layer"):
        if not hasattr(pu, name):
            setattr(pu, name, _unused)
    try:
        import transformers.utils.model_parallel_utils  # noqa: F401
    except Exception:
        m = types.ModuleType("transformers.utils.model_parallel_utils")
        m.assert_device_map = lambda *a, **k: None
        m.get_device_map = lambda *a, **k: None
        sys.modules["transformers.utils.model_parallel_utils"] = m
    if "timm" not in sys.modules or not hasattr(sys.modules.get("timm.models.layers", None), "trunc_normal_"):
        timm = types.ModuleType("timm")
        timm.__spec__ = importlib.machinery.ModuleSpec("timm", None)
        tm, tl, tl2 = types.ModuleType("timm.models"), types.ModuleType("timm.models.layers"), types.ModuleType("timm.layers")
        for mod in (tl, tl2):
            mod.to_2tuple = lambda x: x if isinstance(x, tuple) else (x, x)
            mod.drop_path = lambda x, p=0.0, training=False: x
            mod.trunc_normal_ = lambda t, std=1.0, **k: torch.nn.init.trunc_normal_(t, std=std)
        sys.modules.update({"timm": timm, "timm.models": tm, "timm.models.layers": tl, "timm.layers": tl2})
    try:
        import peft  # noqa: F401
    except Exception:
        pf = types.ModuleType("peft")
        pf.PeftModel = type("PeftModel", (), {})
        sys.modules["peft"] = pf
    # causal_former.py:25 fetches the t5-base config from the hub; answer with the same KIND of config at tiny width
    def _t5_base_kind(cls, *a, **k):
        cfg = T5Config(**t5_kwargs)
        # generic PretrainedConfig defaults that transformers 4.31 (the reference's pin) provided and newer releases dropped
        for name, val in (("add_cross_attention", False), ("is_decoder", False), ("use_cache", True),
                          ("output_attentions", False), ("output_hidden_states", False), ("use_return_dict", True),
                          ("tie_word_embeddings", True), ("chunk_size_feed_forward", 0), ("pruned_heads", {}),
                          ("torchscript", False)):
            try:
                getattr(cfg, name)
            except AttributeError:
                setattr(cfg, name, val)
        return cfg
    T5Config.from_pretrained = classmethod(_t5_base_kind)
    # ModuleUtilsMixin.get_head_mask (transformers 4.31) -- with head_mask=None it is just a list of Nones
    from transformers.modeling_utils import PreTrainedModel
    if not hasattr(PreTrainedModel, "get_head_mask"):
        def get_head_mask(self, head_mask, num_hidden_layers, is_attention_chunked=False):
            if head_mask is not None:
                raise NotImplementedError("head masks are not used by the caption path")
            return [None] * num_hidden_layers
        PreTrainedModel.get_head_mask = get_head_mask
    if REF not in sys.path:
        sys.path.insert(0, REF)


def main():
    t = TINY
    t5_kwargs = dict(d_model=t["d_model"], d_kv=64, d_ff=t["d_ff"], num_layers=t["t5_layers"],
                     num_decoder_layers=t["t5_layers"], num_heads=t["t5_heads"], relative_attention_num_buckets=32,
                     relative_attention_max_distance=128, dropout_rate=0.0, layer_norm_epsilon=1e-6,
                     feed_forward_proj="relu", vocab_size=32128)
    install_shims(t5_kwargs)
    from models.causal_former import CausalFormer
    from models.model import CLIPVisionCfg as RefVisionCfg, _build_vision_tower
    from models.transformer import LayerNorm

    ref_json = json.load(open(os.path.join(REF, "models", "Emu-14B.json")))
    vkw = dict(ref_json["vision_cfg"])
    vkw.update(image_size=t["image_size"], width=t["width"], layers=t["layers"], head_width=t["head_width"],
               mlp_ratio=t["mlp_ratio"], xattn=False)                 # xattn = xformers kernels; same math without them
    visual = _build_vision_tower(embed_dim=ref_json["embed_dim"], vision_cfg=RefVisionCfg(**vkw)).eval().float()
    ln_visual = LayerNorm(t["width"], eps=1e-6).eval().float()
    cformer = CausalFormer(args=None, n_causal=t["n_causal"], vision_width=t["width"], output_dim=t["out_dim"]).eval().float()

    # product-side names/shapes for the same modules (emu_amd/emu1.py, emu_amd/synth.py)
    v = emu1_vision_cfg(image_size=t["image_size"], width=t["width"], layers=t["layers"], head_width=t["head_width"],
                        mlp_ratio=t["mlp_ratio"])
    t5 = T5DecoderCfg(d_model=t["d_model"], num_layers=t["t5_layers"], num_heads=t["t5_heads"], d_ff=t["d_ff"],
                      n_causal=t["n_causal"])
    shapes = synth.vit_param_shapes(v)
    shapes["ln_visual.weight"] = (v.width,)
    shapes["ln_visual.bias"] = (v.width,)
    shapes.update(cformer_param_shapes(t5, v.width, t["out_dim"]))
    W = synth.synth_state_dict(shapes, seed=t["seed"])

    def load(module, prefix):
        sd = module.state_dict()
        mine = {k[len(prefix):]: val for k, val in W.items() if k.startswith(prefix)}
        extra = sorted(set(mine) - set(sd))
        assert not extra, f"{prefix}: product names unknown to the reference: {extra[:6]}"
        for k in sd:
            if k in mine:
                assert tuple(sd[k].shape) == tuple(mine[k].shape), (prefix + k, tuple(sd[k].shape), tuple(mine[k].shape))
        missing = sorted(k for k in sd if k not in mine)
        module.load_state_dict({**{k: sd[k] for k in missing}, **mine}, strict=True)
        return missing

    miss_v = load(visual, "visual.")
    load(ln_visual, "ln_visual.")
    miss_c = load(cformer, "cformer.")
    # parameters of the reference that the caption path never reads (classification head, final norm of the tower ...):
    print("reference-only visual params (unused by forward_features):", miss_v)
    print("reference-only cformer params:", miss_c)

    image = torch.randn(2, 3, t["image_size"], t["image_size"], generator=torch.Generator().manual_seed(21))
    with torch.no_grad():
        feats = visual.forward_features(image)                       # modeling_emu.py:83 / :126
        lnv = ln_visual(feats)
        out = cformer(lnv)
    meta = {"cfg_" + k: np.array(val) for k, val in t.items()}
    os.makedirs(OUT, exist_ok=True)
    np.savez(os.path.join(OUT, "emu1_tiny.npz"), image=image.numpy(), feats=feats.numpy(), ln_visual=lnv.numpy(),
             cformer=out.numpy(), **meta)
    print("feats", tuple(feats.shape), "cformer", tuple(out.shape), float(out.abs().mean()))
    generate_fixture(t, ref_json, vkw, W, image)


def generate_fixture(t, ref_json, vkw, W_vis, image):
    """Emu.generate (modeling_emu.py:100-185) of the REAL class at tiny sizes, exactly as inference.py drives it: whole
    model in bf16, greedy and the default 5-beam search.  The LLaMA wrapper reads ./models/llama_config relative to the
    working directory (modeling_llama.py:6,128), so a temp directory with the reference tokenizer files and a shrunken
    config.json stands in for it."""
    import argparse
    import shutil
    import tempfile
    from emu_amd.emu1 import emu1_llama_cfg
    from models.modeling_emu import Emu
    lh, lf, lheads, ll = 256, 512, 2, 2
    tmp = tempfile.mkdtemp(prefix="emu1_tiny_")
    cfgdir = os.path.join(tmp, "models", "llama_config")
    os.makedirs(cfgdir)
    src = os.path.join(REF, "models", "llama_config")
    for f in os.listdir(src):
        if f != "config.json":
            shutil.copy(os.path.join(src, f), cfgdir)
    cfg = json.load(open(os.path.join(src, "config.json")))
    cfg.update(hidden_size=lh, intermediate_size=lf, num_attention_heads=lheads, num_hidden_layers=ll)
    json.dump(cfg, open(os.path.join(cfgdir, "config.json"), "w"))
    cwd = os.getcwd()
    os.chdir(tmp)
    try:
        args = argparse.Namespace(instruct=False, device=torch.device("cpu"))
        mm = dict(ref_json["multimodal_cfg"]); mm["n_causal"] = t["n_causal"]
        va = dict(ref_json["vladapter_cfg"]); va["n_causal"] = t["n_causal"]
        emu = Emu(embed_dim=ref_json["embed_dim"], multimodal_cfg=mm, vision_cfg=vkw, vladapter_cfg=va,
                  cast_dtype=torch.float, args=args).eval()
    finally:
        os.chdir(cwd)
    emu.decoder.lm.config._attn_implementation = "eager"              # fp32-softmax eager attention = transformers 4.31
    vocab = len(emu.decoder.tokenizer)
    l = emu1_llama_cfg(hidden_size=lh, intermediate_size=lf, num_attention_heads=lheads, num_hidden_layers=ll)
    W = dict(W_vis)
    W.update(synth.synth_state_dict(synth.llama_param_shapes(l, vocab), seed=t["seed"], lm_head_scale=8.0))
    sd = emu.state_dict()
    extra = sorted(set(W) - set(sd))
    assert not extra, f"product names unknown to the reference Emu: {extra[:6]}"
    for k, val in W.items():
        assert tuple(sd[k].shape) == tuple(val.shape), (k, tuple(sd[k].shape), tuple(val.shape))
    print("reference-only Emu params:", sorted(k for k in sd if k not in W))
    emu.load_state_dict({**sd, **W}, strict=True)
    emu = emu.to(torch.bfloat16)                                      # inference.py runs the whole model in bf16
    tok = emu.decoder.tokenizer
    prompt = [emu.image_placeholder + "a photo of"]
    captured = {}
    orig = tok.batch_decode

    def hook(ids, **kw):
        captured["ids"] = ids.clone()
        return orig(ids, **kw)
    tok.batch_decode = hook
    outs = {}
    try:
        for name, nb in (("greedy", 1), ("beam", 5)):
            with torch.no_grad():
                txt = emu.generate({"image": image[:1], "prompt": prompt}, num_beams=nb, max_new_tokens=6)
            outs[name] = captured["ids"].numpy()
            print(name, outs[name].tolist(), txt)
        # num_captions (= num_return_sequences) and no_repeat_ngram_size (modeling_emu.py:110,115,173,176), round 3
        with torch.no_grad():
            txt = emu.generate({"image": image[:1], "prompt": prompt}, num_beams=5, max_new_tokens=8, num_captions=2,
                               no_repeat_ngram_size=2)
        outs["beam_cap2_ngram2"] = captured["ids"].numpy()
        print("beam_cap2_ngram2", outs["beam_cap2_ngram2"].tolist(), txt)
    finally:
        tok.batch_decode = orig
    tok.padding_side = "left"
    enc = tok(prompt, padding="longest", return_tensors="pt", add_special_tokens=True)
    meta = {"cfg_" + k: np.array(val) for k, val in t.items()}
    meta.update(cfg_vocab=np.array(vocab), cfg_lhidden=np.array(lh), cfg_lffn=np.array(lf), cfg_lheads=np.array(lheads),
                cfg_llayers=np.array(ll))
    np.savez(os.path.join(OUT, "emu1_generate_tiny.npz"), image=image[:1].numpy(), ids=enc.input_ids.numpy(),
             mask=enc.attention_mask.numpy(), greedy=outs["greedy"], beam=outs["beam"], beam_cap2_ngram2=outs["beam_cap2_ngram2"],
             **meta)
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
