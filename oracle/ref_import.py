"""Import the REAL reference (``/root/reference/Emu2/emu``) on CPU  -- TEST INFRASTRUCTURE ONLY.

Works only in the build container (the GPU box has no ``/root/reference``); used by
``oracle/make_golden.py`` to freeze reference outputs into ``tests/golden/`` and by
``tests/test_oracle_golden.py::test_live_reference_*`` (skipped when the reference is absent).

Recipe: import transformers first, install a two-symbol ``timm`` shim
(``eva_vit.py:13-16`` needs only ``drop_path`` and ``to_2tuple``), put ``Emu2`` on sys.path.
"""
import importlib.machinery
import json
import os
import shutil
import sys
import tempfile
import types

REF_ROOT = "/root/reference/Emu2"
REF_LLAMA_CFG = os.path.join(REF_ROOT, "emu/conf/llama_config")


def available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "emu"))


def _install_timm_shim():
    if "timm" in sys.modules:
        return
    timm, layers = types.ModuleType("timm"), types.ModuleType("timm.layers")
    timm.__spec__ = importlib.machinery.ModuleSpec("timm", None)
    layers.__spec__ = importlib.machinery.ModuleSpec("timm.layers", None)
    layers.to_2tuple = lambda x: x if isinstance(x, tuple) else (x, x)
    layers.drop_path = lambda x, p, training: x
    sys.modules["timm"], sys.modules["timm.layers"] = timm, layers


def tiny_llama_dir(hidden, ffn, heads, layers) -> str:
    """Temp dir with the reference tokenizer files + a shrunken config.json."""
    d = tempfile.mkdtemp(prefix="emu_tiny_llama_")
    for f in ("tokenizer.model", "tokenizer_config.json", "special_tokens_map.json", "generation_config.json"):
        shutil.copy(os.path.join(REF_LLAMA_CFG, f), d)
    cfg = json.load(open(os.path.join(REF_LLAMA_CFG, "config.json")))
    cfg.update(hidden_size=hidden, intermediate_size=ffn, num_attention_heads=heads, num_hidden_layers=layers)
    json.dump(cfg, open(os.path.join(d, "config.json"), "w"))
    return d


def build_reference(vit_kwargs: dict, llama_dir: str, instruct: bool, state_dict=None, eager=True):
    """Construct the reference ``EmuModel`` (Emu2/emu/emu.py:19-65) on CPU, fp32, eval."""
    import torch
    import transformers  # noqa: F401  (must be imported before the shim)
    from transformers import LlamaForCausalLM  # noqa: F401
    _install_timm_shim()
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    from emu.emu import EmuModel
    from emu.conf.emu_conf import CLIPVisionCfg, TextDecoderCfg
    m = EmuModel(CLIPVisionCfg(**vit_kwargs), TextDecoderCfg(llama_config_path=llama_dir, instruct=instruct)).eval()
    if eager:
        # fp32-softmax eager attention = the semantics of the pinned transformers 4.31
        m.decoder.lm.config._attn_implementation = "eager"
    if state_dict is not None:
        m.load_state_dict(state_dict, strict=True)
    return m.to(torch.float32)
