"""Model configuration objects of the Emu2 path.

``CLIPVisionCfg`` and ``TextDecoderCfg`` keep the reference's interface (``Emu2/emu/conf/emu_conf.py:6-39``: same class
names, keyword names and default values, so ``EmuModel(CLIPVisionCfg(n_query=256), TextDecoderCfg(instruct=True))`` reads
the same on both sides).  They are generated from one table per class that also says what each knob means to the HIP
engines: most EVA-CLIP switches exist only to be accepted, the engines implement exactly one setting of them and refuse
the others at construction time (``emu_amd/vit.py``).  ``LlamaCfg`` carries the decoder shape the reference reads from its
``conf/llama_config/config.json`` (LLaMA-33B by default); a checkpoint's own ``config.json`` overrides it.
"""
import json
import os.path as osp
from dataclasses import dataclass, field, make_dataclass
from typing import Optional

# name, type, default, what the MI355X engines do with it
_VISION_KNOBS = (
    ("eva_model_name", str, "eva-clip-4b-14-x", "label only"),
    ("image_size", int, 448, "input side in pixels (448 / 14 = 32 x 32 patches + cls)"),
    ("patch_size", int, 14, "patch side; the stem is an im2col GEMM with K = 3 * 14 * 14 padded to 592"),
    ("width", int, 1792, "channel width"),
    ("layers", int, 64, "transformer blocks"),
    ("head_width", int, 112, "head dim (padded to 128 lanes inside the attention kernel)"),
    ("mlp_ratio", float, 8.571428571428571, "MLP hidden = int(width * mlp_ratio) = 15360"),
    ("qkv_bias", bool, True, "q and v carry a bias, k does not (zero bias packed for k)"),
    ("drop_path_rate", float, 0.0, "training only: ignored"),
    ("init_value", Optional[float], None, "layer scale: must stay None"),
    ("patch_dropout", float, 0.0, "training only: must stay 0"),
    ("rope", bool, False, "vision RoPE: must stay False"),
    ("global_average_pool", bool, False, "must stay False (all tokens are returned)"),
    ("xattn", bool, False, "xformers switch of the reference: irrelevant here"),
    ("postnorm", bool, True, "True = Emu2's post-norm blocks, False = EVA-CLIP-g pre-norm (Emu1)"),
    ("pt_hw_seq_len", int, 16, "RoPE only: ignored"),
    ("intp_freq", bool, False, "RoPE only: ignored"),
    ("naiveswiglu", bool, False, "must stay False (GELU MLP)"),
    ("subln", bool, False, "must stay False"),
    ("n_query", int, 64, "visual tokens per image handed to the LLaMA (avg-pool to sqrt(n_query)^2)"),
    ("v_query", int, 64, "visual tokens per video frame"),
)


def _vision_namespace():
    def heads(self) -> int:
        return self.width // self.head_width

    def mlp_hidden(self) -> int:
        return int(self.width * self.mlp_ratio)          # the reference truncates the same way (eva_vit.py:270)

    def grid(self) -> int:
        return self.image_size // self.patch_size

    def tokens(self) -> int:
        return self.grid * self.grid + 1

    return {"heads": property(heads), "mlp_hidden": property(mlp_hidden), "grid": property(grid),
            "tokens": property(tokens), "__doc__": "EVA-CLIP vision tower configuration (keyword-compatible with the "
                                                     "reference's CLIPVisionCfg)."}


CLIPVisionCfg = make_dataclass("CLIPVisionCfg", [(n, t, field(default=d)) for n, t, d, _ in _VISION_KNOBS],
                               namespace=_vision_namespace())
CLIPVisionCfg.__module__ = __name__

TextDecoderCfg = make_dataclass(
    "TextDecoderCfg",
    [("llama_config_path", str, field(default=osp.join(osp.dirname(__file__), "llama_config"))),   # tokenizer + config dir
     ("instruct", bool, field(default=False))],                                                    # +[USER]/[ASSISTANT]
    namespace={"__doc__": "Where the decoder's tokenizer / config.json live and whether the chat tokens are added."})
TextDecoderCfg.__module__ = __name__


@dataclass
class LlamaCfg:
    """Decoder shape.  Defaults = LLaMA-33B as shipped with Emu2; ``from_json`` reads a checkpoint's ``config.json``."""
    hidden_size: int = 6656
    intermediate_size: int = 17920
    num_attention_heads: int = 52
    num_hidden_layers: int = 60
    vocab_size: int = 32000            # before the Emu tokenizer extension
    rms_norm_eps: float = 1e-6
    rope_theta: float = 10000.0
    max_position_embeddings: int = 2048

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads

    @classmethod
    def from_json(cls, path: str) -> "LlamaCfg":
        """``path`` = a config.json or the directory holding one; a directory without it (this package ships none)
        means the Emu2 defaults."""
        if osp.isdir(path):
            path = osp.join(path, "config.json")
        if not osp.exists(path):
            return cls()
        with open(path) as f:
            d = json.load(f)
        take = ("hidden_size", "intermediate_size", "num_attention_heads", "num_hidden_layers", "vocab_size", "rms_norm_eps",
                "max_position_embeddings", "rope_theta")
        return cls(**{k: d[k] for k in take if k in d})
