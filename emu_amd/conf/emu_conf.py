"""Model configuration dataclasses.

Field-for-field mirror of the reference's ``Emu2/emu/conf/emu_conf.py:6-39`` (same class
names, field names and defaults) plus ``LlamaCfg`` for the decoder shape that the
reference reads from ``conf/llama_config/config.json``.
"""
from dataclasses import dataclass
from typing import Optional
import json
import os.path as osp


@dataclass
class CLIPVisionCfg:
    eva_model_name: str = "eva-clip-4b-14-x"

    image_size: int = 448
    patch_size: int = 14
    width: int = 1792
    layers: int = 64
    head_width: int = 112
    mlp_ratio: float = 8.571428571428571

    qkv_bias: bool = True
    drop_path_rate: float = 0.

    init_value: Optional[float] = None
    patch_dropout: float = 0.
    rope: bool = False
    global_average_pool: bool = False

    xattn: bool = False
    postnorm: bool = True
    pt_hw_seq_len: int = 16
    intp_freq: bool = False
    naiveswiglu: bool = False
    subln: bool = False

    n_query: int = 64
    v_query: int = 64

    @property
    def heads(self) -> int:
        return self.width // self.head_width

    @property
    def mlp_hidden(self) -> int:
        return int(self.width * self.mlp_ratio)      # reference eva_vit.py:270

    @property
    def grid(self) -> int:
        return self.image_size // self.patch_size

    @property
    def tokens(self) -> int:
        return self.grid * self.grid + 1


@dataclass
class TextDecoderCfg:
    llama_config_path: str = osp.join(osp.dirname(__file__), "llama_config")
    instruct: bool = False


@dataclass
class LlamaCfg:
    """Decoder shape; defaults = reference ``conf/llama_config/config.json`` (LLaMA-33B)."""
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
        if osp.isdir(path):
            path = osp.join(path, "config.json")
        with open(path) as f:
            d = json.load(f)
        keys = {k: d[k] for k in ("hidden_size", "intermediate_size", "num_attention_heads",
                                  "num_hidden_layers", "vocab_size", "rms_norm_eps",
                                  "max_position_embeddings") if k in d}
        if "rope_theta" in d:
            keys["rope_theta"] = d["rope_theta"]
        return cls(**keys)
