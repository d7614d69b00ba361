"""Emu1 caption path (BASELINE.json configs[0]) on the HIP engines: ``Emu.generate`` of the reference's first-generation
model (Emu1/models/modeling_emu.py:100-185) -- EVA-CLIP-g ViT (pre-norm blocks) -> ``ln_visual`` -> CausalFormer (T5-base
decoder over 32 learned queries, Emu1/models/causal_former.py:15-62) -> scatter at the ``<image>`` slots -> LLaMA-13B
``generate`` (beam search, 5 beams, length_penalty 0 by default).

The ViT and the decoder reuse the Emu2 engines (``emu_vit_*`` with ``prenorm=1``, ``emu_llama_*``); the CausalFormer is 12
tiny blocks on 32 tokens, walked by the host over the primitive operators (GEMM, RMS norm, flash cross-attention with
scale 1, materialised 32x32 self-attention with the additive relative-position bias through ``emu_softmax_rows_bf16``).
State-dict keys are the reference's (``visual.*``, ``ln_visual.*``, ``cformer.*``, ``decoder.lm.*``).  LoRA-wrapped
instruct checkpoints must be merged beforehand.  PARITY UNPINNED (see oracle/emu1_ref.py).
"""
from __future__ import annotations

import math
from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, Iterable, List, Optional, Tuple

import torch

from . import ops
from .conf.emu_conf import CLIPVisionCfg, LlamaCfg
from .constants import EOS_TOKEN_ID, IMAGE_TOKEN_ID, PAD_TOKEN_ID
from .llama import EmuHipContext, LlamaEngine
from .vit import VitEngine

BF16 = torch.bfloat16


@dataclass
class T5DecoderCfg:
    """t5-base constants used by CausalFormer (causal_former.py:25-27)."""
    d_model: int = 768
    num_layers: int = 12
    num_heads: int = 12
    d_kv: int = 64
    d_ff: int = 3072
    relative_attention_num_buckets: int = 32
    relative_attention_max_distance: int = 128
    layer_norm_epsilon: float = 1e-6
    n_causal: int = 32


def emu1_vision_cfg(**kw) -> CLIPVisionCfg:
    """Emu1/models/Emu-14B.json vision tower: eva-clip-g-14-x, pre-norm."""
    base = dict(eva_model_name="eva-clip-g-14-x", image_size=224, patch_size=14, width=1408, layers=40, head_width=88,
                mlp_ratio=4.3637, postnorm=False, n_query=32, v_query=32)
    base.update(kw)
    return CLIPVisionCfg(**base)


def emu1_llama_cfg(**kw) -> LlamaCfg:
    base = dict(hidden_size=5120, intermediate_size=13824, num_attention_heads=40, num_hidden_layers=40)
    base.update(kw)
    return LlamaCfg(**base)


def cformer_param_shapes(t5: T5DecoderCfg, vision_width: int, out_dim: int) -> "OrderedDict[str, Tuple[int, ...]]":
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    d, inner = t5.d_model, t5.num_heads * t5.d_kv
    s["cformer.causal_tokens"] = (1, t5.n_causal, d)
    for i in range(t5.num_layers):
        b = f"cformer.cformer.block.{i}.layer."
        for n in ("q", "k", "v"):
            s[b + f"0.SelfAttention.{n}.weight"] = (inner, d)
        s[b + "0.SelfAttention.o.weight"] = (d, inner)
        if i == 0:
            s[b + "0.SelfAttention.relative_attention_bias.weight"] = (t5.relative_attention_num_buckets, t5.num_heads)
        s[b + "0.layer_norm.weight"] = (d,)
        s[b + "1.EncDecAttention.q.weight"] = (inner, d)
        s[b + "1.EncDecAttention.k.weight"] = (inner, vision_width)
        s[b + "1.EncDecAttention.v.weight"] = (inner, vision_width)
        s[b + "1.EncDecAttention.o.weight"] = (d, inner)
        s[b + "1.layer_norm.weight"] = (d,)
        s[b + "2.DenseReluDense.wi.weight"] = (t5.d_ff, d)
        s[b + "2.DenseReluDense.wo.weight"] = (d, t5.d_ff)
        s[b + "2.layer_norm.weight"] = (d,)
    s["cformer.cformer.final_layer_norm.weight"] = (d,)
    s["cformer.projection.weight"] = (out_dim, d)
    s["cformer.projection.bias"] = (out_dim,)
    return s


def emu1_param_shapes(v: CLIPVisionCfg, t5: T5DecoderCfg, l: LlamaCfg, vocab: int):
    from .synth import llama_param_shapes, vit_param_shapes
    s = vit_param_shapes(v)
    s["ln_visual.weight"] = (v.width,)
    s["ln_visual.bias"] = (v.width,)
    s.update(cformer_param_shapes(t5, v.width, l.hidden_size))
    s.update(llama_param_shapes(l, vocab))
    return s


def _relative_position_bucket(rel: torch.Tensor, num_buckets: int, max_distance: int) -> torch.Tensor:
    """decoder (unidirectional) bucketing, modeling_t5.py:456-510 -- host-side table construction."""
    rel = -torch.min(rel, torch.zeros_like(rel))
    max_exact = num_buckets // 2
    large = max_exact + (torch.log(rel.float() / max_exact) / math.log(max_distance / max_exact)
                         * (num_buckets - max_exact)).to(torch.long)
    large = torch.min(large, torch.full_like(large, num_buckets - 1))
    return torch.where(rel < max_exact, rel, large)


class CausalFormer:
    def __init__(self, t5: T5DecoderCfg, vision_width: int, out_dim: int, ctx: EmuHipContext):
        self.cfg, self.ctx, self.device = t5, ctx, ctx.device
        self.vision_width, self.out_dim = vision_width, out_dim
        self.w: Dict[str, torch.Tensor] = {}
        self.want = cformer_param_shapes(t5, vision_width, out_dim)
        self.bias = None

    def load_tensor(self, name: str, t: torch.Tensor) -> bool:
        if name not in self.want:
            return False
        self.w[name] = t.to(self.device, BF16).contiguous()
        return True

    @property
    def ready(self) -> bool:
        return len(self.w) == len(self.want)

    def _self_bias(self) -> torch.Tensor:
        """[heads, n, n] bf16: relative-position bias of block 0 (shared by all blocks, modeling_t5.py:1319) + causal mask."""
        if self.bias is None:
            c, n = self.cfg, self.cfg.n_causal
            ctxp = torch.arange(n)[:, None]
            mem = torch.arange(n)[None, :]
            bucket = _relative_position_bucket(mem - ctxp, c.relative_attention_num_buckets, c.relative_attention_max_distance)
            table = self.w["cformer.cformer.block.0.layer.0.SelfAttention.relative_attention_bias.weight"]
            bias = table[bucket.to(self.device)].permute(2, 0, 1)
            causal = torch.where(mem <= ctxp, 0.0, torch.finfo(BF16).min).to(BF16).to(self.device)
            self.bias = (bias + causal[None]).contiguous()
        return self.bias

    @torch.no_grad()
    def forward(self, img_feats: torch.Tensor) -> torch.Tensor:
        """img_feats [B, T, vision_width] bf16 -> [B, n_causal, out_dim]."""
        c, w = self.cfg, self.w
        B, T, _ = img_feats.shape
        n, H, D, d = c.n_causal, c.num_heads, c.d_kv, c.d_model
        inner = H * D
        feats = img_feats.reshape(B * T, -1).contiguous()
        x = w["cformer.causal_tokens"].expand(B, -1, -1).reshape(B * n, d).contiguous()
        bias = self._self_bias()
        for i in range(c.num_layers):
            b = f"cformer.cformer.block.{i}.layer."
            # causal self-attention with additive position bias (32 x 32 per head: materialised)
            h = ops.rmsnorm(x, w[b + "0.layer_norm.weight"], c.layer_norm_epsilon)
            q = ops.linear(h, w[b + "0.SelfAttention.q.weight"]).view(B, n, H, D)
            k = ops.linear(h, w[b + "0.SelfAttention.k.weight"]).view(B, n, H, D)
            v = ops.linear(h, w[b + "0.SelfAttention.v.weight"]).view(B, n, H, D)
            att = torch.empty(B, n, H, D, device=self.device, dtype=BF16)
            for bi in range(B):
                for hh in range(H):
                    s = ops.linear(q[bi, :, hh, :], k[bi, :, hh, :])                       # [n, n], no 1/sqrt(d)
                    ops.softmax_rows_(s, 1.0, bias=bias[hh])
                    att[bi, :, hh, :] = ops.linear(s, v[bi, :, hh, :].t().contiguous())
            x = ops.linear(att.view(B * n, inner), w[b + "0.SelfAttention.o.weight"], res=x, epi=ops.EPI_RESID)
            # cross-attention over the image features (k/v projections read the 1408-wide features), scale 1
            h = ops.rmsnorm(x, w[b + "1.layer_norm.weight"], c.layer_norm_epsilon)
            q = ops.linear(h, w[b + "1.EncDecAttention.q.weight"]).view(B, n, H, D)
            k = ops.linear(feats, w[b + "1.EncDecAttention.k.weight"]).view(B, T, H, D)
            v = ops.linear(feats, w[b + "1.EncDecAttention.v.weight"]).view(B, T, H, D)
            att = ops.flash_attn(q, k, v, causal=False, scale=1.0)
            x = ops.linear(att.view(B * n, inner), w[b + "1.EncDecAttention.o.weight"], res=x, epi=ops.EPI_RESID)
            # ReLU feed-forward
            h = ops.rmsnorm(x, w[b + "2.layer_norm.weight"], c.layer_norm_epsilon)
            f = torch.relu_(ops.linear(h, w[b + "2.DenseReluDense.wi.weight"]))
            x = ops.linear(f, w[b + "2.DenseReluDense.wo.weight"], res=x, epi=ops.EPI_RESID)
        x = ops.rmsnorm(x, w["cformer.cformer.final_layer_norm.weight"], c.layer_norm_epsilon)
        out = ops.linear(x, w["cformer.projection.weight"], bias=w["cformer.projection.bias"])
        return out.view(B, n, self.out_dim)


def merge_lora_state_dict(sd, r: int = 16, alpha: float = 16.0):
    """Fold peft LoRA adapters into their base weights: W' = W + (alpha / r) * B @ A (fp32 math, caller rounds).

    Handles both peft layouts: base weight under ``<mod>.weight`` or ``<mod>.base_layer.weight``; adapters under
    ``<mod>.lora_A[.default].weight`` / ``<mod>.lora_B[.default].weight``; the ``base_model.model.`` infix that
    ``get_peft_model`` inserts after ``decoder.lm.`` is dropped.  Dicts without adapters pass through unchanged."""
    import re
    scale = float(alpha) / float(r)
    out, lora = {}, {}
    for name, t in sd.items():
        name = name.replace("decoder.lm.base_model.model.", "decoder.lm.")
        m = re.match(r"^(.*)\.lora_([AB])(?:\.[A-Za-z0-9_]+)?\.weight$", name)
        if m:
            lora.setdefault(m.group(1), {})[m.group(2)] = t
            continue
        if ".lora_dropout" in name or name.endswith(".lora_embedding_A") or name.endswith(".lora_embedding_B"):
            continue
        out[name.replace(".base_layer.weight", ".weight").replace(".base_layer.bias", ".bias")] = t
    for mod, ab in lora.items():
        if "A" not in ab or "B" not in ab:
            raise RuntimeError(f"LoRA adapter of {mod} is incomplete")
        key = mod + ".weight"
        if key not in out:
            raise RuntimeError(f"LoRA adapter for {mod} has no base weight")
        if ab["A"].shape[0] != ab["B"].shape[1]:
            raise RuntimeError(f"LoRA rank mismatch at {mod}: A {tuple(ab['A'].shape)}, B {tuple(ab['B'].shape)}")
        base = out[key]
        merged = base.float() + scale * (ab["B"].float() @ ab["A"].float())
        out[key] = merged.to(base.dtype)
    return out


class Emu:
    """Drop-in for the reference's first-generation ``Emu`` model (inference: ``generate``)."""

    def __init__(self, vision_cfg: Optional[CLIPVisionCfg] = None, llama_cfg: Optional[LlamaCfg] = None,
                 t5_cfg: Optional[T5DecoderCfg] = None, vocab: int = 32006, device="cuda", instruct: bool = False,
                 ctx: Optional[EmuHipContext] = None):
        self.vision_cfg = vision_cfg or emu1_vision_cfg()
        self.llama_cfg = llama_cfg or emu1_llama_cfg()
        self.t5_cfg = t5_cfg or T5DecoderCfg()
        dev = torch.device(device)
        if dev.type == "cuda" and dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        self.ctx = ctx or EmuHipContext(dev)
        self.vocab = vocab                                    # 32000 + [PAD],[IMG],[/IMG],<image> (+[USER],[ASSISTANT])
        self.visual = VitEngine(self.vision_cfg, self.ctx)
        self.cformer = CausalFormer(self.t5_cfg, self.vision_cfg.width, self.llama_cfg.hidden_size, self.ctx)
        self.lm = LlamaEngine(self.llama_cfg, vocab, self.ctx)
        # Emu1/requirements.txt:2 leaves transformers unpinned: the beam-search conventions of the current library
        self.hf_semantics = "5.x"
        self.ln_w = self.ln_b = None
        self.n_causal = self.t5_cfg.n_causal
        self.image_placeholder = "[IMG]" + "<image>" * self.n_causal + "[/IMG]"
        self.tokenizer = None

    def load_state_dict(self, sd, strict: bool = True, lora_r: int = 16, lora_alpha: float = 16.0):
        """Reference key names (Emu1/models/modeling_emu.py).  The instruct checkpoint is saved with peft LoRA adapters
        on the attention projections (Emu1/inference.py:40-51: r = 16, alpha = 16 on q/k/v/o_proj); they are merged
        into the base matrices here, so the engines only ever see plain LLaMA weights."""
        sd = merge_lora_state_dict(dict(sd.items() if hasattr(sd, "items") else sd), lora_r, lora_alpha)
        unexpected = []
        for name, t in sd.items():
            if name.startswith("visual."):
                used = self.visual.load_tensor(name[len("visual."):], t)
            elif name.startswith("decoder.lm."):
                used = self.lm.load_tensor(name[len("decoder.lm."):], t) or name.endswith("stu_regress_head.weight")
            elif name.startswith("cformer."):
                used = self.cformer.load_tensor(name, t)
            elif name == "ln_visual.weight":
                self.ln_w, used = t.to(self.ctx.device, BF16).contiguous(), True
            elif name == "ln_visual.bias":
                self.ln_b, used = t.to(self.ctx.device, BF16).contiguous(), True
            else:
                used = False
            if not used and not name.endswith("rotary_emb.inv_freq"):
                unexpected.append(name)
        missing = [n for n, ok in (("visual.*", self.visual.ready), ("cformer.*", self.cformer.ready),
                                   ("decoder.lm.*", self.lm.ready), ("ln_visual.*", self.ln_w is not None and self.ln_b is not None))
                   if not ok]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict for Emu: missing {missing}, unexpected {unexpected[:5]}")
        return missing, unexpected

    @torch.no_grad()
    def encode_image(self, image: torch.Tensor) -> torch.Tensor:
        """modeling_emu.py:124-126: image (cast to bf16) -> ViT-g -> ln_visual -> CausalFormer: [B, 32, hidden]."""
        f = self.visual(image.to(BF16))
        B, T, C = f.shape
        f = ops.layernorm(f.view(B * T, C), self.ln_w, self.ln_b, 1e-6).view(B, T, C)
        return self.cformer.forward(f)

    @torch.no_grad()
    def generate_ids(self, input_ids: torch.Tensor, attention_mask: torch.Tensor, image: Optional[torch.Tensor] = None,
                     num_beams: int = 5, max_new_tokens: int = 50, min_length: int = 1, length_penalty: float = 0.0,
                     do_sample: bool = False, temperature=None, top_k=None, top_p=None, repetition_penalty: float = 1.0,
                     penalty_alpha=None, no_repeat_ngram_size=None, num_return_sequences: int = 1,
                     hf_semantics: Optional[str] = None):
        B, S = input_ids.shape
        x = self.lm.embed_tokens(input_ids).view(B * S, -1)
        if image is not None:
            e = self.encode_image(image)
            e = e.reshape(-1, e.shape[-1]).contiguous()
            rows = torch.nonzero(input_ids.reshape(-1).to(self.ctx.device) == IMAGE_TOKEN_ID).reshape(-1).to(torch.int32)
            if rows.numel() != e.shape[0]:
                raise ValueError(f"shape mismatch: {rows.numel()} <image> slots cannot take {e.shape[0]} visual tokens")
            ops.scatter_rows(e, rows.contiguous(), x)
        x = x.view(B, S, -1)
        # mode selection as transformers does it (modeling_emu.py:162-179 forwards every argument to lm.generate)
        ngram, nret = int(no_repeat_ngram_size or 0), int(num_return_sequences)
        if penalty_alpha is not None and penalty_alpha > 0 and top_k is not None and top_k > 1 and num_beams == 1 and not do_sample:
            if ngram or nret != 1:
                raise NotImplementedError("contrastive search with no_repeat_ngram_size / several returned sequences is not built")
            return self.lm.contrastive_generate(x, attention_mask, max_new_tokens, float(penalty_alpha), int(top_k), min_length,
                                                repetition_penalty, eos_id=EOS_TOKEN_ID, pad_id=PAD_TOKEN_ID)
        if num_beams > 1:
            return self.lm.beam_search_generate(x, attention_mask, num_beams, max_new_tokens, min_length, length_penalty,
                                                eos_id=EOS_TOKEN_ID, pad_id=PAD_TOKEN_ID, do_sample=do_sample,
                                                temperature=temperature, top_k=top_k, top_p=top_p,
                                                repetition_penalty=repetition_penalty, no_repeat_ngram_size=ngram,
                                                num_return_sequences=nret,
                                                hf_semantics=hf_semantics or getattr(self, "hf_semantics", "5.x"))
        if do_sample or repetition_penalty != 1.0 or ngram or nret != 1:
            return self.lm.sample_generate(x, attention_mask, max_new_tokens, min_length, do_sample, temperature, top_k, top_p,
                                           repetition_penalty, eos_id=EOS_TOKEN_ID, pad_id=PAD_TOKEN_ID,
                                           no_repeat_ngram_size=ngram, num_return_sequences=nret)
        return self.lm.greedy_generate(x, attention_mask, max_new_tokens, min_length, eos_id=EOS_TOKEN_ID, pad_id=PAD_TOKEN_ID)

    @torch.no_grad()
    def generate(self, samples, do_sample=False, num_beams=5, max_new_tokens=50, min_length=1, top_p=0.9,
                 repetition_penalty=1.0, length_penalty=0.0, num_captions=1, temperature=1, penalty_alpha=None, top_k=None,
                 no_repeat_ngram_size=None, **kwargs) -> List[str]:
        if kwargs:
            raise TypeError(f"Emu.generate: unsupported generation options {sorted(kwargs)}")
        if self.tokenizer is None:
            raise RuntimeError("set Emu.tokenizer (LlamaTokenizer + [PAD],[IMG],[/IMG],<image>) to use the text API")
        tok = self.tokenizer
        tok.padding_side = "left"
        enc = tok(samples["prompt"], padding="longest", return_tensors="pt", add_special_tokens=True)
        tok.padding_side = "right"
        ids = self.generate_ids(enc.input_ids, enc.attention_mask, samples.get("image"), num_beams, max_new_tokens,
                                min_length, length_penalty, do_sample=do_sample, temperature=temperature, top_k=top_k,
                                top_p=top_p, repetition_penalty=repetition_penalty, penalty_alpha=penalty_alpha,
                                no_repeat_ngram_size=no_repeat_ngram_size, num_return_sequences=num_captions)
        return tok.batch_decode(ids.cpu(), skip_special_tokens=True)
