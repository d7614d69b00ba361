"""CPU restatement of the Emu2 inference hot path (TEST INFRASTRUCTURE ONLY).

Pure torch-on-CPU functional code over a flat ``{state_dict_key: tensor}`` mapping
that uses the reference's own parameter names, so the same weights drive the
reference (when importable), this oracle, and the HIP product path.

Every function cites the reference lines it restates (paths relative to
``/root/reference``).  LLaMA arithmetic lives in the third-party ``transformers``
package (pinned ==4.31.0 by ``Emu2/requirements.txt:2``; 5.15.0 installed here);
its published algorithm is restated and pinned through the reference's own call
sites ``Emu2/emu/emu.py:119,133-138,193,213-229`` (see tests/test_oracle_golden.py).

Pinned against the real reference by ``oracle/make_golden.py`` ->
``tests/golden/*.npz`` (checked by ``tests/test_oracle_golden.py``).

``dtype`` semantics: every function computes in the dtype of the tensors it is
handed.  float32 weights/inputs give the fp32 oracle; bfloat16 gives the same
rounding points as the reference run in bf16 on CPU.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Weights = Dict[str, Tensor]


# --------------------------------------------------------------------------- config
@dataclass
class VitCfg:
    """Mirror of CLIPVisionCfg, Emu2/emu/conf/emu_conf.py:6-33."""
    image_size: int = 448
    patch_size: int = 14
    width: int = 1792
    layers: int = 64
    head_width: int = 112
    mlp_hidden: int = 15360          # int(width * mlp_ratio), eva_vit.py:270
    ln_eps: float = 1e-6             # Emu2/emu/emu.py:37

    @property
    def heads(self) -> int:
        return self.width // self.head_width

    @property
    def grid(self) -> int:
        return self.image_size // self.patch_size

    @property
    def tokens(self) -> int:
        return self.grid * self.grid + 1


@dataclass
class LlamaCfg:
    """Mirror of Emu2/emu/conf/llama_config/config.json."""
    hidden: int = 6656
    heads: int = 52
    layers: int = 60
    ffn: int = 17920
    vocab: int = 32274
    rms_eps: float = 1e-6
    rope_theta: float = 10000.0
    max_pos: int = 2048

    @property
    def head_dim(self) -> int:
        return self.hidden // self.heads


@dataclass
class EmuCfg:
    vit: VitCfg = field(default_factory=VitCfg)
    llama: LlamaCfg = field(default_factory=LlamaCfg)
    n_query: int = 64
    v_query: int = 64


# token ids after the tokenizer extension of Emu2/emu/lm.py:43-63
PAD_ID, BOS_ID, EOS_ID = 32000, 1, 2
BOI_ID, EOI_ID, IMAGE_ID, GIMG_ID = 32001, 32002, 32003, 32004


# --------------------------------------------------------------------------- ViT
def vit_patch_embed(image: Tensor, W: Weights) -> Tensor:
    """PatchEmbed.forward, Emu2/emu/eva_vit.py:329-335 (conv def :327): stride-p conv
    == per-patch GEMM; output [B, gh*gw, C] in row-major patch order."""
    w = W["visual.patch_embed.proj.weight"]
    b = W["visual.patch_embed.proj.bias"]
    p = w.shape[-1]
    x = F.conv2d(image, w, b, stride=p)
    return x.flatten(2).transpose(1, 2)


def vit_attention(x: Tensor, W: Weights, i: int, cfg: VitCfg) -> Tensor:
    """Attention.forward naive branch, Emu2/emu/eva_vit.py:182-252: fused qkv with bias
    cat(q_bias, 0, v_bias) (:194-198); q*scale BEFORE q@k^T (:227-228); softmax in the
    tensor dtype (:245); proj with bias (:250)."""
    pre = f"visual.blocks.{i}.attn."
    B, N, C = x.shape
    qb, vb = W[pre + "q_bias"], W[pre + "v_bias"]
    bias = torch.cat((qb, torch.zeros_like(vb), vb))
    qkv = F.linear(x, W[pre + "qkv.weight"], bias)
    qkv = qkv.reshape(B, N, 3, cfg.heads, -1).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    q = q * (cfg.head_width ** -0.5)
    attn = (q @ k.transpose(-2, -1)).softmax(dim=-1)
    out = (attn @ v).transpose(1, 2).reshape(B, N, -1)
    return F.linear(out, W[pre + "proj.weight"], W[pre + "proj.bias"])


def vit_mlp(x: Tensor, W: Weights, i: int) -> Tensor:
    """Mlp.forward, Emu2/emu/eva_vit.py:105-114: fc1 -> GELU(erf) -> fc2."""
    pre = f"visual.blocks.{i}.mlp."
    h = F.linear(x, W[pre + "fc1.weight"], W[pre + "fc1.bias"])
    h = F.gelu(h)
    return F.linear(h, W[pre + "fc2.weight"], W[pre + "fc2.bias"])


def vit_block(x: Tensor, W: Weights, i: int, cfg: VitCfg) -> Tensor:
    """Block.forward post-norm branch, Emu2/emu/eva_vit.py:296-300:
    x = x + LN1(attn(x)); x = x + LN2(mlp(x)); LayerNorm eps 1e-6 (emu.py:37)."""
    pre = f"visual.blocks.{i}."
    C = x.shape[-1]
    a = vit_attention(x, W, i, cfg)
    x = x + F.layer_norm(a, (C,), W[pre + "norm1.weight"], W[pre + "norm1.bias"], cfg.ln_eps)
    m = vit_mlp(x, W, i)
    x = x + F.layer_norm(m, (C,), W[pre + "norm2.weight"], W[pre + "norm2.bias"], cfg.ln_eps)
    return x


def vit_forward(image: Tensor, W: Weights, cfg: VitCfg) -> Tensor:
    """EVAVisionTransformer.forward_features, Emu2/emu/eva_vit.py:402-431: cat cls,
    + pos_embed, blocks; no final norm (forward :433-445 returns raw features)."""
    x = vit_patch_embed(image, W)
    cls = W["visual.cls_token"].expand(x.shape[0], -1, -1)
    x = torch.cat((cls, x), dim=1) + W["visual.pos_embed"]
    for i in range(cfg.layers):
        x = vit_block(x, W, i, cfg)
    return x


def encode_image(image: Tensor, W: Weights, cfg: EmuCfg, n_query: Optional[int] = None) -> Tensor:
    """EmuModel.encode_image, Emu2/emu/emu.py:77-90: drop cls, [B,C,g,g], avg_pool2d with
    kernel = stride = g // sqrt(n_query), flatten row-major -> [B, n_query, C]."""
    n_query = cfg.n_query if n_query is None else n_query
    x = vit_forward(image, W, cfg.vit)[:, 1:, :]
    b, n, c = x.shape
    g = int(n ** 0.5)
    x = x.permute(0, 2, 1).reshape(b, c, g, g)
    stride = int(g // (n_query ** 0.5))
    x = F.avg_pool2d(x, kernel_size=(stride, stride), stride=stride)
    return x.reshape(b, c, -1).permute(0, 2, 1).contiguous()


# --------------------------------------------------------------------------- LLaMA
def rms_norm(x: Tensor, w: Tensor, eps: float) -> Tensor:
    """transformers LlamaRMSNorm: fp32 variance, cast back, THEN multiply by weight."""
    dt = x.dtype
    h = x.to(torch.float32)
    h = h * torch.rsqrt(h.pow(2).mean(-1, keepdim=True) + eps)
    return w * h.to(dt)


def rope_cos_sin(position_ids: Tensor, head_dim: int, theta: float, dtype) -> Tuple[Tensor, Tensor]:
    """transformers LlamaRotaryEmbedding.forward (default rope): inv_freq fp32,
    emb = cat(freqs, freqs), cos/sin computed in fp32 then cast to the model dtype."""
    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    freqs = position_ids.to(torch.float32)[..., None] * inv          # [B,S,D/2]
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos().to(dtype), emb.sin().to(dtype)


def rotate_half(x: Tensor) -> Tensor:
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def apply_rope(q: Tensor, k: Tensor, cos: Tensor, sin: Tensor) -> Tuple[Tensor, Tensor]:
    """transformers apply_rotary_pos_emb: (x*cos) + (rotate_half(x)*sin), q/k [B,H,S,D]."""
    cos, sin = cos[:, None], sin[:, None]
    return q * cos + rotate_half(q) * sin, k * cos + rotate_half(k) * sin


class KVCache:
    """Per-layer list of (k, v) [B,H,S,D] grown by concatenation (HF DynamicCache semantics)."""

    def __init__(self, layers: int):
        self.k: List[Optional[Tensor]] = [None] * layers
        self.v: List[Optional[Tensor]] = [None] * layers

    def append(self, i: int, k: Tensor, v: Tensor) -> Tuple[Tensor, Tensor]:
        if self.k[i] is None:
            self.k[i], self.v[i] = k, v
        else:
            self.k[i] = torch.cat((self.k[i], k), dim=2)
            self.v[i] = torch.cat((self.v[i], v), dim=2)
        return self.k[i], self.v[i]

    def reorder(self, idx: Tensor) -> None:
        for i in range(len(self.k)):
            if self.k[i] is not None:
                self.k[i] = self.k[i].index_select(0, idx)
                self.v[i] = self.v[i].index_select(0, idx)


def llama_layer(x: Tensor, W: Weights, i: int, cfg: LlamaCfg, cos: Tensor, sin: Tensor,
                mask: Tensor, cache: Optional[KVCache]) -> Tensor:
    """transformers LlamaDecoderLayer (eager attention): RMSNorm -> q/k/v -> RoPE ->
    softmax(QK^T/sqrt(d) + mask) in fp32 -> PV -> o_proj -> +res -> RMSNorm -> SwiGLU -> +res.
    Reached from Emu2/emu/emu.py:133-138 (lm.model) and :213-229 (lm.generate)."""
    pre = f"decoder.lm.model.layers.{i}."
    B, S, _ = x.shape
    H, D = cfg.heads, cfg.head_dim
    h = rms_norm(x, W[pre + "input_layernorm.weight"], cfg.rms_eps)
    q = F.linear(h, W[pre + "self_attn.q_proj.weight"]).view(B, S, H, D).transpose(1, 2)
    k = F.linear(h, W[pre + "self_attn.k_proj.weight"]).view(B, S, H, D).transpose(1, 2)
    v = F.linear(h, W[pre + "self_attn.v_proj.weight"]).view(B, S, H, D).transpose(1, 2)
    q, k = apply_rope(q, k, cos, sin)
    if cache is not None:
        k, v = cache.append(i, k, v)
    scores = (q @ k.transpose(2, 3)) * (D ** -0.5) + mask
    p = torch.softmax(scores, dim=-1, dtype=torch.float32).to(q.dtype)
    a = (p @ v).transpose(1, 2).reshape(B, S, H * D)
    x = x + F.linear(a, W[pre + "self_attn.o_proj.weight"])
    h = rms_norm(x, W[pre + "post_attention_layernorm.weight"], cfg.rms_eps)
    g = F.linear(h, W[pre + "mlp.gate_proj.weight"])
    u = F.linear(h, W[pre + "mlp.up_proj.weight"])
    x = x + F.linear(F.silu(g) * u, W[pre + "mlp.down_proj.weight"])
    return x


def build_mask(attn_mask_full: Tensor, q_len: int, dtype) -> Tensor:
    """Additive mask [B,1,q_len,kv_len]: causal over absolute slots + key padding
    (attention_mask==0 keys masked), as transformers' create_causal_mask does."""
    B, kv = attn_mask_full.shape
    past = kv - q_len
    qi = torch.arange(past, kv)[:, None]
    kj = torch.arange(kv)[None, :]
    allowed = (kj <= qi)[None, None] & attn_mask_full.bool()[:, None, None, :]
    m = torch.zeros(B, 1, q_len, kv, dtype=dtype)
    return m.masked_fill(~allowed, torch.finfo(dtype).min)


def llama_model(embeds: Tensor, attention_mask: Tensor, W: Weights, cfg: LlamaCfg,
                position_ids: Optional[Tensor] = None, cache: Optional[KVCache] = None,
                final_norm: bool = True) -> Tensor:
    """transformers LlamaModel.forward on ``inputs_embeds``.  ``attention_mask`` covers
    past + current slots.  ``position_ids`` default = arange(past, past+S), which is what
    ``lm.model(inputs_embeds=..., attention_mask=...)`` uses (Emu2/emu/emu.py:133-138)."""
    B, S, _ = embeds.shape
    past = attention_mask.shape[1] - S
    if position_ids is None:
        position_ids = torch.arange(past, past + S)[None].expand(B, -1)
    cos, sin = rope_cos_sin(position_ids, cfg.head_dim, cfg.rope_theta, embeds.dtype)
    mask = build_mask(attention_mask, S, embeds.dtype)
    x = embeds
    for i in range(cfg.layers):
        x = llama_layer(x, W, i, cfg, cos, sin, mask, cache)
    if final_norm:
        x = rms_norm(x, W["decoder.lm.model.norm.weight"], cfg.rms_eps)
    return x


def embed_tokens(ids: Tensor, W: Weights) -> Tensor:
    """Emu2/emu/emu.py:119,193."""
    return F.embedding(ids, W["decoder.lm.model.embed_tokens.weight"])


def greedy_generate(embeds: Tensor, attention_mask: Tensor, W: Weights, cfg: LlamaCfg,
                    max_new_tokens: int, min_len: int = 1,
                    return_margins: bool = False, eos_id: Optional[int] = None):
    """``lm.generate(inputs_embeds=..., num_beams=1, do_sample=False)`` as called at
    Emu2/emu/emu.py:213-229: returns ONLY the new ids [B, <=max_new_tokens];
    position_ids = cumsum(attention_mask)-1 (left padding); EOS suppressed while fewer
    than ``min_len`` tokens exist (MinLengthLogitsProcessor; with inputs_embeds the id
    sequence starts empty; ``min_len`` here is the EFFECTIVE minimum, see ``effective_min_len``); finished rows emit PAD;
    stops when all rows finished."""
    B, S, _ = embeds.shape
    EOS = EOS_ID if eos_id is None else int(eos_id)
    cache = KVCache(cfg.layers)
    mask = attention_mask.clone()
    pos = (mask.long().cumsum(-1) - 1).masked_fill(mask == 0, 1)
    x = embeds
    out: List[Tensor] = []
    margins: List[Tensor] = []
    unfinished = torch.ones(B, dtype=torch.long)
    for step in range(max_new_tokens):
        h = llama_model(x, mask, W, cfg, position_ids=pos, cache=cache)
        logits = F.linear(h[:, -1, :], W["decoder.lm.lm_head.weight"]).to(torch.float32)
        if step < min_len:
            logits[:, EOS] = -float("inf")
        top2 = logits.topk(2, dim=-1).values
        margins.append(top2[:, 0] - top2[:, 1])
        nxt = logits.argmax(dim=-1)
        nxt = nxt * unfinished + PAD_ID * (1 - unfinished)
        out.append(nxt)
        unfinished = unfinished * (nxt != EOS).long()
        if unfinished.max() == 0:
            break
        x = embed_tokens(nxt[:, None], W)
        mask = torch.cat((mask, torch.ones(B, 1, dtype=mask.dtype)), dim=1)
        pos = pos[:, -1:] + 1
    ids = torch.stack(out, dim=1)
    if return_margins:
        return ids, torch.stack(margins, dim=1)
    return ids


def beam_search_generate(embeds: Tensor, attention_mask: Tensor, W: Weights, cfg: LlamaCfg, num_beams: int,
                         max_new_tokens: int, min_len: int = 1, length_penalty: float = -1.0,
                         return_margin: bool = False, eos_id: Optional[int] = None):
    """``lm.generate(inputs_embeds=..., num_beams=N, do_sample=False)`` with the reference's defaults
    (Emu2/emu/emu.py:163-172,213-229: num_beams=5, length_penalty=-1, early_stopping unset=False): restatement of
    transformers' beam search -- 2N best continuations of (beam, token) per step; the N best unfinished ones keep
    running; finished ones (EOS or the length limit) enter the N result slots scored log-prob / len**length_penalty;
    stop when the best running beam cannot beat the worst kept result (heuristic at the current length)."""
    B, S, _ = embeds.shape
    nb, NEG = num_beams, -1.0e9
    EOS = EOS_ID if eos_id is None else int(eos_id)        # (``eos_token_id=`` forwarded through EmuModel.generate's **kwargs)
    x = embeds.repeat_interleave(nb, dim=0)
    mask = attention_mask.repeat_interleave(nb, dim=0).clone()
    pos = (mask.long().cumsum(-1) - 1).masked_fill(mask == 0, 1)
    cache = KVCache(cfg.layers)
    V = W["decoder.lm.lm_head.weight"].shape[0]
    run_seq = torch.full((B, nb, max_new_tokens), PAD_ID, dtype=torch.long)
    seqs = run_seq.clone()
    run_sc = torch.zeros(B, nb); run_sc[:, 1:] = NEG
    fin_sc = torch.full((B, nb), NEG)
    finished = torch.zeros(B, nb, dtype=torch.bool)
    lens = torch.zeros(B, nb, dtype=torch.long)
    open_ = torch.ones(B, 1, dtype=torch.bool)
    top_mask = torch.cat([torch.ones(nb, dtype=torch.bool), torch.zeros(nb, dtype=torch.bool)])
    g3 = lambda t, i: torch.gather(t, 1, i[:, :, None].expand(-1, -1, t.shape[2]))
    cur = 0
    margin = float("inf")          # smallest gap at a pruning boundary (N-th vs N+1-th running candidate) seen so far
    while True:
        h = llama_model(x, mask, W, cfg, position_ids=pos, cache=cache)
        logits = F.linear(h[:, -1, :], W["decoder.lm.lm_head.weight"]).to(torch.float32)
        lp = torch.log_softmax(logits, dim=-1)
        if cur < min_len:
            lp[:, EOS] = -float("inf")
        acc = (lp.view(B, nb, V) + run_sc[:, :, None]).reshape(B, nb * V)
        top_lp, top_i = torch.topk(acc, k=2 * nb)
        src, tok = top_i // V, top_i % V
        cand = g3(run_seq, src)
        cand[:, :, cur] = tok
        hits = (tok == EOS) | (cur + 1 >= max_new_tokens)
        r_lp = top_lp + hits.float() * NEG
        nxt = torch.topk(r_lp, k=nb)[1]
        if cur + 1 < max_new_tokens:
            srt = torch.sort(r_lp, dim=1, descending=True)[0]
            margin = min(margin, float((srt[:, nb - 1] - srt[:, nb]).min()))
        run_seq, run_sc, beam_idx = g3(cand, nxt), torch.gather(r_lp, 1, nxt), torch.gather(src, 1, nxt)
        f_lp = top_lp / float((cur + 1) ** length_penalty) + (~open_).float() * NEG
        just = hits & top_mask[None, :]
        f_lp = f_lp + (~just).float() * NEG
        m_seq, m_sc = torch.cat((seqs, cand), 1), torch.cat((fin_sc, f_lp), 1)
        m_fin = torch.cat((finished, just), 1)
        m_len = torch.cat((lens, torch.full((B, 2 * nb), cur + 1)), 1)
        keep = torch.topk(m_sc, k=nb)[1]
        seqs, fin_sc = g3(m_seq, keep), torch.gather(m_sc, 1, keep)
        finished, lens = torch.gather(m_fin, 1, keep), torch.gather(m_len, 1, keep)
        cur += 1
        best_run = run_sc[:, :1] / float(cur ** length_penalty)
        worst = torch.where(finished, fin_sc.min(dim=1, keepdim=True)[0], torch.full_like(fin_sc, NEG))
        open_ = open_ & (best_run > worst).any(dim=-1, keepdim=True)
        if not bool(open_.any()) or bool(hits.all()):
            break
        flat = (beam_idx + torch.arange(B)[:, None] * nb).reshape(-1)
        cache.reorder(flat)
        x = embed_tokens(run_seq[:, :, cur - 1].reshape(-1, 1), W)
        mask = torch.cat((mask, torch.ones(B * nb, 1, dtype=mask.dtype)), dim=1)
        pos = pos[:, -1:] + 1
    out = seqs[:, 0, : int(lens[:, 0].max())]
    if return_margin:
        # also the gap between the two best final results (the returned one must win clearly)
        final_gap = float((fin_sc[:, 0] - fin_sc[:, 1]).min())
        return out, min(margin, final_gap)
    return out


# --------------------------------------------------------------------------- EmuModel
def scatter_image_embeds(text_embeds: Tensor, input_ids: Tensor, image_embeds: Tensor,
                         token_id: int = IMAGE_ID) -> Tensor:
    """Emu2/emu/emu.py:202-203 (video :210-211): masked row overwrite in row-major
    (batch, position) order; slot count must equal the number of rows supplied."""
    idx = input_ids == token_id
    if int(idx.sum()) != image_embeds.shape[0]:
        raise ValueError("number of <image> slots != number of image embedding rows")
    out = text_embeds.clone()
    out[idx] = image_embeds.to(out.dtype)
    return out


def effective_min_len(min_len: int, prompt_len: int, hf_semantics: str = "5.x") -> int:
    """How many generated tokens ``min_length=min_len`` (Emu2/emu/emu.py:220) really enforces when ``lm.generate`` is driven by
    ``inputs_embeds``.  transformers 5.x (the installed library, the source of the golden fixtures) subtracts the prompt length:
    ``GenerationMixin._prepare_generated_length``: ``min_length = max(min_length - inputs_embeds.shape[1], 0)`` -- the reference's
    default min_len = 1 then enforces NOTHING, and EOS may be the first token (pinned by tests/golden/generate_beam_eos_tiny.npz).
    transformers 4.31 (the reference's pin; restated, not installable here) has no such correction: its MinLengthLogitsProcessor
    compares against the EMPTY id sequence, so min_len new tokens are enforced."""
    return int(min_len) if hf_semantics == "4.31" else max(int(min_len) - int(prompt_len), 0)


def emu_generate(input_ids: Tensor, attention_mask: Tensor, image: Optional[Tensor], W: Weights,
                 cfg: EmuCfg, max_new_tokens: int, min_len: int = 1, n_query: Optional[int] = None,
                 return_margins: bool = False, num_beams: int = 1, video: Optional[Tensor] = None,
                 eos_id: Optional[int] = None, hf_semantics: str = "5.x"):
    """EmuModel.generate at the token-id level (greedy or beam search), Emu2/emu/emu.py:184-229; video frames are
    encoded with v_query tokens each and land on the [gIMG] slots (:205-211)."""
    x = embed_tokens(input_ids, W)
    if image is not None:
        e = encode_image(image, W, cfg, n_query)
        e = F.linear(e.reshape(-1, e.shape[-1]), W["project_up.weight"])
        x = scatter_image_embeds(x, input_ids, e)
    if video is not None:
        e = encode_image(video, W, cfg, cfg.v_query)
        e = F.linear(e.reshape(-1, e.shape[-1]), W["project_up.weight"])
        x = scatter_image_embeds(x, input_ids, e, token_id=GIMG_ID)
    min_len = effective_min_len(min_len, input_ids.shape[1], hf_semantics)
    if num_beams > 1:
        return beam_search_generate(x, attention_mask, W, cfg.llama, num_beams, max_new_tokens, min_len,
                                    return_margin=return_margins, eos_id=eos_id)
    return greedy_generate(x, attention_mask, W, cfg.llama, max_new_tokens, min_len, return_margins, eos_id=eos_id)


def _suffix_count(flag: Tensor) -> Tensor:
    """torch.flip(cumsum(flip(flag))) of Emu2/emu/emu.py:124,141."""
    return torch.flip(torch.cumsum(torch.flip(flag.long(), dims=[1]), dim=1), dims=[1])


def emu_generate_image_uncached(prompt_ids: Tensor, image: Optional[Tensor], W: Weights,
                                cfg: EmuCfg) -> Tensor:
    """EmuModel.generate_image exactly as written (64 full forwards, NO KV cache),
    Emu2/emu/emu.py:92-153, at the token-id level for B rows of EQUAL length (no padding):
    iteration n feeds prompt + [IMG] + n x <image>."""
    B = prompt_ids.shape[0]
    prompt_image = None
    if image is not None:
        e = encode_image(image, W, cfg)
        prompt_image = F.linear(e.reshape(-1, e.shape[-1]), W["project_up.weight"])
    target = None
    ids = prompt_ids
    for n in range(cfg.n_query):
        tok = BOI_ID if n == 0 else IMAGE_ID
        ids = torch.cat((ids, torch.full((B, 1), tok, dtype=ids.dtype)), dim=1)
        x = embed_tokens(ids, W)
        is_img = ids == IMAGE_ID
        cnt = _suffix_count(is_img)
        if prompt_image is not None:
            x[is_img & (cnt > n)] = prompt_image.to(x.dtype)
        if target is not None:
            x[is_img & (cnt > 0) & (cnt <= n)] = F.linear(target, W["project_up.weight"])
        h = llama_model(x, torch.ones_like(ids), W, cfg.llama)
        slot = (ids == IMAGE_ID) | (ids == BOI_ID)
        cnt = _suffix_count(slot)
        sel = slot & (cnt > 0) & (cnt <= n + 1)
        target = F.linear(h[sel].reshape(-1, h.shape[-1]), W["project_down.weight"])
    return target.reshape(B, -1, target.shape[-1])


def emu_generate_image_cached(prompt_ids: Tensor, image: Optional[Tensor], W: Weights,
                              cfg: EmuCfg) -> Tensor:
    """KV-cached equivalent of generate_image: one prefill over
    prompt + [IMG], then n_query-1 single-token steps whose input embedding is
    project_up(project_down(h_prev)).  Mathematically identical to the uncached loop
    for unpadded rows; only GEMM-shape-dependent rounding differs."""
    B = prompt_ids.shape[0]
    ids = torch.cat((prompt_ids, torch.full((B, 1), BOI_ID, dtype=prompt_ids.dtype)), dim=1)
    x = embed_tokens(ids, W)
    if image is not None:
        e = encode_image(image, W, cfg)
        e = F.linear(e.reshape(-1, e.shape[-1]), W["project_up.weight"])
        x = scatter_image_embeds(x, ids, e)
    cache = KVCache(cfg.llama.layers)
    mask = torch.ones_like(ids)
    h = llama_model(x, mask, W, cfg.llama, cache=cache)[:, -1:, :]
    outs = [F.linear(h, W["project_down.weight"])]
    for _ in range(cfg.n_query - 1):
        x = F.linear(outs[-1], W["project_up.weight"])
        mask = torch.cat((mask, torch.ones(B, 1, dtype=mask.dtype)), dim=1)
        h = llama_model(x, mask, W, cfg.llama, cache=cache)
        outs.append(F.linear(h, W["project_down.weight"]))
    return torch.cat(outs, dim=1)


# --------------------------------------------------------------------------- helpers
def cast_weights(W: Weights, dtype) -> Weights:
    return {k: v.to(dtype) for k, v in W.items()}


def bf16_round(W: Weights) -> Weights:
    """fp32 tensors holding bf16-representable values (what the HIP path stores)."""
    return {k: v.to(torch.bfloat16).to(torch.float32) for k, v in W.items()}
