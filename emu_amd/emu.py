"""``EmuModel`` -- drop-in for the reference's ``Emu2/emu/emu.py:19-235`` on MI355X.

Same constructor configs, attribute names, method names, keyword arguments, defaults and return types
(``encode_image`` / ``generate`` / ``generate_image``), and the same state-dict keys
(``visual.*``, ``decoder.lm.model.*``, ``decoder.lm.lm_head.weight``, ``project_up.weight``,
``project_down.weight``).  All arithmetic runs in libemu_hip.so (hand-written gfx950 kernels); this file is
host orchestration only.  ``generate_image`` uses the KV-cached formulation (1 prefill + n_query-1 cached steps)
that is mathematically identical to the reference's 64 uncached forwards.
"""
from __future__ import annotations

import os
from typing import Dict, Iterable, List, Optional, Tuple

import torch

from . import ops
from .conf.emu_conf import CLIPVisionCfg, LlamaCfg, TextDecoderCfg
from .constants import *  # noqa: F401,F403  (re-exported like the reference module does)
from .constants import (DEFAULT_gIMG_TOKEN, DEFAULT_IMAGE_TOKEN, DEFAULT_IMG_END_TOKEN, DEFAULT_IMG_PLACEHOLDER,
                        DEFAULT_IMG_TOKEN, DEFAULT_VID_PLACEHOLDER, EOS_TOKEN_ID, IMAGE_TOKEN_ID, IMG_TOKEN_ID,
                        PAD_TOKEN_ID, VOCAB_EMU2, VOCAB_EMU2_CHAT, gIMG_TOKEN_ID, special_tokens_list)
from .llama import EmuHipContext, LlamaEngine

BF16 = torch.bfloat16


def build_tokenizer(llama_config_path: str, instruct: bool):
    """LlamaTokenizer + the Emu special tokens, exactly as the reference builds it (Emu2/emu/lm.py:40-63).
    Pure host logic; needs ``tokenizer.model`` in ``llama_config_path`` (ship it with the checkpoint)."""
    import transformers
    from .constants import DEFAULT_BOS_TOKEN, DEFAULT_EOS_TOKEN, DEFAULT_PAD_TOKEN
    if not os.path.exists(os.path.join(llama_config_path, "tokenizer.model")):
        raise FileNotFoundError(
            f"no tokenizer.model under {llama_config_path!r}: point TextDecoderCfg.llama_config_path (or the "
            "EMU_LLAMA_CONFIG environment variable) at the checkpoint's llama_config directory")
    tok = transformers.LlamaTokenizer.from_pretrained(llama_config_path)
    tok.add_special_tokens(dict(pad_token=DEFAULT_PAD_TOKEN, bos_token=DEFAULT_BOS_TOKEN, eos_token=DEFAULT_EOS_TOKEN,
                                additional_special_tokens=special_tokens_list(instruct)))
    tok.truncation_side = tok.padding_side = "left"          # emu.py:58
    return tok


class _Decoder:
    """Stand-in for ``EmuForClsAndRegression`` (Emu2/emu/lm.py:30-102): exposes ``.lm`` and ``.tokenizer``."""

    def __init__(self, lm: LlamaEngine, cfg: TextDecoderCfg):
        self.lm = lm
        self.args = cfg
        self._tokenizer = None
        self.image_token_id = IMAGE_TOKEN_ID
        self.img_token_id = IMG_TOKEN_ID

    @property
    def tokenizer(self):
        if self._tokenizer is None:
            path = os.environ.get("EMU_LLAMA_CONFIG", self.args.llama_config_path)
            self._tokenizer = build_tokenizer(path, self.args.instruct)
        return self._tokenizer

    @tokenizer.setter
    def tokenizer(self, tok):
        self._tokenizer = tok

    def get_num_layers(self):
        return self.lm.cfg.num_hidden_layers


class EmuModel:
    def __init__(self, vision_cfg: CLIPVisionCfg = CLIPVisionCfg(), text_decoder_cfg: TextDecoderCfg = TextDecoderCfg(),
                 *, llama_cfg: Optional[LlamaCfg] = None, device="cuda", tp_rank: int = 0, tp_size: int = 1,
                 ctx: Optional[EmuHipContext] = None):
        from .vit import VitEngine
        self.vision_cfg, self.text_decoder_cfg = vision_cfg, text_decoder_cfg
        if llama_cfg is None:
            llama_cfg = LlamaCfg.from_json(text_decoder_cfg.llama_config_path)
        self.llama_cfg = llama_cfg
        dev = torch.device(device)
        if dev.type == "cuda" and dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        self.ctx = ctx or EmuHipContext(dev, tp_rank, tp_size)
        self.vocab = VOCAB_EMU2_CHAT if text_decoder_cfg.instruct else VOCAB_EMU2
        self.visual = VitEngine(vision_cfg, self.ctx)
        self.decoder = _Decoder(LlamaEngine(llama_cfg, self.vocab, self.ctx), text_decoder_cfg)
        self.project_up: Optional[torch.Tensor] = None       # [hidden, width]   EVA -> LM   (emu.py:53)
        self.project_down: Optional[torch.Tensor] = None     # [width, hidden]   LM -> EVA   (emu.py:55)
        self.n_query = vision_cfg.n_query
        self.v_query = vision_cfg.v_query
        self.image_placeholder = DEFAULT_IMG_TOKEN + DEFAULT_IMAGE_TOKEN * self.n_query + DEFAULT_IMG_END_TOKEN
        self.video_placeholder = DEFAULT_IMG_TOKEN + DEFAULT_gIMG_TOKEN * self.v_query + DEFAULT_IMG_END_TOKEN
        self.use_graph = False
        # beam-search conventions: the transformers release the reference pins (Emu2/requirements.txt:2); "5.x" = the installed
        # library's vectorised search, the one the golden fixtures can pin (LlamaEngine.beam_search_generate)
        self.hf_semantics = "4.31"

    _warned_431 = False          # one warning per process: the default beam mode has no library vector (generate_ids)

    # ------------------------------------------------------------------ nn.Module-like surface
    def device(self, module=None):
        return self.ctx.device

    def dtype(self, module=None):
        return BF16

    def eval(self):
        return self

    def load_state_dict(self, state_dict, strict: bool = True):
        return self.load_weights(state_dict.items(), strict=strict)

    def load_weights(self, items: Iterable[Tuple[str, torch.Tensor]], strict: bool = True):
        """Streaming loader (reference key names); tensors may live on CPU or GPU, any float dtype."""
        unexpected = []
        for name, t in items:
            if name.startswith("visual."):
                used = self.visual.load_tensor(name[len("visual."):], t)
            elif name.startswith("decoder.lm."):
                used = self.decoder.lm.load_tensor(name[len("decoder.lm."):], t)
            elif name == "project_up.weight":
                self.project_up, used = t.to(self.ctx.device, BF16).contiguous(), True
            elif name == "project_down.weight":
                self.project_down, used = t.to(self.ctx.device, BF16).contiguous(), True
            else:
                used = False
            if not used and not name.endswith("rotary_emb.inv_freq"):
                unexpected.append(name)
        missing = []
        if not self.visual.ready:
            missing.append("visual.*")
        if not self.decoder.lm.ready:
            missing.append("decoder.lm.*")
        if self.project_up is None:
            missing.append("project_up.weight")
        if self.project_down is None:
            missing.append("project_down.weight")
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict for EmuModel: missing {missing}, unexpected {unexpected}")
        return missing, unexpected

    # ------------------------------------------------------------------ encode_image (emu.py:77-90)
    @torch.no_grad()
    def encode_image(self, image: torch.Tensor, *, n_query=None):
        n_query = n_query if n_query is not None else self.n_query
        g = self.vision_cfg.grid
        stride = int(g // (n_query ** 0.5))

        def enc(img):
            return ops.avgpool_tokens(self.visual(img), g, stride)    # [k, 1+g*g, C] -> [k, n_query, C]
        # tensor-parallel ranks split the IMAGES of a prompt between them (the ViT is replicated) and all-gather the pooled tokens
        if self.ctx.tp_size > 1 and image.shape[0] >= 2 and self._image_parallel_ok():
            from .tp import image_parallel_encode
            return image_parallel_encode(image, enc, self.ctx.tp_rank, self.ctx.tp_size, self._all_gather)
        return enc(image)

    def _image_parallel_ok(self) -> bool:
        import torch.distributed as dist
        return dist.is_available() and dist.is_initialized() and dist.get_world_size() == self.ctx.tp_size

    @staticmethod
    def _all_gather(t: torch.Tensor):
        """torch.distributed.all_gather of a device tensor: RCCL over xGMI; under a gloo group (ranks sharing one GPU in the
        validation runs) the exchange goes through the host."""
        import torch.distributed as dist
        if dist.get_backend() == "gloo":
            parts = [torch.empty_like(t, device="cpu") for _ in range(dist.get_world_size())]
            dist.all_gather(parts, t.cpu())
            return [p.to(t.device) for p in parts]
        parts = [torch.empty_like(t) for _ in range(dist.get_world_size())]
        dist.all_gather(parts, t)
        return parts

    def _project(self, x2d: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
        return ops.linear(x2d.contiguous(), w)

    def _prompt_embeds(self, input_ids: torch.Tensor, image: Optional[torch.Tensor], n_query: Optional[int],
                       token_id: int = IMAGE_TOKEN_ID, embeds: Optional[torch.Tensor] = None) -> torch.Tensor:
        """embed_tokens + masked overwrite of the <image> rows with project_up(encode_image) (emu.py:193-203)."""
        lm = self.decoder.lm
        B, S = input_ids.shape
        x = lm.embed_tokens(input_ids).view(B * S, -1) if embeds is None else embeds
        if image is not None:
            e = self.encode_image(image, n_query=n_query)
            e = self._project(e.view(-1, e.shape[-1]), self.project_up)
            rows = torch.nonzero(input_ids.reshape(-1).to(self.ctx.device) == token_id).reshape(-1).to(torch.int32)
            if rows.numel() != e.shape[0]:
                raise ValueError(f"shape mismatch: {rows.numel()} image slots in the prompt cannot take "
                                 f"{e.shape[0]} image embedding rows")
            ops.scatter_rows(e, rows.contiguous(), x)
        return x

    # ------------------------------------------------------------------ generate (emu.py:155-235)
    @torch.no_grad()
    def generate_ids(self, input_ids: torch.Tensor, attention_mask: torch.Tensor, image: Optional[torch.Tensor] = None,
                     video: Optional[torch.Tensor] = None, max_new_tokens: int = 10, min_len: int = 1,
                     stop_on_eos: bool = True, num_beams: int = 1, length_penalty: float = -1.0, do_sample: bool = False,
                     temperature=None, top_k=None, top_p=None, repetition_penalty: float = 1.0,
                     penalty_alpha: Optional[float] = None, no_repeat_ngram_size: int = 0,
                     num_return_sequences: int = 1, hf_semantics: Optional[str] = None,
                     eos_token_id: Optional[int] = None, min_new_tokens: Optional[int] = None) -> torch.Tensor:
        """``generate`` at the token-id level: returns the NEW ids [B, n] (what HF returns for inputs_embeds).  Mode
        selection as transformers does it: contrastive search (penalty_alpha > 0, top_k > 1, one beam, no sampling), beam
        search / beam sampling (num_beams > 1), sampling or penalised greedy, plain greedy (device-side loop, hipGraph).
        ``hf_semantics`` (default: ``self.hf_semantics`` = "4.31", the transformers release the reference pins): beam-search
        conventions, see ``LlamaEngine.beam_search_generate``; "5.x" = the installed library the golden fixtures come from.
        The two releases also differ in what ``min_length=min_len`` (emu.py:220) enforces when generation is driven by
        ``inputs_embeds``: 4.31 compares against the EMPTY id sequence (min_len new tokens), 5.x subtracts the prompt length first
        (``_prepare_generated_length``: the default min_len = 1 then enforces nothing and EOS may be the first token -- pinned by
        the real reference's ids in tests/golden/generate_beam_eos_tiny.npz).  ``eos_token_id`` / ``min_new_tokens``: the options
        of that name the reference forwards to ``lm.generate`` through ``**kwargs`` (emu.py:175,228)."""
        B, S = input_ids.shape
        sem = hf_semantics or getattr(self, "hf_semantics", "4.31")
        eos = EOS_TOKEN_ID if eos_token_id is None else int(eos_token_id)
        if min_new_tokens is not None:
            min_len = max(int(min_len), int(min_new_tokens)) if sem == "4.31" else int(min_new_tokens)
        elif sem != "4.31":
            min_len = max(int(min_len) - S, 0)
        if num_beams > 1 and sem == "4.31" and min_len < 1:
            # 4.31's BeamHypotheses.add scores an EOS hypothesis over hyp.shape[-1] ** length_penalty with the EOS excluded: a
            # hypothesis that ends at the first step (cur == 0) has length 0 (0 ** lp), a case that release only avoids through its
            # min_length = 1 default; the restatement here does not define it either (checked before any device work)
            raise ValueError("beam search under hf_semantics='4.31' needs min_len >= 1 (an EOS hypothesis of length 0 is scored over "
                             "0 ** length_penalty in that release); pass min_len >= 1 or hf_semantics='5.x'")
        x = self._prompt_embeds(input_ids, image, self.n_query, IMAGE_TOKEN_ID)
        if video is not None:
            x = self._prompt_embeds(input_ids, video, self.v_query, gIMG_TOKEN_ID, embeds=x)
        ngram, nret = int(no_repeat_ngram_size or 0), int(num_return_sequences)
        if (penalty_alpha is not None and penalty_alpha > 0 and top_k is not None and top_k > 1 and num_beams == 1
                and not do_sample):
            if ngram or nret != 1:
                raise NotImplementedError("contrastive search with no_repeat_ngram_size / several returned sequences is not built")
            return self.decoder.lm.contrastive_generate(x.view(B, S, -1), attention_mask, max_new_tokens, float(penalty_alpha),
                                                        int(top_k), min_len, repetition_penalty, eos_id=eos,
                                                        pad_id=PAD_TOKEN_ID)
        if num_beams > 1:
            if sem == "4.31" and not EmuModel._warned_431:
                # honesty in code, not only in INTEGRATION.md: this is the DEFAULT mode (emu.py:163-172: 5 beams, length_penalty -1)
                EmuModel._warned_431 = True
                import warnings
                warnings.warn("emu_amd: beam search under hf_semantics='4.31' (the transformers release the reference pins) follows a "
                              "RESTATEMENT of that release's BeamSearchScorer: no vector from the 4.31 library itself exists in "
                              "tests/golden/ (the library is not installable offline).  hf_semantics='5.x' is the mode pinned to the "
                              "installed library's ids; the two differ only where a hypothesis ends on EOS (INTEGRATION.md).",
                              stacklevel=2)
            return self.decoder.lm.beam_search_generate(x.view(B, S, -1), attention_mask, num_beams, max_new_tokens, min_len,
                                                        length_penalty, eos_id=eos, pad_id=PAD_TOKEN_ID,
                                                        do_sample=do_sample, temperature=temperature, top_k=top_k, top_p=top_p,
                                                        repetition_penalty=repetition_penalty, no_repeat_ngram_size=ngram,
                                                        num_return_sequences=nret, hf_semantics=sem)
        if do_sample or repetition_penalty != 1.0 or ngram or nret != 1:
            return self.decoder.lm.sample_generate(x.view(B, S, -1), attention_mask, max_new_tokens, min_len, do_sample,
                                                   temperature, top_k, top_p, repetition_penalty, eos_id=eos,
                                                   pad_id=PAD_TOKEN_ID, no_repeat_ngram_size=ngram, num_return_sequences=nret)
        return self.decoder.lm.greedy_generate(x.view(B, S, -1), attention_mask, max_new_tokens, min_len,
                                               eos_id=eos, pad_id=PAD_TOKEN_ID, use_graph=self.use_graph,
                                               stop_on_eos=stop_on_eos)

    @torch.no_grad()
    def generate(self, text: List[str], image: Optional[torch.Tensor] = None, video: Optional[torch.Tensor] = None,
                 image_placeholder: str = DEFAULT_IMG_PLACEHOLDER, video_placeholder: str = DEFAULT_VID_PLACEHOLDER,
                 num_beams=5, max_new_tokens=10, min_len=1, do_sample=False, penalty_alpha=None, top_p=None,
                 top_k=None, temperature=None, length_penalty=-1, repetition_penalty=1.0, synced_gpus=False,
                 skip_special_tokens=True, **kwargs):
        # the reference forwards **kwargs to transformers' generate (emu.py:175,228); the options this engine honours
        # are mapped, anything else is refused rather than silently ignored
        min_new = int(kwargs.pop("min_new_tokens")) if "min_new_tokens" in kwargs else None
        if "min_length" in kwargs:
            min_len = max(int(min_len), int(kwargs.pop("min_length")))
        eos_override = kwargs.pop("eos_token_id", None)
        kwargs.pop("use_cache", None)                   # always cached
        ngram = int(kwargs.pop("no_repeat_ngram_size", 0) or 0)
        nret = int(kwargs.pop("num_return_sequences", 1) or 1)
        if kwargs:
            raise TypeError(f"EmuModel.generate: unsupported generation options {sorted(kwargs)} "
                            "(the HIP engine implements greedy / beam / sampling / contrastive search with the arguments of the signature)")
        tok = self.decoder.tokenizer
        text = [t.replace(image_placeholder, self.image_placeholder).replace(video_placeholder, self.video_placeholder)
                for t in text]
        inputs = tok(text, padding="longest", return_tensors="pt")
        ids = self.generate_ids(inputs.input_ids, inputs.attention_mask, image, video, max_new_tokens, min_len,
                                num_beams=num_beams, length_penalty=length_penalty, do_sample=do_sample,
                                temperature=temperature, top_k=top_k, top_p=top_p, repetition_penalty=repetition_penalty,
                                penalty_alpha=penalty_alpha, no_repeat_ngram_size=ngram, num_return_sequences=nret,
                                eos_token_id=eos_override, min_new_tokens=min_new)
        return tok.batch_decode(ids.cpu(), skip_special_tokens=skip_special_tokens)

    # ------------------------------------------------------------------ generate_image (emu.py:92-153)
    @torch.no_grad()
    def generate_image_ids(self, prompt_ids: torch.Tensor, image: Optional[torch.Tensor] = None,
                           attention_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Visual-embedding regression at the token-id level: prompt + [IMG] prefill, then n_query-1 cached steps with input
        project_up(project_down(h_prev)).  Rows of different length arrive LEFT-padded with ``attention_mask``; every row then
        sees the positions 0..len-1 of its own tokens (pads are masked out of the attention), i.e. exactly what it computes
        alone -- the rows share one weight stream per step instead of being run one by one."""
        lm = self.decoder.lm
        B, S0 = prompt_ids.shape
        ids = torch.cat((prompt_ids.to(torch.int64), torch.full((B, 1), IMG_TOKEN_ID, dtype=torch.int64)), dim=1)
        S = S0 + 1
        if S + self.n_query > lm.cfg.max_position_embeddings:
            raise ValueError("prompt too long for generate_image")
        x = self._prompt_embeds(ids, image, self.n_query, IMAGE_TOKEN_ID)
        mask = torch.ones(B, S, dtype=torch.int64)
        if attention_mask is not None:
            mask[:, :S0] = attention_mask.to(torch.int64)
        # positions from the mask: arange(S) for an unpadded row (what the reference's lm.model call uses), the row's own
        # 0..len-1 behind left padding
        hidden, kstart, pos = lm.prefill(x.view(B, S, -1), mask, hf_generate_positions=True)
        h = lm.final_norm_rows(hidden[:, -1, :].contiguous())
        first = self._project(h, self.project_down)                                # [B, width]
        if self.n_query == 1:
            return first[:, None, :]
        # replayed from a hipGraph where the host would bound the loop (tensor-parallel shards: a step is ~2.7 ms of small
        # launches at TP = 8); at TP = 1 the eager loop stays ahead of the GPU and measured 2 % FASTER than the replay
        # (10.25 vs 10.48 ms per step: profiles/r04_bench_tp1_*.json), so it is the default there
        if not getattr(self, "regress_graph", self.ctx.tp_size > 1):
            outs = [first]
            for j in range(self.n_query - 1):
                xin = self._project(outs[-1], self.project_up)                    # [B, hidden]
                hj = lm.decode_embeds(xin, pos, S + j, kstart)
                pos = pos + 1
                outs.append(self._project(lm.final_norm_rows(hj), self.project_down))
            return torch.stack(outs, dim=1)                                        # [B, n_query, width]
        return self._regress_replayed(first, pos, S, kstart)

    def _regress_replayed(self, first: torch.Tensor, pos: torch.Tensor, S: int, kstart: torch.Tensor) -> torch.Tensor:
        """The n_query - 1 cached steps of ``generate_image`` with the loop state on the device: one step -- project_up, the decoder
        step on the persistent KV cache, final norm, project_down, ``emu_regress_advance_bf16`` (stores the embedding at the
        device-side step index, makes it the next input, advances positions and slots) -- is captured into a hipGraph once per
        (batch, cache) and replayed; nothing returns to the host before the last step (at TP = 8 a shard's step is 2.7 ms of small
        launches: an eager Python loop would bound it)."""
        from ._lib import check, lib
        lm = self.decoder.lm
        B, width, hidden, dev = first.shape[0], first.shape[1], lm.cfg.hidden_size, self.ctx.device
        L = lib()
        key = (B, lm.s_max, self.n_query, lm.kcache.data_ptr(), lm.mode_epoch)
        st = getattr(self, "_regress_state", None)
        if st is None or st["key"] != key:
            i32 = dict(dtype=torch.int32, device=dev)
            st = dict(key=key, prev=torch.empty(B, width, device=dev, dtype=BF16), xin=torch.empty(B, hidden, device=dev, dtype=BF16),
                      normed=torch.empty(B, hidden, device=dev, dtype=BF16), cur=torch.empty(B, width, device=dev, dtype=BF16),
                      out_all=torch.empty(self.n_query, B, width, device=dev, dtype=BF16), pos=torch.empty(B, **i32),
                      slot=torch.empty(B, **i32), step=torch.empty(1, **i32), kstart=torch.empty(B, **i32), graph=None,
                      ws=torch.empty(max(int(L.emu_llama_workspace_bytes(lm.handle, B, 1)), B * hidden * 2), dtype=torch.uint8, device=dev))
            self._regress_state = st
        st["prev"].copy_(first); st["out_all"][0].copy_(first)
        st["pos"].copy_(pos); st["slot"].fill_(S); st["step"].fill_(1); st["kstart"].copy_(kstart)

        def body():
            sm = ops.stream(dev)
            ops.linear(st["prev"], self.project_up, out=st["xin"])
            check(L.emu_llama_forward(lm.handle, st["xin"].data_ptr(), B, 1, st["pos"].data_ptr(), st["slot"].data_ptr(),
                                      st["kstart"].data_ptr(), None, lm.s_max, st["ws"].data_ptr(), st["ws"].numel(), sm),
                  "emu_llama_forward", self.ctx.handle)
            check(L.emu_llama_final_norm(lm.handle, st["xin"].data_ptr(), st["normed"].data_ptr(), B, sm), "emu_llama_final_norm")
            ops.linear(st["normed"], self.project_down, out=st["cur"])
            check(L.emu_regress_advance_bf16(st["cur"].data_ptr(), st["out_all"].data_ptr(), st["prev"].data_ptr(), st["pos"].data_ptr(),
                                             st["slot"].data_ptr(), st["step"].data_ptr(), B, width, sm), "emu_regress_advance_bf16")

        n = self.n_query - 1
        if st["graph"] is None:
            body()                                              # warm-up outside capture: a real step
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                body()
            st["graph"] = g
            n -= 1
        for _ in range(n):
            st["graph"].replay()
        lm.ctx.check_p2p()
        lm.check_decode_fused()
        return st["out_all"].permute(1, 0, 2).clone()                               # [B, n_query, width] (never a view of the state)

    @torch.no_grad()
    def generate_image(self, text: List[str], image: Optional[torch.Tensor] = None,
                       placeholder: str = DEFAULT_IMG_PLACEHOLDER, warn_ragged: bool = True):
        tok = self.decoder.tokenizer
        text = [t.replace(placeholder, self.image_placeholder) for t in text]
        inputs = tok(text, padding="longest", return_tensors="pt")
        if not bool(inputs.attention_mask.all()):
            # the reference re-pads every iteration and calls lm.model without position_ids, so its
            # left-padded rows see RoPE positions shifted by their pad count; here every row keeps the positions of its own
            # tokens, which is what the same rows compute un-padded (= the reference at batch size 1).
            import warnings
            if warn_ragged:
                warnings.warn("generate_image: prompts of different token lengths are computed as at batch size 1 (each row on "
                              "its own positions); the reference's batched call shifts the RoPE positions of its left-padded "
                              "rows and returns different embeddings for them", stacklevel=2)
            if tok.padding_side != "left":
                tok.padding_side, side = "left", tok.padding_side
                try:
                    inputs = tok(text, padding="longest", return_tensors="pt")
                finally:
                    tok.padding_side = side
            return self.generate_image_ids(inputs.input_ids, image, inputs.attention_mask)
        return self.generate_image_ids(inputs.input_ids, image)
