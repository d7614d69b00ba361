"""LLaMA decoder engine: host side of the ``emu_llama_*`` C ABI.

Plays the role of ``transformers.LlamaForCausalLM`` inside the reference's ``EmuForClsAndRegression``
(Emu2/emu/lm.py:30-40) for the calls the hot path makes: ``lm.model.embed_tokens`` (emu.py:119,193),
``lm.model(inputs_embeds=...)`` (emu.py:133-138) and greedy ``lm.generate`` (emu.py:213-229).
All arithmetic runs in libemu_hip.so; torch only owns the memory.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Iterable, Optional, Tuple

import torch

from . import ops
from ._lib import EmuHipError, LlamaCfgC, check, lib
from .conf.emu_conf import LlamaCfg
from .tp import ShardPlan

BF16 = torch.bfloat16
EOS_POLL = 16          # greedy decode: steps between two host looks at the emitted ids (early stop on EOS)
_LAYER_KEYS = ("self_attn.q_proj.weight", "self_attn.k_proj.weight", "self_attn.v_proj.weight",
               "self_attn.o_proj.weight", "mlp.gate_proj.weight", "mlp.up_proj.weight", "mlp.down_proj.weight",
               "input_layernorm.weight", "post_attention_layernorm.weight")


def rope_tables(head_dim: int, max_pos: int, theta: float, device) -> Tuple[torch.Tensor, torch.Tensor]:
    """cos/sin [max_pos, D] bf16 exactly as transformers' LlamaRotaryEmbedding builds them (fp32 math, cast)."""
    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    freqs = torch.arange(max_pos, dtype=torch.float32)[:, None] * inv[None, :]
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos().to(BF16).to(device).contiguous(), emb.sin().to(BF16).to(device).contiguous()


class EmuHipContext:
    """One per (process, device): owns the emu_ctx handle and, for tp_size > 1, the RCCL communicator.  One process drives
    one GPU: creating the context makes its device the process's current device (explicitly, here and nowhere else --
    ``ops.stream`` refuses operands of another device instead of switching to it)."""

    def __init__(self, device: torch.device, tp_rank: int = 0, tp_size: int = 1):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise ValueError("emu_amd needs a GPU device (no CPU path)")
        if self.device.index is not None and self.device.index != torch.cuda.current_device():
            torch.cuda.set_device(self.device)
        self.tp_rank, self.tp_size = tp_rank, tp_size
        self.p2p = False
        self.p2p_fence_free = False
        h = C.c_void_p()
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        check(lib().emu_ctx_create(idx, tp_rank, tp_size, C.byref(h)), "emu_ctx_create")
        self.handle = h

    def init_tp(self, broadcast_bytes, force: bool = False, allgather_bytes=None, rccl: bool = True,
                p2p_timeout_ms: int = 0) -> None:
        """Create the RCCL communicator.  ``broadcast_bytes(b: bytes|None) -> bytes`` must return rank 0's
        128-byte unique id on every rank (e.g. via the torch.distributed store).  ``force`` creates a communicator
        even for tp_size 1 (a 1-rank RCCL all-reduce: used to exercise the RCCL + hipGraph path on a single GPU).

        ``allgather_bytes(b: bytes) -> list[bytes]`` (every rank's bytes, in rank order) additionally sets up the one-shot
        peer-to-peer all-reduce for the decode-sized messages (csrc/p2p.hip): IPC handles are exchanged, every rank runs a
        self-test, and the path is enabled only if it passed on ALL ranks -- otherwise the all-reduces stay on RCCL and
        ``self.p2p`` is False.  ``rccl=False`` skips the communicator (ranks sharing one GPU in the tests, which RCCL
        refuses); the P2P path is then mandatory."""
        if self.tp_size == 1 and not force:
            return
        if rccl:
            buf = (C.c_char * 128)()
            if self.tp_rank == 0:
                check(lib().emu_tp_unique_id(buf), "emu_tp_unique_id")
                uid = broadcast_bytes(bytes(buf))
            else:
                uid = broadcast_bytes(None)
            buf = (C.c_char * 128).from_buffer_copy(uid)
            with torch.cuda.device(self.device):
                check(lib().emu_tp_init(self.handle, buf), "emu_tp_init", self.handle)
        if allgather_bytes is not None:
            self.p2p = self._init_p2p(allgather_bytes, p2p_timeout_ms)
        if not rccl and not self.p2p:
            raise RuntimeError("tensor parallelism without RCCL needs the peer-to-peer all-reduce, and its self-test failed")

    def _init_p2p(self, allgather_bytes, timeout_ms: int) -> bool:
        L = lib()
        h = (C.c_char * 64)()
        ok = L.emu_tp_p2p_create(self.handle, h) == 0
        handles = allgather_bytes(bytes(h) if ok else b"")
        if not all(len(x) == 64 for x in handles):           # some rank could not export: nobody maps anything
            return False
        ok = L.emu_tp_p2p_open(self.handle, (C.c_char * (64 * self.tp_size)).from_buffer_copy(b"".join(handles)), timeout_ms) == 0
        ok = all(x == b"1" for x in allgather_bytes(b"1" if ok else b"0"))   # also the barrier: every peer block is mapped
        if ok:
            # self-test: four all-reduces (both slots, one reuse each) of rank-dependent vectors; exact in bf16
            n = 6656
            base = torch.arange(n, device=self.device, dtype=torch.float32) % 13
            want = sum(base * (r + 1) + r for r in range(self.tp_size)).to(BF16)
            s = ops.stream(self.device)
            for it in range(4):
                x = (base * (self.tp_rank + 1) + self.tp_rank).to(BF16)
                ok = ok and L.emu_tp_p2p_allreduce_bf16(self.handle, x.data_ptr(), n, s) == 0
                torch.cuda.synchronize(self.device)
                ok = ok and bool(torch.equal(x, want))
            ok = ok and L.emu_tp_p2p_giveups() == 0
        ok = all(x == b"1" for x in allgather_bytes(b"1" if ok else b"0"))
        if ok:
            # the self-test above ran the FENCED form (system-scope release / acquire: the memory model's guarantee).  The fence-free
            # form (3.3 instead of 6.8 us per decode all-reduce) rests on write-through + acknowledgement behaviour instead, so it is
            # switched on only after a soak of it passed on EVERY rank of this very job (real xGMI links where the ranks have their
            # own GPUs); EMU_P2P_FENCE_FREE=0 keeps the fenced form, =1 skips the soak (tools on one GPU).
            want_ff = os.environ.get("EMU_P2P_FENCE_FREE", "auto")
            ff = want_ff == "1"
            if want_ff == "auto":
                check(L.emu_tp_p2p_set_fenced(self.handle, 0), "emu_tp_p2p_set_fenced", self.handle)
                ff = self._soak_p2p()
                ff = all(x == b"1" for x in allgather_bytes(b"1" if ff else b"0"))
                if not ff and self.tp_rank == 0:
                    import warnings
                    warnings.warn("fence-free peer-to-peer all-reduce failed its soak; keeping the fenced form")
            check(L.emu_tp_p2p_set_fenced(self.handle, 0 if ff else 1), "emu_tp_p2p_set_fenced", self.handle)
            self.p2p_fence_free = ff
            check(L.emu_tp_p2p_enable(self.handle, 1), "emu_tp_p2p_enable", self.handle)
        elif self.tp_rank == 0:
            import warnings
            warnings.warn("peer-to-peer all-reduce self-test failed; tensor-parallel all-reduces stay on RCCL")
        return ok

    def _soak_p2p(self, iters: int = 384) -> bool:
        """Back-to-back all-reduces of sequence- and rank-dependent vectors in the CURRENT form of the exchange, one host
        synchronisation at the end, every word of every result checked against the closed form (small integers: exact in bf16).
        Both slots are reused ~190 times each with no host gap between uses -- the window in which a late write-through or a
        stale read would surface -- and two slot-sized (256 KiB) messages exercise all 32 pieces."""
        L, n, dev = lib(), 6656, self.device
        i = torch.arange(n, device=dev, dtype=torch.int64)
        k = torch.arange(iters, device=dev, dtype=torch.int64)[:, None]
        val = lambda r: ((i * 7 + k * 3 + r * 5) % 16).to(torch.float32)
        x = val(self.tp_rank).to(BF16).contiguous()
        want = sum(val(r) for r in range(self.tp_size)).to(BF16)
        s = ops.stream(dev)
        ok = True
        for j in range(iters):
            ok = ok and L.emu_tp_p2p_allreduce_bf16(self.handle, x[j].data_ptr(), n, s) == 0
        nb = 128 * 1024
        ib = torch.arange(nb, device=dev, dtype=torch.int64)
        big = [((ib * 3 + j + self.tp_rank * 11) % 8).to(BF16) for j in range(2)]
        for j in range(2):
            ok = ok and L.emu_tp_p2p_allreduce_bf16(self.handle, big[j].data_ptr(), nb, s) == 0
        torch.cuda.synchronize(dev)
        ok = ok and bool(torch.equal(x, want))
        for j in range(2):
            ok = ok and bool(torch.equal(big[j], sum(((ib * 3 + j + r * 11) % 8).to(torch.float32) for r in range(self.tp_size)).to(BF16)))
        return ok and L.emu_tp_p2p_giveups() == 0

    def check_p2p(self) -> None:
        """Raise if a device-side wait of the P2P all-reduce ever timed out (the sums since then are invalid)."""
        if self.p2p and lib().emu_tp_p2p_giveups() != 0:
            # the path cannot recover (its per-piece sequence counters have diverged across the ranks): leave it for good, so
            # that whatever this process runs next goes through RCCL (where a communicator exists) instead of summing stale slots
            self.p2p = False
            lib().emu_tp_p2p_enable(self.handle, 0)
            raise RuntimeError("peer-to-peer all-reduce timed out waiting for a peer rank; the results of this generation are invalid "
                               "(the peer-to-peer path is now switched off for this process)")

    def allreduce(self, t: torch.Tensor) -> torch.Tensor:
        check(lib().emu_allreduce_bf16(self.handle, t.data_ptr(), t.numel(), ops.stream(self.device)), "emu_allreduce_bf16", self.handle)
        return t

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                lib().emu_ctx_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class LlamaEngine:
    def __init__(self, cfg: LlamaCfg, vocab: int, ctx: EmuHipContext):
        self.cfg, self.vocab, self.ctx = cfg, vocab, ctx
        self.device = ctx.device
        self.plan = ShardPlan(cfg.hidden_size, cfg.num_attention_heads, cfg.head_dim, cfg.intermediate_size,
                              ctx.tp_size, ctx.tp_rank)
        c = LlamaCfgC(cfg.hidden_size, self.plan.heads_local, cfg.head_dim, self.plan.ffn_local,
                      cfg.num_hidden_layers, vocab, cfg.max_position_embeddings, cfg.rms_norm_eps)
        h = C.c_void_p()
        check(lib().emu_llama_create(ctx.handle, C.byref(c), C.byref(h)), "emu_llama_create", ctx.handle)
        self.handle = h
        # prefill() always hands the rows of a prompt over in slot order, which is what the fused RoPE / KV-append / V^T epilogue of
        # the qkv projection needs to be told (include/emu_hip.h: emu_llama_set_prefill_fusion); set_prefill_fusion(False) = the
        # three-launch sequence (A/B timing, parity tests)
        self.set_prefill_fusion(True)
        self._keep: Dict[str, torch.Tensor] = {}          # packed weights (owned here, pointers held by the lib)
        self._pending: Dict[int, Dict[str, torch.Tensor]] = {}
        self.layers_loaded = set()          # indices of the packed layers (a reload must not count twice)
        self.cos, self.sin = rope_tables(cfg.head_dim, cfg.max_position_embeddings, cfg.rope_theta, self.device)
        self.embed = self.final_norm = self.lm_head = None
        self.head_rows = None                               # (row0, row1) of the vocabulary this rank's lm_head holds under TP
        self.kcache = self.vcache = None
        self.kv_batch = self.s_max = 0
        self._ws = None
        # every captured decode graph (GreedyState, beam search, EmuModel's regress loop) carries the epoch it was captured under in
        # its key: a mode switch (fp8 weights, decode tail, fused layers) bumps it, so no graph of another mode is ever replayed
        self.mode_epoch = 0
        # whole decoder layers per launch (csrc/decode_layer.hip): built, bit-identical, and measured SLOWER than the launches on
        # MI355X (12.45 vs 10.57 ms per token at TP = 1, 3.95 vs 3.57 for a TP = 8 shard: profiles/r05_decode_fused_*.log), so it
        # is an option (EMU_DECODE_FUSED=1|2 or set_decode_fused), off by default
        self.decode_fused = 0
        mode = int(os.environ.get("EMU_DECODE_FUSED", "0"))
        if mode:
            self.set_decode_fused(mode)
        # tensor parallelism: the two-lane prefill (prompts of >= N rows run as two row halves whose all-reduces travel behind the
        # other half's GEMMs, emu_llama_set_tp_overlap) is OPT-IN: EMU_TP_OVERLAP=N (e.g. 1024) or set_tp_overlap(N).  What it costs
        # is measured (+17...21 % per rank under graph replay, +40...47 % eager, profiles/r05_tp_prefill_two_lane_per_rank_cost.log);
        # what it hides is an estimate until a multi-GPU node has run it (bench.py --tp-prefill-leg collects that evidence), and it
        # issues RCCL all-reduces of one communicator from two streams, which no run has exercised against a real ring.
        self.tp_overlap_rows = 0
        if ctx.tp_size > 1 and int(os.environ.get("EMU_TP_OVERLAP", "0")) > 0:
            self.set_tp_overlap(int(os.environ["EMU_TP_OVERLAP"]))

    def _mode_changed(self) -> None:
        self.mode_epoch += 1
        self.__dict__.pop("_beam_graphs", None)

    def set_decode_tail(self, enable: bool) -> None:
        """Decode attention in one launch (the last split workgroup of a head merges the splits) or, the default, with the separate
        combine launch.  Same bits; measured 0.4 % slower in one launch.  Invalidates captured decode graphs."""
        check(lib().emu_llama_set_decode_tail(self.handle, 1 if enable else 0), "emu_llama_set_decode_tail", self.ctx.handle)
        self._mode_changed()

    def set_decode_fused(self, enable: int, layers_per_launch: int = 0) -> None:
        """One-row decode steps (greedy decode, ``generate_image``'s regress loop) with bf16 weights run whole decoder layers per
        launch (csrc/decode_layer.hip: ``layers_per_launch`` layers, 0 = all) instead of six launches + two all-reduces per layer;
        bit-identical to the launches.  Under tensor parallelism 1 = cut at the all-reduces (four launches per layer), 2 = all-reduces
        inside the launch over the P2P comm blocks (one GPU per rank), 3 = stand-alone launches whose last workgroup runs the
        all-reduce in its tail (attention, o_proj, down_proj as single-role launches: five per layer, safe on a shared device);
        4 = the persistent weight-streaming engine (csrc/decode_engine.hip: the attention launches, then ONE launch per layer for
        o_proj -> all-reduce -> gate/up -> down -> all-reduce -> the next layer's qkv; tensor-parallel shards with rows of at most
        13 KiB, i.e. TP >= 4, over the P2P comm blocks, one GPU per rank; anything else silently keeps the launches).  OFF by default:
        every mode measured slower than the launches (see __init__; EMU_DECODE_FUSED=<mode> in the environment switches one on).
        Invalidates captured decode graphs."""
        check(lib().emu_llama_set_decode_fused(self.handle, int(enable), int(layers_per_launch)), "emu_llama_set_decode_fused",
              self.ctx.handle)
        self.decode_fused = int(enable)
        self._mode_changed()

    def decode_fused_stats(self) -> Tuple[int, int]:
        """(give-ups of the fused path's bounded waits -- non-zero means garbage was computed --, forwards that took the fused path)."""
        g, f = C.c_uint(0), C.c_long(0)
        check(lib().emu_llama_decode_fused_stats(self.handle, C.byref(g), C.byref(f)), "emu_llama_decode_fused_stats", self.ctx.handle)
        return int(g.value), int(f.value)

    def check_decode_fused(self) -> None:
        g, _ = self.decode_fused_stats()
        if g:
            raise EmuHipError(f"fused decode layers: {g} in-kernel wait(s) ran into the time limit; the step's outputs are invalid")

    def set_tp_overlap(self, min_rows: int) -> None:
        """Tensor-parallel prefill of prompts with at least ``min_rows`` rows (B = 1; clamped up to 512) in two row halves: the
        all-reduce of one half's o_proj / down_proj partial sums runs on the context's second stream while this stream computes the
        other half's stage (include/emu_hip.h: emu_llama_set_tp_overlap).  0 = every all-reduce serially between the GEMMs.  Results
        agree with the serial schedule to bf16 rounding (a half may take another GEMM tile configuration)."""
        check(lib().emu_llama_set_tp_overlap(self.handle, int(min_rows)), "emu_llama_set_tp_overlap", self.ctx.handle)
        self.tp_overlap_rows = 0 if min_rows <= 0 else max(512, int(min_rows))

    def tp_overlap_count(self) -> int:
        """Forwards that took the two-half schedule."""
        return int(lib().emu_llama_tp_overlap_count(self.handle))

    def set_prefill_fusion(self, enable: bool) -> None:
        """Whether ``prefill`` promises the library slot-ordered rows (the fused RoPE / KV-append / V^T / norm epilogues).  The
        promise itself is per call (emu_llama_set_prefill_fusion is consumed by the next T > 1 forward): ``prefill`` renews it
        before every forward it issues; a direct ``forward`` call with rows in any other slot order never inherits it."""
        self.prefill_fusion = bool(enable)
        if not enable:
            check(lib().emu_llama_set_prefill_fusion(self.handle, 0), "emu_llama_set_prefill_fusion", self.ctx.handle)

    # ------------------------------------------------------------------ weights
    def _dev(self, t: torch.Tensor) -> torch.Tensor:
        return t.to(device=self.device, dtype=BF16).contiguous()

    def load_tensor(self, name: str, t: torch.Tensor) -> bool:
        """Consume one reference-named tensor (``model.layers.N...`` / ``model.embed_tokens.weight`` /
        ``model.norm.weight`` / ``lm_head.weight``, i.e. keys relative to ``decoder.lm.``). Returns True if used."""
        if name == "model.embed_tokens.weight":
            self.embed = self._dev(t)
        elif name == "model.norm.weight":
            self.final_norm = self._dev(t)
        elif name == "lm_head.weight":
            if self.ctx.tp_size > 1 and os.environ.get("EMU_TP_VOCAB_SHARD", "1") != "0":
                # vocabulary-sharded head: this rank keeps rows [r0, r1) only -- 4035 of 32 274 at TP = 8, 54 MB instead of
                # 430 MB of the token's byte budget; emu_llama_logits all-reduces the rows whole again (bit-identical logits)
                r0, r1 = self.plan.vocab_range(t.shape[0])
                self.lm_head = self._dev(t[r0:r1])
                self.head_rows = (r0, r1)
            else:
                self.lm_head = self._dev(t)
        elif name.startswith("model.layers."):
            rest = name[len("model.layers."):]
            idx, key = rest.split(".", 1)
            if key not in _LAYER_KEYS:
                return False                                   # e.g. rotary_emb.inv_freq buffers of old checkpoints
            d = self._pending.setdefault(int(idx), {})
            d[key] = t
            if len(d) == len(_LAYER_KEYS):
                self._pack_layer(int(idx), self._pending.pop(int(idx)))
        else:
            return False
        if self.embed is not None and self.final_norm is not None and self.lm_head is not None:
            check(lib().emu_llama_set_head(self.handle, self.final_norm.data_ptr(), self.lm_head.data_ptr(),
                                           self.embed.data_ptr(), self.cos.data_ptr(), self.sin.data_ptr()),
                  "emu_llama_set_head")
            if self.head_rows is not None:
                check(lib().emu_llama_set_head_shard(self.handle, self.head_rows[0], self.head_rows[1] - self.head_rows[0]),
                      "emu_llama_set_head_shard", self.ctx.handle)
        return True

    def _pack_layer(self, i: int, d: Dict[str, torch.Tensor]) -> None:
        g = lambda k: d[k].to(device=self.device, dtype=BF16)
        p = self.plan.pack_layer(g("self_attn.q_proj.weight"), g("self_attn.k_proj.weight"),
                                 g("self_attn.v_proj.weight"), g("self_attn.o_proj.weight"),
                                 g("mlp.gate_proj.weight"), g("mlp.up_proj.weight"), g("mlp.down_proj.weight"))
        p["ln1"] = self._dev(d["input_layernorm.weight"])
        p["ln2"] = self._dev(d["post_attention_layernorm.weight"])
        for k, v in p.items():
            self._keep[f"{i}.{k}"] = v
        check(lib().emu_llama_set_layer(self.handle, i, p["wqkv"].data_ptr(), p["wo"].data_ptr(), p["wgu"].data_ptr(),
                                        p["wdown"].data_ptr(), p["ln1"].data_ptr(), p["ln2"].data_ptr()),
              "emu_llama_set_layer")
        self.layers_loaded.add(i)

    def load_weights(self, items: Iterable[Tuple[str, torch.Tensor]], prefix: str = "decoder.lm.") -> None:
        for name, t in items:
            if name.startswith(prefix):
                self.load_tensor(name[len(prefix):], t)

    @property
    def ready(self) -> bool:
        return (len(self.layers_loaded) == self.cfg.num_hidden_layers and self.embed is not None
                and self.final_norm is not None and self.lm_head is not None)

    def weight_bytes_per_token(self) -> int:
        """Algorithmic bytes one decode step of THIS shard must stream (all packed matrices once + lm_head)."""
        b = 1 if getattr(self, "fp8_decode", False) else 2
        per = sum(v.numel() * b for k, v in self._keep.items() if k.split(".")[1] in ("wqkv", "wo", "wgu", "wdown"))
        return per + (self.lm_head.numel() * b if self.lm_head is not None else 0)

    # ------------------------------------------------------------------ optional fp8 decode stream
    def quantize_fp8(self) -> None:
        """Build per-row-scaled e4m3fn copies of every packed matrix (+ lm_head) on the device and register them for the
        decode stream (B*T <= 2 rows).  Prefill keeps the bf16 weights, so both sets stay resident (288 GB HBM:
        33B bf16 + fp8 = 98 GB).  Not a reference feature (the reference is bf16 end to end); off unless enabled."""
        if not self.ready:
            raise RuntimeError("quantize_fp8: load all weights first")
        if getattr(self, "_fp8", None):
            return
        self._fp8 = {}
        for i in range(self.cfg.num_hidden_layers):
            a = []
            for k in ("wqkv", "wo", "wgu", "wdown"):
                q, sc = ops.quantize_fp8_rows(self._keep[f"{i}.{k}"])
                self._fp8[f"{i}.{k}"] = (q, sc)
                a += [q.data_ptr(), sc.data_ptr()]
            check(lib().emu_llama_set_layer_fp8(self.handle, i, *a), "emu_llama_set_layer_fp8", self.ctx.handle)
        q, sc = ops.quantize_fp8_rows(self.lm_head)
        self._fp8["lm_head"] = (q, sc)
        check(lib().emu_llama_set_head_fp8(self.handle, q.data_ptr(), sc.data_ptr()), "emu_llama_set_head_fp8")

    def use_fp8(self, enable: bool = True, prefill: bool = False) -> None:
        """Switch the decode stream between the bf16 and the fp8 weights (invalidates captured decode graphs).
        ``prefill=True`` also runs the prefill GEMMs W8A8 on the block-scaled fp8 MFMA (activations quantised per row)."""
        if enable:
            self.quantize_fp8()
        check(lib().emu_llama_use_fp8(self.handle, (2 if prefill else 1) if enable else 0), "emu_llama_use_fp8", self.ctx.handle)
        self.fp8_decode = bool(enable)
        self._mode_changed()

    def fp8_dequantized(self, key: str) -> torch.Tensor:
        """fp32 value of a registered fp8 tensor (tests: feed the oracle the exact weights the stream uses)."""
        q, sc = self._fp8[key]
        return q.view(torch.float8_e4m3fn).to(torch.float32) * sc[:, None]

    # ------------------------------------------------------------------ KV cache / workspace
    KV_BUCKETS = (256, 512, 1024)

    def kv_capacity(self, need: int) -> int:
        """KV slots to allocate for a request that will touch ``need`` positions: the smallest bucket that holds it
        (requests up to 1024 positions run the one-launch decode attention and a smaller cache), else the model
        maximum.  Buckets keep the allocation -- and any captured decode graph's launch geometry -- reusable."""
        cap = self.cfg.max_position_embeddings
        if need > cap:
            raise ValueError("prompt + max_new_tokens exceeds max_position_embeddings")
        for b in self.KV_BUCKETS:
            if need <= b <= cap:
                return b
        return cap

    def alloc_kv(self, batch: int, s_max: int, zero: bool = True, which: str = "main") -> None:
        """Make a cache of ``batch`` rows x ``s_max`` slots the engine's current one.  Two caches persist side by side, the
        "main" one (prefill, greedy / sampling) and the "beam" one (``fan_out_kv``: prompts x beams rows), each re-used while its
        shape stays the same: the pointers a captured hipGraph holds stay valid from call to call."""
        slots = self.__dict__.setdefault("_kv_slots", {})
        cur = slots.get(which)
        if cur is None or cur[2] != batch or cur[3] != s_max:
            L, Hl, D = self.cfg.num_hidden_layers, self.plan.heads_local, self.cfg.head_dim
            slots[which] = None                            # free the old one first
            cur = None
            make = torch.zeros if zero else torch.empty
            cur = (make(L, batch, Hl, s_max, D, device=self.device, dtype=BF16),
                   make(L, batch, Hl, s_max, D, device=self.device, dtype=BF16), batch, s_max)
            slots[which] = cur
            self._ws = None
            self.__dict__.pop("_beam_graphs", None)        # graphs captured on the replaced cache are void
        if self.kcache is cur[0]:
            self.set_kv_share(0, 0)
            return
        self.kcache, self.vcache, self.kv_batch, self.s_max = cur
        check(lib().emu_llama_set_kv(self.handle, self.kcache.data_ptr(), self.vcache.data_ptr(), batch, s_max),
              "emu_llama_set_kv")
        self._kv_share = False

    KV_SHARE_MAX = 8                                   # DECODE_SHARE_MAX of the decode attention kernel

    def release_kv(self, which: Optional[str] = None) -> None:
        """Free the persistent KV caches (``which`` = "main" / "beam" / None for both): alloc_kv keeps the main and the beam cache
        resident for the life of the engine (several GB per rank after one 5-beam search at 60 layers) so that captured graphs stay
        valid; a server that is done with beam search hands the memory back here.  Captured decode graphs are dropped."""
        slots = self.__dict__.get("_kv_slots", {})
        for k in ([which] if which else list(slots)):
            cur = slots.pop(k, None)
            if cur is not None and self.kcache is cur[0]:      # the engine's current cache goes: the next prefill allocates anew
                self.kcache = self.vcache = None
        if self.kcache is None:
            # the library must not keep pointers into memory the allocator may hand out again: detach, so that a forward / step /
            # regress call before the next alloc_kv fails with -22 instead of writing K / V into freed memory
            self.kv_batch = self.s_max = 0
            check(lib().emu_llama_set_kv(self.handle, None, None, 0, 0), "emu_llama_set_kv", self.ctx.handle)
        self._mode_changed()

    def set_kv_share(self, rows_per_prompt: int, shared_slots: int) -> None:
        """Groups of ``rows_per_prompt`` consecutive cache rows keep their first ``shared_slots`` slots (the prompt) in the
        group's first row only (include/emu_hip.h: emu_llama_set_kv_share); (0, 0) = every row owns its slots."""
        if rows_per_prompt <= 1 and not getattr(self, "_kv_share", False):
            return
        check(lib().emu_llama_set_kv_share(self.handle, int(rows_per_prompt), int(shared_slots)), "emu_llama_set_kv_share")
        self._kv_share = rows_per_prompt > 1

    def fan_out_kv(self, B: int, n: int, S: int, s_max: int) -> None:
        """After a prefill of B prompts: a cache of B * n rows (n beams / candidates per prompt) that continue the prompts.
        transformers copies the prompt's keys and values to every beam (``_expand_inputs_for_generation``); here, for
        n <= 8, they stay in ONE row per prompt and the decode attention reads them from there (``set_kv_share``): no n-fold
        copy (6 GB at S = 770, 5 beams, 60 layers) and the prompt's keys cross the memory system once per step, not n times."""
        k_old, v_old = self.kcache, self.vcache
        # no zero fill: every slot a row reads was written first (the prompt's by the copy below -- with set_kv_share only the
        # group's first row is ever read there -- and a generated slot by the step that appends it)
        self.alloc_kv(B * n, s_max, zero=False, which="beam")
        if 2 <= n <= self.KV_SHARE_MAX:
            first = torch.arange(B, device=self.device) * n
            self.kcache[:, first, :, :S] = k_old[:, :, :, :S]
            self.vcache[:, first, :, :S] = v_old[:, :, :, :S]
            self.set_kv_share(n, S)
        else:
            rep = torch.arange(B, device=self.device).repeat_interleave(n)
            self.kcache[:, :, :, :S] = k_old[:, rep, :, :S]
            self.vcache[:, :, :, :S] = v_old[:, rep, :, :S]

    def _workspace(self, B: int, T: int) -> torch.Tensor:
        need = lib().emu_llama_workspace_bytes(self.handle, B, T)
        need = max(need, B * T * self.cfg.hidden_size * 2)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, device=self.device, dtype=torch.uint8)
        return self._ws

    # ------------------------------------------------------------------ forward
    def embed_tokens(self, ids: torch.Tensor) -> torch.Tensor:
        """lm.model.embed_tokens (emu.py:119,193): ids [..] -> [.., hidden] bf16."""
        flat = ids.reshape(-1).to(device=self.device, dtype=torch.int32).contiguous()
        return ops.embed_gather(flat, self.embed).view(*ids.shape, self.cfg.hidden_size)

    def forward(self, hidden: torch.Tensor, B: int, T: int, pos: torch.Tensor, slot: torch.Tensor,
                kstart: Optional[torch.Tensor], ctx: int, ctx_dev: Optional[torch.Tensor] = None) -> torch.Tensor:
        """All decoder layers in place on the residual stream ``hidden`` [B*T, hidden] (not final-normed)."""
        assert hidden.is_contiguous() and hidden.dtype == BF16 and hidden.shape == (B * T, self.cfg.hidden_size)
        ws = self._workspace(B, T)
        check(lib().emu_llama_forward(self.handle, hidden.data_ptr(), B, T, pos.data_ptr(), slot.data_ptr(),
                                      ops._p(kstart), ops._p(ctx_dev), ctx, ws.data_ptr(), ws.numel(), ops.stream(self.device)),
              "emu_llama_forward", self.ctx.handle)
        return hidden

    def set_layer_range(self, l0: int = 0, l1: int = -1) -> None:
        """Parity hook (include/emu_hip.h: emu_llama_set_layer_range): ``forward`` / ``prefill`` run layers [l0, l1) only;
        (0, -1) restores the whole stack.  Tests only."""
        check(lib().emu_llama_set_layer_range(self.handle, int(l0), int(l1)), "emu_llama_set_layer_range", self.ctx.handle)

    def final_norm_rows(self, hidden: torch.Tensor) -> torch.Tensor:
        out = torch.empty_like(hidden)
        check(lib().emu_llama_final_norm(self.handle, hidden.data_ptr(), out.data_ptr(), hidden.shape[0], ops.stream(self.device)),
              "emu_llama_final_norm")
        return out

    def logits(self, hidden_rows: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """final RMSNorm + lm_head on rows [M, hidden] -> [M, vocab] bf16."""
        M = hidden_rows.shape[0]
        if out is None:
            out = torch.empty(M, self.vocab, device=self.device, dtype=BF16)
        ws = self._workspace(max(M, 1), 1)
        check(lib().emu_llama_logits(self.handle, hidden_rows.data_ptr(), hidden_rows.stride(0), M, out.data_ptr(),
                                     out.stride(0), ws.data_ptr(), ws.numel(), ops.stream(self.device)),
              "emu_llama_logits", self.ctx.handle)
        return out

    # ------------------------------------------------------------------ model-level calls
    def prefill(self, embeds: torch.Tensor, attention_mask: torch.Tensor, s_max: Optional[int] = None,
                hf_generate_positions: bool = True):
        """Run S prompt slots through all layers and fill the KV cache.

        ``embeds`` [B,S,hidden] bf16, ``attention_mask`` [B,S] (1 = real token, left padded).
        ``hf_generate_positions``: position_ids = cumsum(mask)-1 as ``lm.generate`` does; False = arange(S) as the
        bare ``lm.model(...)`` call of generate_image does (emu.py:133-138).
        Returns (hidden [B,S,hidden] residual stream, kstart [B] int32, next_pos [B] int32)."""
        B, S, H = embeds.shape
        am = attention_mask.to(device=self.device, dtype=torch.int64)
        n_real = am.sum(dim=1)
        self.alloc_kv(B, s_max or self.cfg.max_position_embeddings)
        if S > self.s_max:
            raise ValueError(f"prompt length {S} exceeds KV capacity {self.s_max}")
        if hf_generate_positions:
            pos = (am.cumsum(-1) - 1).masked_fill(am == 0, 1)
            next_pos = n_real
        else:
            pos = torch.arange(S, device=self.device)[None].expand(B, -1)
            next_pos = torch.full((B,), S, device=self.device)
        slot = torch.arange(S, device=self.device)[None].expand(B, -1)
        kstart = (S - n_real).to(torch.int32).contiguous()
        hidden = embeds.to(device=self.device, dtype=BF16).reshape(B * S, H).contiguous().clone()
        if self.prefill_fusion:                          # slot = arange(S) below: the promise the fused epilogues need, for this call
            check(lib().emu_llama_set_prefill_fusion(self.handle, 1), "emu_llama_set_prefill_fusion", self.ctx.handle)
        self.forward(hidden, B, S, pos.reshape(-1).to(torch.int32).contiguous(),
                     slot.reshape(-1).to(torch.int32).contiguous(), kstart, ctx=S)
        return hidden.view(B, S, H), kstart, next_pos.to(torch.int32).contiguous()

    def decode_embeds(self, x: torch.Tensor, pos: torch.Tensor, slot_idx: int, kstart: torch.Tensor) -> torch.Tensor:
        """One cached step on explicit input embeddings x [B, hidden] (used by generate_image)."""
        B = x.shape[0]
        hidden = x.to(BF16).contiguous().clone()
        slot = torch.full((B,), slot_idx, device=self.device, dtype=torch.int32)
        self.forward(hidden, B, 1, pos, slot, kstart, ctx=slot_idx + 1)
        return hidden

    @torch.no_grad()
    def greedy_generate(self, embeds: torch.Tensor, attention_mask: torch.Tensor, max_new_tokens: int,
                        min_len: int = 1, eos_id: int = 2, pad_id: int = 32000, use_graph: bool = False,
                        stop_on_eos: bool = True) -> torch.Tensor:
        """``lm.generate(inputs_embeds=..., num_beams=1, do_sample=False)`` (emu.py:213-229): returns only the
        new ids [B, n].  The whole token loop stays on the device (embed -> layers -> logits -> argmax ->
        state advance per step, optionally replayed from one hipGraph); EOS/PAD bookkeeping of finished rows is
        applied afterwards on the host, which is equivalent because rows never interact."""
        B, S, H = embeds.shape
        s_max = self.kv_capacity(S + max_new_tokens)
        hidden, kstart, next_pos = self.prefill(embeds, attention_mask, s_max)
        last = hidden[:, -1, :]
        logits = torch.empty(B, self.vocab, device=self.device, dtype=BF16)
        self.logits(last, out=logits)
        cur = ops.argmax(logits, suppress_id=eos_id if min_len >= 1 else -1)
        out_ids = torch.full((max_new_tokens, B), pad_id, device=self.device, dtype=torch.int32)
        out_ids[0] = cur
        if max_new_tokens > 1:
            st = GreedyState(self, B, cur, next_pos, S, kstart, out_ids)
            n_eager = max(0, min_len - 1)
            steps = max_new_tokens - 1
            for i in range(steps):
                suppress = eos_id if i < n_eager else -1
                if suppress >= 0:
                    st.step_eager_suppress(suppress)
                elif use_graph:
                    st.step_graph()
                else:
                    st.step()
                # the reference's loop ends when every row has emitted EOS (transformers' stopping criteria): look at the
                # device-side ids every EOS_POLL steps (one small host read; the launches in between stay queued ahead)
                if stop_on_eos and (i + 1) % EOS_POLL == 0 and i + 1 < steps:
                    if bool((out_ids[: i + 2] == eos_id).any(dim=0).all()):
                        break
        ids = out_ids.t().to(torch.int64)                       # [B, max_new]
        self.ctx.check_p2p()
        self.check_decode_fused()
        if not stop_on_eos:
            return ids
        return apply_eos_padding(ids, eos_id, pad_id)


    # ------------------------------------------------------------------ sampling
    @torch.no_grad()
    def sample_generate(self, embeds: torch.Tensor, attention_mask: torch.Tensor, max_new_tokens: int, min_len: int = 1,
                        do_sample: bool = True, temperature: Optional[float] = None, top_k: Optional[int] = None,
                        top_p: Optional[float] = None, repetition_penalty: float = 1.0, eos_id: int = 2,
                        pad_id: int = 32000, no_repeat_ngram_size: int = 0, num_return_sequences: int = 1) -> torch.Tensor:
        """``lm.generate(inputs_embeds=..., num_beams=1)`` with logits processing: repetition penalty, min_length, then
        (when sampling) temperature / top-k / top-p warpers and a multinomial draw from torch's global CUDA generator
        (transformers' processor order).  The decoder runs on the HIP engine; the tiny per-step logits post-processing is
        host-driven, so this path is not graph-replayed.  ``num_return_sequences`` = n samples every prompt n times (rows
        prompt-major, as the library expands its inputs); without sampling the library refuses n > 1 and so does this."""
        if num_return_sequences > 1:
            if not do_sample:
                raise ValueError("Greedy methods without beam search do not support `num_return_sequences` different than 1")
            embeds = embeds.repeat_interleave(num_return_sequences, dim=0)
            attention_mask = attention_mask.repeat_interleave(num_return_sequences, dim=0)
        B, S, H = embeds.shape
        dev = self.device
        s_max = self.kv_capacity(S + max_new_tokens)
        hidden, kstart, pos = self.prefill(embeds, attention_mask, s_max)
        row = hidden[:, -1, :]
        out = torch.full((B, max_new_tokens), pad_id, dtype=torch.int64, device=dev)
        unfinished = torch.ones(B, dtype=torch.int64, device=dev)
        hid = torch.empty(B, H, device=dev, dtype=BF16)
        n = 0
        for step in range(max_new_tokens):
            scores = process_logits(self.logits(row).float(), out[:, :step], step < min_len, eos_id, do_sample, temperature,
                                    top_k, top_p, repetition_penalty, no_repeat_ngram_size=int(no_repeat_ngram_size or 0))
            if do_sample:
                nxt = torch.multinomial(torch.softmax(scores, dim=-1), num_samples=1).squeeze(1)
            else:
                nxt = scores.argmax(dim=-1)
            nxt = nxt * unfinished + pad_id * (1 - unfinished)
            out[:, step] = nxt
            n = step + 1
            unfinished = unfinished * (nxt != eos_id).long()
            if int(unfinished.max().item()) == 0 or n == max_new_tokens:
                break
            ops.embed_gather(nxt.to(torch.int32).contiguous(), self.embed, out=hid)
            slot = torch.full((B,), S + step, device=dev, dtype=torch.int32)
            self.forward(hid, B, 1, pos, slot, kstart, ctx=S + step + 1)
            pos = pos + 1
            row = hid
        self.ctx.check_p2p()
        return out[:, :n]

    # ------------------------------------------------------------------ contrastive search
    @torch.no_grad()
    def contrastive_generate(self, embeds: torch.Tensor, attention_mask: torch.Tensor, max_new_tokens: int,
                             penalty_alpha: float, top_k: int, min_len: int = 1, repetition_penalty: float = 1.0,
                             eos_id: int = 2, pad_id: int = 32000, trace: Optional[dict] = None,
                             force_ids: Optional[torch.Tensor] = None) -> torch.Tensor:
        """``lm.generate(inputs_embeds=..., penalty_alpha=a, top_k=k)`` -- contrastive search as the reference's pinned
        transformers 4.31 runs it (Emu2/emu/emu.py:167,223 forwards both arguments): per step the k most probable tokens
        are each run one step further, and the one maximising (1 - a) * p(token) - a * max_j cos(h_token, h_j) over the
        final-norm hidden states h_j of everything before it is kept.  The k candidates are k extra rows of one decoder
        step that share the prompt's KV rows (as beams do); the kept candidate's KV slot is then copied to its siblings.
        a = 0 or k = 1 degenerate to greedy search (tested against the real reference's greedy ids).  Deviation: hidden
        states of left-padding positions are left out of the similarity (the library includes whatever the masked rows
        hold).  Parity beyond those invariants is unpinned: the transformers in this image no longer ships the mode.

        Diagnostics for the tests: ``trace`` receives ``margin`` (smallest winner / runner-up score gap) and ``steps`` (per
        step the candidate ids, their probabilities, their degeneration penalties and the selection, on the host);
        ``force_ids`` [B, n] replays a given token path (each forced token must be one of the step's k candidates) so two
        engines can be compared number by number along the same path."""
        B, S, H = embeds.shape
        k, V, dev = int(top_k), self.vocab, self.device
        s_max = self.kv_capacity(S + max_new_tokens)
        hidden, kstart, next_pos = self.prefill(embeds, attention_mask, s_max)
        ctx_h = self.final_norm_rows(hidden.reshape(B * S, H).contiguous()).view(B, S, H).float()
        ctx_ok = torch.arange(S, device=dev)[None, :] >= kstart[:, None].to(dev)
        logit = self.logits(hidden[:, -1, :]).float()
        self.fan_out_kv(B, k, S, s_max)
        kstart_k = kstart.repeat_interleave(k).contiguous()
        pos = next_pos.repeat_interleave(k).contiguous()
        out = torch.full((B, max_new_tokens), pad_id, dtype=torch.int64, device=dev)
        unfinished = torch.ones(B, dtype=torch.int64, device=dev)
        hid = torch.empty(B * k, H, device=dev, dtype=BF16)
        ar = torch.arange(B, device=dev)
        n = 0
        for step in range(max_new_tokens):
            scores = process_logits(logit, out[:, :step], step < min_len, eos_id, False, repetition_penalty=repetition_penalty)
            top_p_, top_ids = torch.topk(torch.softmax(scores, dim=-1), k=k, dim=-1)          # [B, k]
            ops.embed_gather(top_ids.reshape(-1).to(torch.int32).contiguous(), self.embed, out=hid)
            slot = torch.full((B * k,), S + step, device=dev, dtype=torch.int32)
            self.forward(hid, B * k, 1, pos, slot, kstart_k, ctx=S + step + 1)
            nh = self.final_norm_rows(hid).float().view(B, k, H)
            cand_logits = self.logits(hid).float().view(B, k, V)
            cn = ctx_h / ctx_h.norm(dim=2, keepdim=True)
            nn_ = nh / nh.norm(dim=2, keepdim=True)
            cos = torch.einsum("bld,bkd->bkl", cn, nn_).masked_fill(~ctx_ok[:, None, :], -float("inf"))
            pen = cos.max(dim=-1)[0]
            score = (1.0 - penalty_alpha) * top_p_ - penalty_alpha * pen
            sel = score.argmax(dim=-1)                                                                    # [B]
            if force_ids is not None:
                hit = top_ids == force_ids[:, step, None].to(dev)
                if not bool((hit.any(dim=1) | ~unfinished.bool()).all()):
                    raise ValueError(f"contrastive_generate: forced token of step {step} is not among the {k} candidates")
                sel = torch.where(unfinished.bool(), hit.float().argmax(dim=1), sel)
            if trace is not None:
                t2 = torch.topk(score, k=2, dim=-1)[0]
                gap = torch.where(unfinished.bool(), t2[:, 0] - t2[:, 1], torch.full_like(t2[:, 0], float("inf")))
                trace["margin"] = min(trace.get("margin", float("inf")), float(gap.min()))
                trace.setdefault("steps", []).append(dict(ids=top_ids.cpu(), p=top_p_.float().cpu(), pen=pen.float().cpu(),
                                                          sel=sel.cpu(), live=unfinished.bool().cpu()))
            nxt = top_ids[ar, sel]
            nxt = nxt * unfinished + pad_id * (1 - unfinished)
            out[:, step] = nxt
            n = step + 1
            unfinished = unfinished * (nxt != eos_id).long()
            if int(unfinished.max().item()) == 0 or n == max_new_tokens:
                break
            # the kept candidate's KV slot goes to its k - 1 siblings; its hidden state joins the context
            src = (ar * k + sel).repeat_interleave(k)
            self.kcache[:, :, :, S + step] = self.kcache[:, src, :, S + step]
            self.vcache[:, :, :, S + step] = self.vcache[:, src, :, S + step]
            ctx_h = torch.cat((ctx_h, nh[ar, sel][:, None, :]), dim=1)
            ctx_ok = torch.cat((ctx_ok, torch.ones(B, 1, dtype=torch.bool, device=dev)), dim=1)
            logit = cand_logits[ar, sel]
            pos = pos + 1
        self.set_kv_share(0, 0)
        self.ctx.check_p2p()
        return out[:, :n]

    # ------------------------------------------------------------------ beam search
    @torch.no_grad()
    def beam_search_generate(self, embeds: torch.Tensor, attention_mask: torch.Tensor, num_beams: int,
                             max_new_tokens: int, min_len: int = 1, length_penalty: float = -1.0, eos_id: int = 2,
                             pad_id: int = 32000, do_sample: bool = False, temperature: Optional[float] = None,
                             top_k: Optional[int] = None, top_p: Optional[float] = None,
                             repetition_penalty: float = 1.0, hf_semantics: str = "4.31",
                             trace: Optional[dict] = None, no_repeat_ngram_size: int = 0,
                             num_return_sequences: int = 1) -> torch.Tensor:
        """``lm.generate(inputs_embeds=..., num_beams=N, do_sample=False, early_stopping=False)`` -- the reference's
        DEFAULT decoding mode (num_beams=5, length_penalty=-1, Emu2/emu/emu.py:163-172,213-229).  Restates
        transformers' vectorised beam search: per step keep the 2N best continuations over beams x vocab, the N best
        non-finished ones keep running, finished ones (EOS, or the length limit) compete for the N result slots with
        score / len**length_penalty, and the loop ends when no running beam can beat the worst kept result.
        One prefill for the B prompts; every beam then gets a cache row, of which only the GENERATED slots are its own
        (the prompt's stay in one row per prompt, ``fan_out_kv``) and are re-ordered by beam index each step (gathered on
        the device).  Returns the best sequence per prompt [B, <= max_new_tokens].

        ``do_sample=True`` is the library's *beam-search multinomial sampling*: the per-beam log-probabilities go through
        the logits pipeline (repetition penalty, min length, temperature / top-k / top-p with min_tokens_to_keep = 2)
        BEFORE the beam scores are added, and the 2N continuations are drawn without replacement from
        softmax(accumulated scores) instead of taken by top-k.  ``repetition_penalty`` alone gives penalised beam search.

        ``hf_semantics`` selects the ORDER of the sampling pipeline, which changed between the transformers the reference
        pins (4.31, ``Emu2/requirements.txt:2``) and the one installed here (5.x, the only one that can be run, so the
        only one the golden fixture pins): "4.31" = ``beam_sample`` of that release -- logits processors, THEN the running
        beam scores are added, THEN the warpers (temperature / top-k / top-p) act on the accumulated rows, and the 2N
        draws are sorted by score before the beam bookkeeping (so "only the first N candidates may finish" is by rank, not
        by draw order), and all N beams start at score 0 instead of (0, -1e9, ...); "5.x" = processors and warpers on the per-beam log-probabilities, scores added afterwards, draws
        kept in draw order.  With temperature 1 and no top-k / top-p the two differ only in the sort.
        It also selects the SCORER conventions, in every mode (do_sample or not): "5.x" = the vectorised search of the installed
        library, which the golden fixtures pin (a hypothesis that ends, by EOS or at the length limit, is scored over cur + 1
        tokens; only the first N candidates may end; the early-stopping heuristic looks at the best running beam); "4.31" =
        ``BeamSearchScorer`` / ``BeamHypotheses`` of the pinned release: ``add`` divides an EOS hypothesis by
        ``hyp.shape[-1] ** length_penalty`` where the hypothesis excludes the EOS (cur tokens: with ``inputs_embeds`` the ids
        start empty), ``is_done`` compares the best of ALL 2N candidates at (cur + 1) ** length_penalty with the worst kept
        score, and at the length limit ``finalize`` adds every running beam at L ** length_penalty unless the prompt is done.
        With the reference's default length_penalty = -1 the two rank hypotheses of different lengths differently.  The 4.31
        conventions are restated from that release (it cannot be installed here): UNPINNED beyond a per-hypothesis restatement
        in tests/test_host_logic.py.

        ``no_repeat_ngram_size`` adds the library's NoRepeatNGramLogitsProcessor to the pipeline; ``num_return_sequences`` = n
        returns the n best results of every prompt ([B * n, len], prompt-major: what Emu1's ``num_captions`` asks for,
        Emu1/models/modeling_emu.py:110,173).

        ``trace`` (a dict, diagnostics for the tests): receives ``margin`` = the smallest gap seen between the N-th and the
        (N+1)-th running candidate at a pruning step and between the two best final results."""
        if hf_semantics not in ("4.31", "5.x"):
            raise ValueError("hf_semantics must be '4.31' or '5.x'")
        if not 1 <= num_return_sequences <= num_beams:
            raise ValueError("`num_return_sequences` has to be smaller or equal to `num_beams`")       # the library's own check
        ngram = int(no_repeat_ngram_size or 0)
        B, S, H = embeds.shape
        nb, V, dev = num_beams, self.vocab, self.device
        s_max = self.kv_capacity(S + max_new_tokens)
        hidden, kstart, next_pos = self.prefill(embeds, attention_mask, s_max)
        logits = self.logits(hidden[:, -1, :])                                          # [B, V]
        # rows b*nb .. b*nb + nb - 1 are the beams of prompt b; its keys / values stay in the first of them (fan_out_kv)
        self.fan_out_kv(B, nb, S, s_max)
        kstart_b = kstart.repeat_interleave(nb).contiguous()
        pos = next_pos.repeat_interleave(nb).contiguous()

        max_len = max_new_tokens
        NEG = -1.0e9
        # The deterministic mode (the reference's default) runs each step's selection and bookkeeping in ONE kernel
        # (emu_beam_step_bf16); sampling, penalties, n-gram bans and the diagnostics keep the torch pipeline below.
        if (not do_sample and repetition_penalty == 1.0 and not ngram and trace is None and nb <= 8 and max_len <= 256
                and V >= 2 * nb and hasattr(self, "handle") and logits.dtype == BF16):
            return self._beam_search_device(logits, B, S, nb, max_len, min_len, length_penalty, eos_id, pad_id, kstart_b, pos,
                                            int(num_return_sequences), hf_semantics == "4.31")
        logits = logits.float()
        running_seq = torch.full((B, nb, max_len), pad_id, dtype=torch.int64, device=dev)
        sequences = running_seq.clone()
        running_scores = torch.zeros(B, nb, device=dev)
        if not (do_sample and hf_semantics == "4.31"):
            running_scores[:, 1:] = NEG             # 4.31's beam_sample starts EVERY beam at 0 (the draws tell them apart)
        beam_scores = torch.full((B, nb), NEG, device=dev)
        finished = torch.zeros(B, nb, dtype=torch.bool, device=dev)
        seq_len = torch.zeros(B, nb, dtype=torch.int64, device=dev)                     # generated length of kept results
        heuristic_open = torch.ones(B, 1, dtype=torch.bool, device=dev)
        top_mask = torch.cat([torch.ones(nb, dtype=torch.bool), torch.zeros(nb, dtype=torch.bool)]).to(dev)
        gather = lambda t, idx: torch.gather(t, 1, idx.reshape(B, -1, *([1] * (t.dim() - 2))).expand(-1, -1, *t.shape[2:]))
        hid = torch.empty(B * nb, H, device=dev, dtype=BF16)

        cur = 0
        lp_rows = logits[:, None, :].expand(B, nb, V)                                   # step 0: every beam = the prompt
        margin = float("inf")
        v431 = hf_semantics == "4.31"
        old = do_sample and v431
        while True:
            log_probs = torch.log_softmax(lp_rows, dim=-1)
            if do_sample or repetition_penalty != 1.0 or ngram:
                log_probs = process_logits(log_probs.reshape(B * nb, V), running_seq[:, :, :cur].reshape(B * nb, cur), cur < min_len,
                                           eos_id, do_sample and not old, temperature, top_k, top_p, repetition_penalty,
                                           min_keep=2, no_repeat_ngram_size=ngram).view(B, nb, V)
            elif cur < min_len:
                log_probs = log_probs.clone()
                log_probs[..., eos_id] = -float("inf")
            acc = log_probs + running_scores[:, :, None]
            if old:                                    # 4.31 beam_sample: the warpers see the ACCUMULATED scores
                acc = warp_logits(acc.reshape(B * nb, V), temperature, top_k, top_p, min_keep=2).view(B, nb, V)
            acc = acc.reshape(B, nb * V)
            if do_sample:
                top_idx = torch.multinomial(torch.softmax(acc, dim=-1), num_samples=2 * nb)
                top_lp = torch.gather(acc, 1, top_idx)
                if old:                                # ... and its draws are ranked before the beam scorer sees them
                    top_lp, order = torch.sort(top_lp, descending=True, dim=1)
                    top_idx = torch.gather(top_idx, 1, order)
            else:
                top_lp, top_idx = torch.topk(acc, k=2 * nb)
            src_beam = top_idx // V
            tok = top_idx % V
            cand_seq = gather(running_seq, src_beam)
            cand_seq[:, :, cur] = tok
            at_limit = cur + 1 >= max_len
            # 4.31: at the length limit the non-EOS candidates still become running beams (finalize adds them below)
            hits = (tok == eos_id) if v431 else (tok == eos_id) | at_limit
            # running beams for the next step: best N non-finished candidates
            run_lp = top_lp + hits.float() * NEG
            nxt = torch.topk(run_lp, k=nb)[1]
            if trace is not None and cur + 1 < max_len:
                srt = torch.sort(run_lp, dim=1, descending=True)[0]
                margin = min(margin, float((srt[:, nb - 1] - srt[:, nb]).min()))
            running_seq = gather(cand_seq, nxt)
            running_scores = torch.gather(run_lp, 1, nxt)
            beam_idx = torch.gather(src_beam, 1, nxt)                                   # which old beam each new beam extends
            # finished results: only the top-N candidates may finish; merge with the kept ones
            # 4.31 BeamHypotheses.add: an EOS hypothesis is scored over the cur tokens ahead of the EOS
            fin_lp = top_lp / float((cur if (v431 and cur > 0) else cur + 1) ** length_penalty)
            fin_lp = fin_lp + (~heuristic_open).float() * NEG
            just = hits & top_mask[None, :]
            fin_lp = fin_lp + (~just).float() * NEG
            m_seq = torch.cat((sequences, cand_seq), dim=1)
            m_sc = torch.cat((beam_scores, fin_lp), dim=1)
            m_fin = torch.cat((finished, just), dim=1)
            m_len = torch.cat((seq_len, torch.full((B, 2 * nb), cur + 1, dtype=torch.int64, device=dev)), dim=1)
            keep = torch.topk(m_sc, k=nb)[1]
            sequences = gather(m_seq, keep)
            beam_scores = torch.gather(m_sc, 1, keep)
            finished = torch.gather(m_fin, 1, keep)
            seq_len = torch.gather(m_len, 1, keep)
            cur += 1
            # early-stop heuristic (early_stopping=False): can the best running beam still beat the worst kept result?
            # (4.31 is_done: the best of all 2N candidates, an EOS one included)
            best_run = (top_lp[:, :1] if v431 else running_scores[:, :1]) / float(cur ** length_penalty)
            worst_fin = torch.where(finished, beam_scores.min(dim=1, keepdim=True)[0], torch.full_like(beam_scores, NEG))
            heuristic_open = heuristic_open & (best_run > worst_fin).any(dim=-1, keepdim=True)
            if v431 and cur >= max_len:
                # 4.31 finalize: the running beams (now L tokens) join the kept results at L ** length_penalty unless done
                fin2 = running_scores / float(cur ** length_penalty) + (~heuristic_open).float() * NEG
                m_sc = torch.cat((beam_scores, fin2), dim=1)
                keep = torch.topk(m_sc, k=nb)[1]
                sequences = gather(torch.cat((sequences, running_seq), dim=1), keep)
                beam_scores = torch.gather(m_sc, 1, keep)
                finished = torch.gather(torch.cat((finished, torch.ones_like(finished)), dim=1), 1, keep)
                seq_len = torch.gather(torch.cat((seq_len, torch.full((B, nb), cur, dtype=torch.int64, device=dev)), dim=1), 1, keep)
            # (every candidate is a hit exactly when the length limit is reached: a beam contributes EOS at most once, so the 2N
            # candidates are never all EOS -- known on the host, one device read per step instead of two)
            if cur >= max_len or not bool(heuristic_open.any()):
                break
            # advance the model: reorder the cache rows by beam, feed the chosen tokens
            flat = (beam_idx + torch.arange(B, device=dev)[:, None] * nb).reshape(-1)
            ctx = S + cur - 1
            # beams of one prompt share the prompt's KV slots, so only the generated slots [S, ctx) move with the beam
            # permutation -- a few KB per layer instead of the whole live cache
            if ctx > S:
                self.kcache[:, :, :, S:ctx] = self.kcache[:, flat, :, S:ctx]
                self.vcache[:, :, :, S:ctx] = self.vcache[:, flat, :, S:ctx]
            toks = running_seq[:, :, cur - 1].reshape(-1).to(torch.int32).contiguous()
            ops.embed_gather(toks, self.embed, out=hid)
            slot = torch.full((B * nb,), ctx, device=dev, dtype=torch.int32)
            self.forward(hid, B * nb, 1, pos, slot, kstart_b, ctx=ctx + 1)
            pos = pos + 1
            lp_rows = self.logits(hid).float().view(B, nb, V)
        self.set_kv_share(0, 0)
        nret = int(num_return_sequences)
        out_len = int(seq_len[:, :nret].max().item())
        if trace is not None:
            trace["margin"] = min(margin, float((beam_scores[:, 0] - beam_scores[:, 1]).min()))
        self.ctx.check_p2p()
        if nret == 1:
            return sequences[:, 0, :out_len]
        return sequences[:, :nret, :out_len].reshape(B * nret, out_len)

    BEAM_POLL = 4          # graph-replayed beam steps between two looks at the done flags

    def _beam_search_device(self, logits0: torch.Tensor, B: int, S: int, nb: int, max_len: int, min_len: int,
                            length_penalty: float, eos_id: int, pad_id: int, kstart_b: torch.Tensor, pos: torch.Tensor,
                            nret: int, v431: bool = True) -> torch.Tensor:
        """The loop of ``beam_search_generate`` for the deterministic mode, with nothing left on the host but the replay of a
        hipGraph: per step {``emu_beam_advance`` (slot / position of the token to feed from the device-side step counter) ->
        ``emu_llama_beam_reorder_kv`` (the generated KV slots follow the beam permutation) -> embedding gather -> the decoder
        step on prompts x beams rows -> logits -> ``emu_beam_step_bf16`` (log-softmax, 2N-best selection and the scorer's
        bookkeeping, csrc/beam.hip: the same statements as the torch pipeline, which stays the specification and the fallback) ->
        counter + 1}.  The first step (every beam continues the prompt) runs eagerly on the prefill's logits; the graph is
        captured once per (prompt length, batch, beams, limits) on the persistent beam cache and replayed ``BEAM_POLL`` steps at
        a time between two reads of the done flags -- steps replayed beyond the end of the search change nothing (kept results
        are frozen once a prompt is done, and the kernels do nothing from the length limit on).  ``self.beam_graph = False``
        runs the same launches eagerly."""
        dev, V, rows = self.device, self.vocab, B * nb
        L = lib()
        key = (B, nb, S, self.s_max, max_len, int(min_len), float(length_penalty), int(eos_id), bool(v431), self.kcache.data_ptr(),
               self.mode_epoch)
        cache = self.__dict__.setdefault("_beam_graphs", {})
        st = cache.get(key)
        if st is None:
            cache.clear()                                   # one signature at a time: its buffers are a few MB, its graph ~430 nodes
            i32 = dict(dtype=torch.int32, device=dev)
            st = dict(running_seq=torch.empty(B, nb, max_len, **i32), sequences=torch.empty(B, nb, max_len, **i32),
                      running_scores=torch.empty(B, nb, device=dev), beam_scores=torch.empty(B, nb, device=dev),
                      finished=torch.empty(B, nb, dtype=torch.uint8, device=dev), seq_len=torch.empty(B, nb, **i32),
                      still_open=torch.empty(B, dtype=torch.uint8, device=dev), next_tok=torch.empty(rows, **i32),
                      beam_flat=torch.empty(rows, dtype=torch.int64, device=dev),
                      hid=torch.empty(rows, self.cfg.hidden_size, device=dev, dtype=BF16), slot=torch.empty(rows, **i32),
                      pos=torch.empty(rows, **i32), pos0=torch.empty(rows, **i32), kstart=torch.empty(rows, **i32),
                      cur=torch.empty(1, **i32), lg=torch.empty(rows, V, device=dev, dtype=BF16),
                      ws=torch.empty(L.emu_beam_step_workspace_bytes(B, nb, V), dtype=torch.uint8, device=dev), graph=None)
            # the decoder step's workspace: owned here, because the pointers a captured graph holds must outlive any
            # re-allocation of the engine's shared workspace by a later, larger call
            need = max(int(L.emu_llama_workspace_bytes(self.handle, rows, 1)), rows * self.cfg.hidden_size * 2)
            st["fws"] = torch.empty(need, dtype=torch.uint8, device=dev)
            cache[key] = st
        st["running_seq"].fill_(pad_id); st["sequences"].fill_(pad_id)
        st["running_scores"].zero_(); st["running_scores"][:, 1:] = -1.0e9
        st["beam_scores"].fill_(-1.0e9); st["finished"].zero_(); st["seq_len"].zero_(); st["still_open"].fill_(1)
        st["pos0"].copy_(pos); st["kstart"].copy_(kstart_b)
        stream = lambda: ops.stream(self.device)           # (inside a capture the current stream is the capturing one)

        def beam_step(lg, ld_prompt, ld_beam, cur, cur_dev):
            check(L.emu_beam_step_bf16(lg.data_ptr(), ld_prompt, ld_beam, V, B, nb, max_len, cur, cur_dev, int(min_len), eos_id,
                                       float(length_penalty), int(v431), st["running_seq"].data_ptr(), st["sequences"].data_ptr(),
                                       st["running_scores"].data_ptr(), st["beam_scores"].data_ptr(), st["finished"].data_ptr(),
                                       st["seq_len"].data_ptr(), st["still_open"].data_ptr(), st["next_tok"].data_ptr(),
                                       st["beam_flat"].data_ptr(), st["ws"].data_ptr(), st["ws"].numel(), stream()),
                  "emu_beam_step_bf16", self.ctx.handle)

        def body():
            """One step, every index read on the device."""
            sm = stream()
            check(L.emu_beam_advance(st["cur"].data_ptr(), st["pos"].data_ptr(), st["slot"].data_ptr(), st["pos0"].data_ptr(), S, rows,
                                     max_len, 0, sm), "emu_beam_advance")
            check(L.emu_llama_beam_reorder_kv(self.handle, st["beam_flat"].data_ptr(), st["cur"].data_ptr(), nb, S, max_len, sm),
                  "emu_llama_beam_reorder_kv", self.ctx.handle)
            ops.embed_gather(st["next_tok"], self.embed, out=st["hid"])
            check(L.emu_llama_forward(self.handle, st["hid"].data_ptr(), rows, 1, st["pos"].data_ptr(), st["slot"].data_ptr(),
                                      st["kstart"].data_ptr(), None, min(S + max_len, self.s_max), st["fws"].data_ptr(),
                                      st["fws"].numel(), sm), "emu_llama_forward", self.ctx.handle)
            check(L.emu_llama_logits(self.handle, st["hid"].data_ptr(), st["hid"].stride(0), rows, st["lg"].data_ptr(),
                                     st["lg"].stride(0), st["fws"].data_ptr(), st["fws"].numel(), sm), "emu_llama_logits",
                  self.ctx.handle)
            beam_step(st["lg"], nb * st["lg"].stride(0), st["lg"].stride(0), 0, st["cur"].data_ptr())
            check(L.emu_beam_advance(st["cur"].data_ptr(), None, None, None, S, rows, max_len, 1, sm), "emu_beam_advance")

        beam_step(logits0, logits0.stride(0), 0, 0, None)     # step 0: every beam continues the prompt
        st["cur"].fill_(1)
        done_steps = 1
        use_graph = getattr(self, "beam_graph", True)
        while done_steps < max_len:
            if not bool(st["still_open"].any()):
                break
            n = min(self.BEAM_POLL, max_len - done_steps)
            if use_graph and st["graph"] is None:
                body()                                      # warm-up outside capture: a real step
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    body()
                st["graph"] = g
                done_steps += 1
                continue
            for _ in range(n):
                if use_graph:
                    st["graph"].replay()
                else:
                    body()
            done_steps += n
        self.set_kv_share(0, 0)
        out_len = int(st["seq_len"][:, :nret].max().item())
        self.ctx.check_p2p()
        seqs = st["sequences"].long()
        if nret == 1:
            return seqs[:, 0, :out_len].clone()
        return seqs[:, :nret, :out_len].reshape(B * nret, out_len)


def process_logits(scores: torch.Tensor, generated: torch.Tensor, suppress_eos: bool, eos_id: int, do_sample: bool,
                   temperature: Optional[float] = None, top_k: Optional[int] = None, top_p: Optional[float] = None,
                   repetition_penalty: float = 1.0, min_keep: int = 1, no_repeat_ngram_size: int = 0) -> torch.Tensor:
    """transformers' logits pipeline for ``generate(inputs_embeds=...)`` in its order: RepetitionPenaltyLogitsProcessor over
    the ids generated so far (with inputs_embeds the prompt contributes no ids), MinLengthLogitsProcessor (EOS -> -inf),
    then -- only when sampling -- TemperatureLogitsWarper, TopKLogitsWarper, TopPLogitsWarper.  ``scores`` [B, V] fp32,
    ``generated`` [B, n] int64.  ``min_keep`` is the warpers' min_tokens_to_keep (2 under beam search: one EOS id + 1).
    Pure torch: pinned against the library's own processors in tests/test_host_logic.py."""
    if repetition_penalty != 1.0 and generated.shape[1] > 0:
        g = torch.gather(scores, 1, generated)
        g = torch.where(g < 0, g * repetition_penalty, g / repetition_penalty)
        scores = scores.scatter(1, generated, g)
    if no_repeat_ngram_size and no_repeat_ngram_size > 0:
        scores = ban_repeated_ngrams(scores, generated, int(no_repeat_ngram_size))
    if suppress_eos:
        scores = scores.clone()
        scores[:, eos_id] = -float("inf")
    if do_sample:
        scores = warp_logits(scores, temperature, top_k, top_p, min_keep)
    return scores


def ban_repeated_ngrams(scores: torch.Tensor, generated: torch.Tensor, n: int) -> torch.Tensor:
    """transformers' NoRepeatNGramLogitsProcessor (after the repetition penalty, ahead of MinLength in the library's processor
    list): a token that would complete an n-gram already present in the row's generated ids gets -inf.  ``generated`` [B, cur]
    (with inputs_embeds the prompt contributes no ids).  Host loop: rows x cur is a few hundred at most."""
    B, cur = generated.shape
    if cur + 1 < n:
        return scores
    scores = scores.clone()
    gen = generated.tolist()
    for b in range(B):
        row = gen[b]
        prefix = tuple(row[cur - (n - 1):]) if n > 1 else ()
        for i in range(cur - n + 1):
            if tuple(row[i:i + n - 1]) == prefix:
                scores[b, row[i + n - 1]] = -float("inf")
    return scores


def warp_logits(scores: torch.Tensor, temperature: Optional[float] = None, top_k: Optional[int] = None,
                top_p: Optional[float] = None, min_keep: int = 1) -> torch.Tensor:
    """transformers' sampling warpers in their order: TemperatureLogitsWarper, TopKLogitsWarper, TopPLogitsWarper."""
    if temperature is not None and temperature != 1.0:
        scores = scores / temperature
    if top_k is not None and top_k > 0:
        kth = torch.topk(scores, min(max(top_k, min_keep), scores.shape[-1]))[0][..., -1, None]
        scores = scores.masked_fill(scores < kth, -float("inf"))
    if top_p is not None and top_p < 1.0:
        srt, idx = torch.sort(scores, descending=False)
        cum = srt.softmax(dim=-1).cumsum(dim=-1)
        remove = cum <= (1 - top_p)
        remove[..., -min_keep:] = False
        scores = scores.masked_fill(remove.scatter(1, idx, remove), -float("inf"))
    return scores


def apply_eos_padding(ids: torch.Tensor, eos_id: int, pad_id: int) -> torch.Tensor:
    """HF greedy bookkeeping: after a row emits EOS it emits PAD; generation stops once every row is finished."""
    ids = ids.clone()
    B, n = ids.shape
    is_eos = ids == eos_id
    first = torch.where(is_eos.any(dim=1), is_eos.float().argmax(dim=1), torch.full((B,), n, device=ids.device))
    ar = torch.arange(n, device=ids.device)[None]
    ids[ar > first[:, None]] = pad_id
    keep = int(min(n, int(first.max().item()) + 1))
    return ids[:, :keep]


class GreedyState:
    """Device-resident state of the greedy token loop (see emu_llama_greedy_step)."""

    def __init__(self, eng: LlamaEngine, B: int, first_ids: torch.Tensor, next_pos: torch.Tensor, S: int,
                 kstart: torch.Tensor, out_ids: torch.Tensor):
        self.eng, self.B = eng, B
        dev = eng.device
        self.cur = first_ids.to(torch.int32).clone()
        self.pos = next_pos.to(torch.int32).clone()
        self.slot = torch.full((B,), S, device=dev, dtype=torch.int32)
        self.ctx = torch.tensor([S + 1], device=dev, dtype=torch.int32)
        self.step_idx = torch.tensor([1], device=dev, dtype=torch.int32)   # out_ids[0] already holds first_ids
        self.kstart = kstart
        self.out_ids = out_ids
        self.hidden = torch.empty(B, eng.cfg.hidden_size, device=dev, dtype=BF16)
        self.logits = torch.empty(B, eng.vocab, device=dev, dtype=BF16)
        self.ws = eng._workspace(B, 1)
        self.graph, self._epoch = None, -1

    def reset(self, first_ids: torch.Tensor, next_pos: torch.Tensor, S: int) -> None:
        """Rewind the device-side state to the end of the prompt, in place (a captured graph stays valid)."""
        self.cur.copy_(first_ids.to(torch.int32))
        self.pos.copy_(next_pos.to(torch.int32))
        self.slot.fill_(S)
        self.ctx.fill_(S + 1)
        self.step_idx.fill_(1)

    def step(self) -> None:
        e = self.eng
        check(lib().emu_llama_greedy_step(e.handle, self.B, self.cur.data_ptr(), self.pos.data_ptr(),
                                          self.slot.data_ptr(), self.kstart.data_ptr(), self.ctx.data_ptr(),
                                          self.step_idx.data_ptr(), self.out_ids.data_ptr(), e.s_max,
                                          self.hidden.data_ptr(), self.logits.data_ptr(), self.logits.stride(0),
                                          self.ws.data_ptr(), self.ws.numel(), ops.stream(e.device)),
              "emu_llama_greedy_step", e.ctx.handle)

    def step_graph(self) -> None:
        """Replay the step from a hipGraph captured on first use (launch-bound at TP>1: ~550 launches/token)."""
        if self.graph is None or self._epoch != self.eng.mode_epoch:
            self.step()                                   # warm-up outside capture (lazy module loads, weight table upload)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self.step()
            self.graph, self._epoch = g, self.eng.mode_epoch
            return                                        # capture does not execute: state advanced once by warm-up...
        self.graph.replay()

    def step_eager_suppress(self, suppress: int) -> None:
        """A step with a suppressed id (min_length > 1): same as step() but argmax masks ``suppress``."""
        e = self.eng
        ops.embed_gather(self.cur, e.embed, out=self.hidden)
        e.forward(self.hidden, self.B, 1, self.pos, self.slot, self.kstart, ctx=e.s_max, ctx_dev=self.ctx)
        e.logits(self.hidden, out=self.logits)
        ops.argmax(self.logits, suppress_id=suppress, out=self.cur)
        st = int(self.step_idx.item())
        self.out_ids[st] = self.cur
        self.pos += 1
        self.slot += 1
        self.ctx += 1
        self.step_idx += 1
