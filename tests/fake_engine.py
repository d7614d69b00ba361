"""CPU stand-in for emu_amd.llama.LlamaEngine that runs the oracle arithmetic, so the product's HOST logic (beam search
bookkeeping, cache replication / re-ordering, position handling) can be tested without a GPU.  Only the methods the
host logic calls are provided; kcache/vcache are plain tensors laid out like the product's [L, B, H, S_max, D]."""
import torch

from emu_amd.conf.emu_conf import LlamaCfg
from emu_amd.llama import LlamaEngine
from oracle import emu2_ref as R


class _NoComm:                                       # the product checks its communicator after a generation
    def check_p2p(self):
        pass


class FakeEngine:
    def __init__(self, lcfg: LlamaCfg, vocab: int, W, rcfg: R.LlamaCfg, dtype=torch.float32):
        """``dtype`` = the arithmetic of the stand-in: fp32 (default) or bf16 with bf16 weights (how the reference runs Emu1)."""
        self.cfg, self.vocab, self.W, self.rcfg, self.dtype = lcfg, vocab, W, rcfg, dtype
        self.device = torch.device("cpu")
        self.embed = W["decoder.lm.model.embed_tokens.weight"]
        self.kcache = self.vcache = None
        self.kv_batch = self.s_max = 0
        self.ctx = _NoComm()

    KV_BUCKETS = LlamaEngine.KV_BUCKETS
    kv_capacity = LlamaEngine.kv_capacity            # the product's own bucket logic
    KV_SHARE_MAX = LlamaEngine.KV_SHARE_MAX
    fan_out_kv = LlamaEngine.fan_out_kv              # ... and its own cache fan-out for beams / candidates

    def set_kv_share(self, rows_per_prompt, shared_slots):
        """Like the device engine: groups of rows keep slots [0, shared_slots) in the group's first row only."""
        self.share = (int(rows_per_prompt), int(shared_slots)) if rows_per_prompt > 1 else (0, 0)

    def _rows(self, cache_l, B, n_past):
        """Layer cache rows [B, H, n_past, D] as the attention sees them (shared slots come from the group's first row)."""
        out = cache_l[:B, :, :n_past].clone()
        n, sh = getattr(self, "share", (0, 0))
        if n > 1:
            first = (torch.arange(B) // n) * n
            out[:, :, :min(sh, n_past)] = cache_l[first, :, :min(sh, n_past)]
        return out

    def alloc_kv(self, batch, s_max, zero=True, which="main"):
        L, H, D = self.rcfg.layers, self.rcfg.heads, self.rcfg.head_dim
        # zero=False (the product's fan-out allocates without a fill): NaN here, so a read of a never-written slot cannot pass
        make = torch.zeros if zero else (lambda *a, **k: torch.full(a, float("nan"), **k))
        self.kcache = make(L, batch, H, s_max, D, dtype=self.dtype)
        self.vcache = make(L, batch, H, s_max, D, dtype=self.dtype)
        self.kv_batch, self.s_max = batch, s_max
        self.share = (0, 0)

    def _run(self, x, pos, n_past, kstart):
        """x [B,T,H]; writes k/v of the T new slots at [n_past, n_past+T)."""
        B, T, _ = x.shape
        cache = R.KVCache(self.rcfg.layers)
        if n_past > 0:
            for i in range(self.rcfg.layers):
                cache.k[i] = self._rows(self.kcache[i], B, n_past)
                cache.v[i] = self._rows(self.vcache[i], B, n_past)
        mask = (torch.arange(n_past + T)[None, :] >= kstart[:, None]).long()
        h = R.llama_model(x, mask, self.W, self.rcfg, position_ids=pos.long().view(B, T), cache=cache, final_norm=False)
        for i in range(self.rcfg.layers):
            self.kcache[i, :B, :, n_past:n_past + T] = cache.k[i][:, :, n_past:]
            self.vcache[i, :B, :, n_past:n_past + T] = cache.v[i][:, :, n_past:]
        return h

    def prefill(self, embeds, attention_mask, s_max=None, hf_generate_positions=True):
        B, S, _ = embeds.shape
        am = attention_mask.long()
        self.alloc_kv(B, s_max or self.cfg.max_position_embeddings)
        pos = (am.cumsum(-1) - 1).masked_fill(am == 0, 1) if hf_generate_positions else torch.arange(S)[None].expand(B, -1)
        kstart = (S - am.sum(1)).to(torch.int32)
        h = self._run(embeds.to(self.dtype), pos, 0, kstart)
        nxt = am.sum(1) if hf_generate_positions else torch.full((B,), S)
        return h, kstart, nxt.to(torch.int32)

    def forward(self, hidden, B, T, pos, slot, kstart, ctx, ctx_dev=None):
        assert T == 1
        h = self._run(hidden.to(self.dtype).view(B, 1, -1), pos, int(slot[0]), kstart)
        hidden.copy_(h.view(B, -1))
        return hidden

    def final_norm_rows(self, hidden):
        return R.rms_norm(hidden.to(self.dtype), self.W["decoder.lm.model.norm.weight"], self.rcfg.rms_eps)

    def logits(self, rows, out=None):
        h = R.rms_norm(rows.to(self.dtype), self.W["decoder.lm.model.norm.weight"], self.rcfg.rms_eps)
        return torch.nn.functional.linear(h, self.W["decoder.lm.lm_head.weight"])
