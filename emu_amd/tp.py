"""Tensor-parallel shard plan of the LLaMA decoder (one process per GPU, RCCL all-reduce over xGMI).

New design: the reference's only multi-GPU scheme is whole-layer placement (Emu2/emu/chat.py:235-283,
Emu2/emu/mixin.py:14-85), which keeps one GPU busy at a time.  Here q/k/v and gate/up are column-sharded,
o_proj and down_proj row-sharded, and the two partial sums per layer are all-reduced.

LLaMA-33B has 52 heads, which 8 does not divide: heads are padded to ``heads_pad = ceil(H / tp) * tp``
(56 for tp=8); the extra heads have all-zero q/k/v rows and all-zero o_proj columns, so they contribute
exactly 0 to the output.  ffn (17920 = 8 * 2240) must divide evenly.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict

import torch


@dataclass(frozen=True)
class ShardPlan:
    hidden: int
    heads: int
    head_dim: int
    ffn: int
    tp_size: int
    tp_rank: int

    def __post_init__(self):
        if not (0 <= self.tp_rank < self.tp_size):
            raise ValueError("tp_rank out of range")
        if self.ffn % self.tp_size or (self.ffn // self.tp_size) % 8:
            raise ValueError(f"intermediate size {self.ffn} cannot be split into {self.tp_size} shards of a multiple of 8")

    @property
    def heads_pad(self) -> int:
        return -(-self.heads // self.tp_size) * self.tp_size

    @property
    def heads_local(self) -> int:
        return self.heads_pad // self.tp_size

    @property
    def ffn_local(self) -> int:
        return self.ffn // self.tp_size

    @property
    def head_range(self):
        """Global head indices [h0, h1) owned by this rank (indices >= heads are zero padding)."""
        h0 = self.tp_rank * self.heads_local
        return h0, h0 + self.heads_local

    @property
    def ffn_range(self):
        f0 = self.tp_rank * self.ffn_local
        return f0, f0 + self.ffn_local

    def vocab_range(self, vocab: int):
        """Rows [r0, r1) of lm_head this rank streams: ceil(vocab / tp) rows per rank (32 274 -> 4035 at TP = 8, the last rank 4029)."""
        per = -(-vocab // self.tp_size)
        r0 = min(self.tp_rank * per, vocab)
        return r0, min(r0 + per, vocab)

    # ---- packing ------------------------------------------------------------------------------------
    def _head_rows(self, w: torch.Tensor) -> torch.Tensor:
        """Rows of a [H*D, hidden] projection that belong to this rank's heads, zero rows for padding."""
        D = self.head_dim
        h0, h1 = self.head_range
        real = max(0, min(h1, self.heads) - h0)
        out = w.new_zeros(self.heads_local * D, w.shape[1])
        if real > 0:
            out[: real * D] = w[h0 * D:(h0 + real) * D]
        return out

    def pack_layer(self, q, k, v, o, gate, up, down) -> Dict[str, torch.Tensor]:
        """Reference tensors (transformers LlamaDecoderLayer names) -> packed shard tensors of emu_hip.h."""
        D = self.head_dim
        wqkv = torch.cat([self._head_rows(q), self._head_rows(k), self._head_rows(v)], dim=0).contiguous()
        h0, h1 = self.head_range
        real = max(0, min(h1, self.heads) - h0)
        wo = o.new_zeros(o.shape[0], self.heads_local * D)
        if real > 0:
            wo[:, : real * D] = o[:, h0 * D:(h0 + real) * D]
        f0, f1 = self.ffn_range
        wgu = torch.stack([gate[f0:f1], up[f0:f1]], dim=1).reshape(2 * self.ffn_local, gate.shape[1]).contiguous()
        wdown = down[:, f0:f1].contiguous()
        return {"wqkv": wqkv, "wo": wo.contiguous(), "wgu": wgu, "wdown": wdown}


def image_parallel_encode(images: torch.Tensor, encode, rank: int, world: int, all_gather):
    """Data-parallel ViT over the images of a prompt (BASELINE configs[2]: 4 images): instead of every
    tensor-parallel rank encoding all n images (the ViT is replicated: 8.7 GB), rank r encodes images r, r + world, ... and the
    pooled visual tokens are all-gathered -- zero intra-layer communication, one small exchange ([n, n_query, width] bf16, 3.4 MB
    per image at n_query 256) per prompt.  Replaces nothing in the reference (its multi-GPU scheme is layer placement,
    Emu2/emu/mixin.py:44-81); the single-GPU call (Emu2/emu/emu.py:199-203) is ``encode(images)``.

    ``encode``: [k, 3, H, W] -> [k, n_query, width]; ``all_gather``: tensor [per, n_query, width] -> list of ``world`` such
    tensors in rank order (torch.distributed.all_gather semantics).  Every rank runs the same number of images (the last ones
    are padded with a repeat of the final image and dropped after the gather), so the collective is uniform.  Returns the
    tokens of all n images in their original order, identical on every rank."""
    n = images.shape[0]
    if world <= 1 or n < 2:
        return encode(images)
    per = (n + world - 1) // world
    idx = [min(rank + j * world, n - 1) for j in range(per)]         # round-robin: image i lives on rank i % world
    local = encode(images[idx].contiguous())
    parts = all_gather(local.contiguous())                              # world x [per, n_query, width]
    out = [None] * n
    for r in range(world):
        for j in range(per):
            i = r + j * world
            if i < n:
                out[i] = parts[r][j]
    return torch.stack(out, dim=0)


class CfgPair:
    """The two ranks that split the UNet's classifier-free-guidance pair (Emu2/emu/diffusion.py:131-145): ranks 0 and 1
    of the default process group.  ``half``: 0 (cond) / 1 (uncond) on those ranks, None elsewhere (such ranks keep running a full
    replica).  Every rank of the default group must construct it (``dist.new_group`` is collective).  Under gloo (ranks sharing
    one GPU in the validation runs) the exchanges go through the host; under nccl (= RCCL) they stay on the device."""

    def __init__(self):
        import torch.distributed as dist
        self.dist = dist
        self.rank = dist.get_rank()
        self.group = dist.new_group([0, 1])
        self.half = self.rank if self.rank < 2 else None
        self.host = dist.get_backend() == "gloo"

    def all_gather(self, t: torch.Tensor):
        """[cond part, uncond part] of a device tensor held by each of the two ranks."""
        src = t.cpu() if self.host else t
        parts = [torch.empty_like(src) for _ in range(2)]
        self.dist.all_gather(parts, src, group=self.group)
        return [p.to(t.device) for p in parts] if self.host else parts

    def broadcast(self, t: torch.Tensor) -> torch.Tensor:
        """rank 0's tensor on both ranks (the initial latents: every rank draws from its own generator)."""
        if self.host:
            c = t.cpu()
            self.dist.broadcast(c, src=0, group=self.group)
            return c.to(t.device)
        self.dist.broadcast(t, src=0, group=self.group)
        return t
