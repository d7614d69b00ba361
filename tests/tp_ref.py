"""Test helper: the engine's tensor-parallel data flow for one LLaMA layer restated in torch on CPU, operating on
the PACKED shard tensors produced by emu_amd.tp.ShardPlan (so the plan, the interleaving and the residual/
all-reduce placement are what is being tested).  Mirrors emu_llama_forward in emu_amd/csrc/engine.hip."""
import torch
import torch.nn.functional as F

from oracle import emu2_ref as R


def sharded_layer_partial(x, p, ln1, ln2, plan, cfg, cos, sin, mask, allreduce):
    """x [B,S,H] fp32 replicated; p = plan.pack_layer(...) of THIS rank; allreduce(t) sums over ranks."""
    B, S, H = x.shape
    Hl, D, Fl = plan.heads_local, plan.head_dim, plan.ffn_local
    h = R.rms_norm(x, ln1, cfg.rms_eps)
    qkv = F.linear(h, p["wqkv"]).view(B, S, 3, Hl, D)
    q, k, v = (qkv[:, :, i].transpose(1, 2) for i in range(3))
    q, k = R.apply_rope(q, k, cos, sin)
    s = (q @ k.transpose(2, 3)) * (D ** -0.5) + mask
    a = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B, S, Hl * D)
    part = F.linear(a, p["wo"])
    if plan.tp_rank == 0:
        part = part + x                       # the residual enters the sum exactly once
    x = allreduce(part)
    h = R.rms_norm(x, ln2, cfg.rms_eps)
    gu = F.linear(h, p["wgu"])
    act = F.silu(gu[..., 0::2]) * gu[..., 1::2]
    part = F.linear(act, p["wdown"])
    if plan.tp_rank == 0:
        part = part + x
    return allreduce(part)


def sharded_two_lane_prefill(x, layers, plan, cfg, cos, sin, allreduce, Ma):
    """The two-lane tensor-parallel prefill of emu_amd/csrc/engine.hip::llama_prefill_overlapped restated in torch: one unpadded
    batch element whose rows [0, Ma) (lane A) and [Ma, S) (lane B) walk all layers as separate chains; B attends A's keys / values
    of the same layer (the cache) plus its own, A never sees B.  ``layers``: [(packed shard tensors, ln1, ln2)].  The all-reduces are
    issued in the engine's host order -- per layer o_proj of A, of B, then down_proj of A, of B -- which is what keeps the ranks'
    collectives matched and lets each one wait only for the one before it."""
    B, S, H = x.shape
    assert B == 1 and 0 < Ma < S
    Hl, D = plan.heads_local, plan.head_dim
    out = x.clone()
    halves = ((0, Ma), (Ma, S))
    for p, ln1, ln2 in layers:
        kv_a, mid = None, [None, None]
        for i, (r0, r1) in enumerate(halves):                                  # attention stage of A, then of B
            xh = out[:, r0:r1]
            h = R.rms_norm(xh, ln1, cfg.rms_eps)
            qkv = F.linear(h, p["wqkv"]).view(1, r1 - r0, 3, Hl, D)
            q, k, v = (qkv[:, :, j].transpose(1, 2) for j in range(3))
            q, k = R.apply_rope(q, k, cos[:, r0:r1], sin[:, r0:r1])
            if r0 == 0:
                kv_a = (k, v)                                              # what lane A leaves in the cache for lane B
            else:
                k, v = torch.cat([kv_a[0], k], dim=2), torch.cat([kv_a[1], v], dim=2)
            qi = torch.arange(r0, r1)[:, None]
            kj = torch.arange(0, r1)[None, :]
            mask = torch.zeros(r1 - r0, r1, dtype=x.dtype).masked_fill(kj > qi, torch.finfo(x.dtype).min)
            s = (q @ k.transpose(2, 3)) * (D ** -0.5) + mask
            a = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(1, r1 - r0, Hl * D)
            part = F.linear(a, p["wo"])
            if plan.tp_rank == 0:
                part = part + xh
            mid[i] = allreduce(part)
        for i, (r0, r1) in enumerate(halves):                                  # MLP stage of A, then of B
            h = R.rms_norm(mid[i], ln2, cfg.rms_eps)
            gu = F.linear(h, p["wgu"])
            part = F.linear(F.silu(gu[..., 0::2]) * gu[..., 1::2], p["wdown"])
            if plan.tp_rank == 0:
                part = part + mid[i]
            out[:, r0:r1] = allreduce(part)
    return out
