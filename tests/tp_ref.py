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
