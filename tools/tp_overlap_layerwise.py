#!/usr/bin/env python
"""Two-lane vs serial tensor-parallel prefill, LAYER BY LAYER at full depth on a common input (rank 0's shard of LLaMA-33B, 1-rank
communicator): layer l runs under both schedules on the serial run's input of that layer, so every layer's own difference is seen
without the amplification of the layers behind it (a schedule bug that only shows at depth -- other activation statistics, other
tile choices -- would stand out; rounding differences stay at the bf16 level everywhere).
Usage: python tools/tp_overlap_layerwise.py [tp] [S]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emu_amd import synth
from emu_amd.conf.emu_conf import LlamaCfg
from emu_amd.llama import EmuHipContext, LlamaEngine

tp = int(sys.argv[1]) if len(sys.argv) > 1 else 8
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1544
dev = torch.device("cuda", 0)
real = EmuHipContext(dev, 0, 1)
real.init_tp(lambda b: b, force=True)


class ShardView:
    def __init__(self, ctx, size):
        self.__dict__.update(ctx=ctx, tp_rank=0, tp_size=size)

    def __getattr__(self, k):
        return getattr(self.ctx, k)


l = LlamaCfg()
V = 32274
eng = LlamaEngine(l, V, ShardView(real, tp))
eng.load_weights(synth.iter_synth(synth.llama_param_shapes(l, V), seed=0, device=dev, dtype=torch.bfloat16))
x = (torch.randn(1, S, l.hidden_size, device=dev) * 0.1).to(torch.bfloat16)
mask = torch.ones(1, S, dtype=torch.long, device=dev)
rel = lambda a, b: float((a.float() - b.float()).norm() / b.float().norm())
rows = lambda a, b: float(((a.float() - b.float()).norm(dim=-1) / b.float().norm(dim=-1).clamp_min(1e-12)).max())
worst, worst_row, out = 0.0, 0.0, []
with torch.no_grad():
    cap = eng.kv_capacity(S + 64)
    h = x
    for i in range(l.num_hidden_layers):
        eng.set_layer_range(i, i + 1)
        eng.set_tp_overlap(0)
        hs = eng.prefill(h, mask, cap)[0].clone()
        ks = eng.kcache[i, :, :, :S].clone()
        eng.set_tp_overlap(1024)
        n0 = eng.tp_overlap_count()
        hl = eng.prefill(h, mask, cap)[0].clone()
        assert eng.tp_overlap_count() == n0 + 1
        r, rr, rk = rel(hl, hs), rows(hl, hs), rel(eng.kcache[i, :, :, :S], ks)
        worst, worst_row = max(worst, r), max(worst_row, rr)
        out.append(r)
        if i % 6 == 0 or r > 5e-3:
            print(f"layer {i:2d}: two-lane vs serial on a common input: rel-L2 {r:.2e}, worst row {rr:.2e}, K plane {rk:.2e}; |input| per element {float(h.float().pow(2).mean().sqrt()):.3f}", flush=True)
        h = hs
eng.set_layer_range(0, -1)
print(f"tp={tp} S={S}: per-layer rel-L2 over {l.num_hidden_layers} layers: max {worst:.2e}, median {sorted(out)[len(out) // 2]:.2e}; worst single row {worst_row:.2e}")
