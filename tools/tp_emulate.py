#!/usr/bin/env python
"""Per-rank decode cost of a TP shard on ONE GPU: rank 0's 1/tp slice of LLaMA-33B with a 1-rank RCCL communicator in the
loop (the all-reduce launches are real, their cross-GPU latency is not).  With a third argument "p2p" the all-reduces are the
one-shot peer-to-peer kernel (csrc/p2p.hip) with one rank instead.  Usage: python tools/tp_emulate.py [tp] [steps] [p2p|rccl] [modes, e.g. 0,1,2]
(modes: emu_llama_set_decode_fused -- 0 launches, 1 fused layers cut at the all-reduces, 2 all-reduce inside the launch, 3 the all-reduce in the
tail of the o_proj / down_proj launches, 4 the persistent weight-streaming engine (csrc/decode_engine.hip); tp = 1: 0,1)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emu_amd import synth, ops
from emu_amd.conf.emu_conf import LlamaCfg
from emu_amd.llama import EmuHipContext, LlamaEngine, GreedyState

tp = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = torch.device("cuda", 0)
real = EmuHipContext(dev, 0, 1)
p2p = len(sys.argv) > 3 and sys.argv[3] == "p2p"
if p2p:
    real.init_tp(lambda b: b, force=True, allgather_bytes=lambda b: [b], rccl=False)
elif not (len(sys.argv) > 3 and sys.argv[3] == "none"):       # "none": no communicator at all (the tp = 1 decode, modes 0,1)
    real.init_tp(lambda b: b, force=True)
per_launch = int(os.environ.get("EMU_DL_PER", "0"))           # layers per fused launch (0 = all)
tail = os.environ.get("EMU_DECODE_TAIL") == "1"              # decode attention with the in-kernel split merge (one launch less per layer)


class ShardView:                      # the engine plans its shard from (tp_rank, tp_size); RCCL sees the 1-rank context
    def __init__(self, ctx, size):
        self.__dict__.update(ctx=ctx, tp_rank=0, tp_size=size)

    def __getattr__(self, k):
        return getattr(self.ctx, k)


l = LlamaCfg()
V = 32274
eng = LlamaEngine(l, V, ShardView(real, tp))
eng.load_weights(synth.iter_synth(synth.llama_param_shapes(l, V), seed=0, device=dev, dtype=torch.bfloat16))
if tail:
    eng.set_decode_tail(True)
S = 770
x = (torch.randn(1, S, l.hidden_size, device=dev) * 0.1).to(torch.bfloat16)
mask = torch.ones(1, S, dtype=torch.long)
modes = [int(m) for m in sys.argv[4].split(",")] if len(sys.argv) > 4 else [0, 1, 2, 3]
names = {0: "launches (8 per layer)", 1: "fused, cut at the all-reduces (4 launches per layer)", 2: "fused, all-reduce inside (1 launch per token)",
         3: "tail all-reduce (5 launches per layer: qkv | attention | o_proj + all-reduce | gate/up | down + all-reduce)",
         4: "persistent engine (3 launches per layer: attention | combine | o_proj -> all-reduce -> gate/up -> down -> all-reduce -> next qkv)"}
with torch.no_grad():
    hidden, kstart, next_pos = eng.prefill(x, mask, eng.kv_capacity(S + steps + 24))
    cur = ops.argmax(eng.logits(hidden[:, -1, :]), suppress_id=2)
    out = torch.zeros(steps + 16, 1, device=dev, dtype=torch.int32)
    first = None
    for mode in modes:
        eng.set_decode_fused(mode, per_launch)
        for graph in (True, False):
            st = GreedyState(eng, 1, cur, next_pos, S, kstart, out)
            fn = st.step_graph if graph else st.step
            for _ in range(4):
                fn()
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(steps):
                fn()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t) / steps * 1e3
            ids = out[: steps + 4, 0].tolist()
            first = first or ids
            print(f"tp={tp} shard on one GPU, {'p2p' if p2p else 'rccl'} all-reduce, {names[mode]}, {'hipGraph' if graph else 'eager'}: "
                  f"{ms:.3f} ms/token ({eng.weight_bytes_per_token() / 1e9:.2f} GB of weights per token per rank; ids "
                  f"{'match' if ids == first else 'DIFFER'}; give-ups {eng.decode_fused_stats()[0]}{'; decode tail merge' if tail else ''})", flush=True)
