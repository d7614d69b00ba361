#!/usr/bin/env python
"""Time the LLaMA-33B prefill alone (synthetic weights of the true shape): python tools/prefill_time.py [S] [reps] [--graph]
(--graph: the same call replayed from a hipGraph: what the eager launches of ~900 kernels cost beyond their kernel time)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emu_amd import synth
from emu_amd.conf.emu_conf import LlamaCfg
from emu_amd.llama import EmuHipContext, LlamaEngine

graph = "--graph" in sys.argv
argv = [a for a in sys.argv[1:] if a != "--graph"]
S = int(argv[0]) if len(argv) > 0 else 770
reps = int(argv[1]) if len(argv) > 1 else 5
dev = torch.device("cuda", 0)
l = LlamaCfg()
V = 32274
eng = LlamaEngine(l, V, EmuHipContext(dev))
eng.load_weights(synth.iter_synth(synth.llama_param_shapes(l, V), seed=0, device=dev, dtype=torch.bfloat16))
if os.environ.get("EMU_TUNE"):                        # A/B switches of single dispatch decisions (include/emu_hip.h: emu_gemm_tune)
    from emu_amd._lib import lib
    lib().emu_gemm_tune(int(os.environ["EMU_TUNE"]))
if os.environ.get("EMU_PREFILL_FUSION") == "0":      # A/B: the rope_kv + transpose_v launches instead of the qkv epilogue
    eng.set_prefill_fusion(False)
x = (torch.randn(1, S, l.hidden_size, device=dev) * 0.1).to(torch.bfloat16)
mask = torch.ones(1, S, dtype=torch.long, device=dev)
from emu_amd._lib import lib
tunes = [int(t) for t in os.environ["EMU_TUNES"].split(",")] if os.environ.get("EMU_TUNES") else [int(os.environ.get("EMU_TUNE", "0"))]
for tune in tunes:                                     # EMU_TUNES=0,65536,0,65536: same-run A/B of emu_gemm_tune masks
    lib().emu_gemm_tune(tune)
    ts = []
    with torch.no_grad():
        cap = eng.kv_capacity(S + 64)
        run = lambda: eng.prefill(x, mask, cap)
        if graph:
            run(); torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = eng.prefill(x, mask, cap)
            run = g.replay
        for i in range(reps + 2):
            torch.cuda.synchronize(); t = time.perf_counter()
            run()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t) * 1e3)
    print(f"prefill S={S}{' (hipGraph replay)' if graph else ''}{'' if eng.prefill_fusion else ' (rope_kv + transpose_v launches)'}: min {min(ts[2:]):.2f} ms  "
          f"median {sorted(ts[2:])[len(ts[2:]) // 2]:.2f} ms  [tune {tune}]", flush=True)
