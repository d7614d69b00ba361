#!/usr/bin/env python
"""Where a 5-beam decode step spends its time at the true LLaMA-33B shape (debug aid).  Usage: python tools/beam_time.py [layers]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emu_amd import synth, ops
from emu_amd.conf.emu_conf import LlamaCfg
from emu_amd.llama import EmuHipContext, LlamaEngine

L = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda", 0)
l = LlamaCfg(num_hidden_layers=L)
V = 32274
eng = LlamaEngine(l, V, EmuHipContext(dev))
eng.load_weights(synth.iter_synth(synth.llama_param_shapes(l, V), seed=0, device=dev, dtype=torch.bfloat16))
S, nb = 770, 5
x = (torch.randn(1, S, l.hidden_size, device=dev) * 0.1).to(torch.bfloat16)
mask = torch.ones(1, S, dtype=torch.long)


def t(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


eng.alloc_kv(nb, 1024)
hid = (torch.randn(nb, l.hidden_size, device=dev) * 0.1).to(torch.bfloat16)
pos = torch.full((nb,), 800, device=dev, dtype=torch.int32)
slot = torch.full((nb,), 800, device=dev, dtype=torch.int32)
ks = torch.zeros(nb, device=dev, dtype=torch.int32)
print("forward M=5 (ms):", t(lambda: eng.forward(hid.clone(), nb, 1, pos, slot, ks, ctx=801)))
print("forward M=1 (ms):", t(lambda: eng.forward(hid[:1].clone(), 1, 1, pos[:1], slot[:1], ks[:1], ctx=801)) if False else "")
print("logits M=5 (ms):", t(lambda: eng.logits(hid)))
lp = torch.randn(1, nb, V, device=dev)
def host():
    a = torch.log_softmax(lp, dim=-1)
    acc = (a + torch.zeros(1, nb, 1, device=dev)).reshape(1, nb * V)
    torch.topk(acc, k=2 * nb)
print("host softmax+topk (ms):", t(host))
print("beam call 24 tokens (ms):", t(lambda: eng.beam_search_generate(x, mask, nb, 24, min_len=24), n=2))
print("prefill (ms):", t(lambda: eng.prefill(x, mask, 1024), n=3))
