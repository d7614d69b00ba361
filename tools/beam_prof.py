#!/usr/bin/env python
"""Where a 5-beam step's time goes (the reference's default decoding mode, Emu2/emu/emu.py:163-172): the whole
beam_search_generate call against (a) the bare 5-row model steps it contains, enqueued back to back with no bookkeeping, and
(b) the prefill alone.  Full-size decoder, synthetic weights.  Usage: python tools/beam_prof.py [new_tokens] [S]
Run under `rocprofv3 --kernel-trace --stats` for the per-kernel split."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emu_amd import LlamaCfg, synth, ops
from emu_amd.constants import VOCAB_EMU2_CHAT
from emu_amd.llama import EmuHipContext, LlamaEngine

n_new = int(sys.argv[1]) if len(sys.argv) > 1 else 24
S = int(sys.argv[2]) if len(sys.argv) > 2 else 770
layers = int(os.environ.get("EMU_PROF_LAYERS", "60"))
nb = 5
dev = torch.device("cuda", 0)
lcfg = LlamaCfg(num_hidden_layers=layers)
lm = LlamaEngine(lcfg, VOCAB_EMU2_CHAT, EmuHipContext(dev))
lm.load_weights(synth.iter_synth(synth.llama_param_shapes(lcfg, VOCAB_EMU2_CHAT), seed=0, device=dev, dtype=torch.bfloat16))
x = (torch.randn(1, S, lcfg.hidden_size, device=dev) * 0.02).to(torch.bfloat16)
mask = torch.ones(1, S, dtype=torch.long)


def timed(fn):
    torch.cuda.synchronize(); t = time.perf_counter()
    r = fn()
    torch.cuda.synchronize()
    return time.perf_counter() - t, r


with torch.no_grad():
    lm.beam_search_generate(x, mask, nb, 2, min_len=2)
    for rep in range(2):
        t_call, out = timed(lambda: lm.beam_search_generate(x, mask, nb, n_new, min_len=n_new))
        t_pf, _ = timed(lambda: lm.prefill(x, mask, lm.kv_capacity(S + n_new)))
        steps = out.shape[1] - 1
        # the bare model steps: same cache geometry, 5 rows, no bookkeeping, no host sync
        lm.alloc_kv(nb, lm.kv_capacity(S + n_new))
        hid = torch.empty(nb, lcfg.hidden_size, device=dev, dtype=torch.bfloat16)
        toks = torch.randint(0, 32000, (nb,), device=dev, dtype=torch.int32)
        kstart = torch.zeros(nb, device=dev, dtype=torch.int32)
        pos = torch.full((nb,), S, device=dev, dtype=torch.int32)

        def bare():
            p = pos
            for i in range(steps):
                ctx = S + i
                ops.embed_gather(toks, lm.embed, out=hid)
                slot = torch.full((nb,), ctx, device=dev, dtype=torch.int32)
                lm.forward(hid, nb, 1, p, slot, kstart, ctx=ctx + 1)
                p = p + 1
                lm.logits(hid)
        t_bare, _ = timed(bare)
        # the model part of the real call, by events around forward + logits (the rest of a step is beam bookkeeping)
        ev = []
        fwd, lg = lm.forward, lm.logits

        def fwd_t(*a_, **k_):
            e0 = torch.cuda.Event(enable_timing=True); e0.record()
            r = fwd(*a_, **k_)
            ev.append([e0, None])
            return r

        def lg_t(*a_, **k_):
            r = lg(*a_, **k_)
            if ev and ev[-1][1] is None:
                e1 = torch.cuda.Event(enable_timing=True); e1.record()
                ev[-1][1] = e1
            return r
        lm.forward, lm.logits = fwd_t, lg_t
        t_call2, _ = timed(lambda: lm.beam_search_generate(x, mask, nb, n_new, min_len=n_new))
        lm.forward, lm.logits = fwd, lg
        model_ms = sum(e0.elapsed_time(e1) for e0, e1 in ev[1:] if e1 is not None)          # ev[0] is the prefill
        print(f"rep {rep}: instrumented call {t_call2*1e3:.1f} ms: model part of the {len(ev)-1} steps {model_ms/max(1,len(ev)-1):.2f} ms/step", flush=True)
        print(f"rep {rep}: call {t_call*1e3:.1f} ms, prefill {t_pf*1e3:.1f} ms, {steps} steps: {(t_call-t_pf)/steps*1e3:.2f} ms/step "
              f"of which bare 5-row model step {t_bare/steps*1e3:.2f} ms", flush=True)
