#!/usr/bin/env python
"""Per-workgroup timeline of ONE fused decode launch (csrc/decode_layer.hip) from the -DEMU_TRACE twin library:
    python -m emu_amd.build --trace
    EMU_HIP_TOOLS=1 EMU_HIP_LIB=emu_amd/csrc/libemu_hip_trace.so python tools/decode_trace.py [tp] [layers] [mode]
Prints, per (layer, role): workgroups, first entry / last exit (us from the launch's first entry), median lifetime, median time
waiting for the input, median time after the input was ready -- where a layer's ~170 us go when its six launches are one."""
import ctypes as C, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emu_amd import synth, ops
from emu_amd._lib import lib
from emu_amd.conf.emu_conf import LlamaCfg
from emu_amd.llama import EmuHipContext, LlamaEngine, GreedyState

tp = int(sys.argv[1]) if len(sys.argv) > 1 else 1
layers = int(sys.argv[2]) if len(sys.argv) > 2 else 4
mode = int(sys.argv[3]) if len(sys.argv) > 3 else (2 if tp > 1 else 1)
assert lib().emu_gemm_trace_built(), "needs the -DEMU_TRACE library (EMU_HIP_TOOLS=1 EMU_HIP_LIB=.../libemu_hip_trace.so)"
dev = torch.device("cuda", 0)
real = EmuHipContext(dev, 0, 1)
ctx = real
if tp > 1:
    real.init_tp(lambda b: b, force=True, allgather_bytes=lambda b: [b], rccl=False)

    class ShardView:
        def __init__(self, c, size):
            self.__dict__.update(ctx=c, tp_rank=0, tp_size=size)

        def __getattr__(self, k):
            return getattr(self.ctx, k)
    ctx = ShardView(real, tp)
cfg = LlamaCfg(num_hidden_layers=layers)
V = 4096
eng = LlamaEngine(cfg, V, ctx)
eng.load_weights(synth.iter_synth(synth.llama_param_shapes(cfg, V), seed=0, device=dev, dtype=torch.bfloat16))
eng.set_decode_fused(mode)
S = 770
x = (torch.randn(1, S, cfg.hidden_size, device=dev) * 0.1).to(torch.bfloat16)
with torch.no_grad():
    hidden, kstart, next_pos = eng.prefill(x, torch.ones(1, S, dtype=torch.long), eng.kv_capacity(S + 40))
    cur = ops.argmax(eng.logits(hidden[:, -1, :].contiguous()), suppress_id=2)
    out = torch.zeros(64, 1, device=dev, dtype=torch.int32)
    st = GreedyState(eng, 1, cur, next_pos, S, kstart, out)
    for _ in range(6):
        st.step()
    buf = torch.zeros(4 * 40000 * layers, dtype=torch.int64, device=dev)
    lib().emu_llama_set_decode_trace(eng.handle, buf.data_ptr())
    st.step()
    torch.cuda.synchronize()
    lib().emu_llama_set_decode_trace(eng.handle, None)
t = buf.cpu().numpy().reshape(-1, 4)
t = t[t[:, 3] != 0]
t0 = t[:, 1].min()
names = "q attn o gu down".split()
print(f"tp={tp} layers={layers} mode={mode}: {len(t)} workgroups, launch span {(t[:, 3].max() - t0) / 100:.1f} us (ticks of 10 ns)")
print(f"{'layer':>5} {'role':>5} {'wgs':>6} {'first in':>9} {'last in':>9} {'first rdy':>9} {'last out':>9} {'life med':>9} {'wait med':>9} {'wait max':>9} {'work med':>9}")
for l in sorted(set(t[:, 0] >> 8)):
    for r in range(5):
        m = t[(t[:, 0] >> 8 == l) & (t[:, 0] & 255 == r)]
        if not len(m):
            continue
        ent, rdy, ext = m[:, 1] - t0, m[:, 2] - t0, m[:, 3] - t0
        waited = m[:, 2] != 0
        rdy_eff = np.where(waited, rdy, ent)
        print(f"{l:5d} {names[r]:>5} {len(m):6d} {ent.min() / 100:9.1f} {ent.max() / 100:9.1f} {rdy_eff.min() / 100:9.1f} {ext.max() / 100:9.1f} "
              f"{np.median(ext - ent) / 100:9.2f} {np.median(rdy_eff - ent) / 100:9.2f} {(rdy_eff - ent).max() / 100:9.2f} {np.median(ext - rdy_eff) / 100:9.2f}")
