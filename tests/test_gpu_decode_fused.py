"""GPU: whole decoder layers per launch (csrc/decode_layer.hip) against the launches they replace -- bit for bit.

The fused path keeps every workgroup's rows and summation order, so hidden states, KV caches and token ids must be IDENTICAL to
the multi-launch step (emu_llama_forward with the fusion off), for the block forms (LLaMA-33B widths), the wave forms (a TP = 8
shard's o_proj / down_proj), across a 128-key split boundary, replayed from a hipGraph, cut into several launches, and with the
all-reduce inside the launch (one-rank comm block: the protocol runs, the peers are this rank).
Replaces: the per-layer module loop reached from Emu2/emu/emu.py:133-138, :213-229; Emu2/emu/mixin.py:44-81."""
import pytest
import torch

pytestmark = pytest.mark.gpu

BF16 = torch.bfloat16


def _engine(cfg, vocab, tp=1, p2p=False):
    from emu_amd import synth
    from emu_amd.llama import EmuHipContext, LlamaEngine
    dev = torch.device("cuda", 0)
    real = EmuHipContext(dev, 0, 1)
    ctx = real
    if tp > 1:
        if p2p:
            real.init_tp(lambda b: b, force=True, allgather_bytes=lambda b: [b], rccl=False)

        class ShardView:                      # rank 0's 1/tp shard; the comm block sees the 1-rank context
            def __init__(self, c, size):
                self.__dict__.update(ctx=c, tp_rank=0, tp_size=size)

            def __getattr__(self, k):
                return getattr(self.ctx, k)
        ctx = ShardView(real, tp)
    eng = LlamaEngine(cfg, vocab, ctx)
    eng.load_weights(synth.iter_synth(synth.llama_param_shapes(cfg, vocab), seed=3, device=dev, dtype=BF16))
    return eng


def _run(eng, mode, S, steps, per_launch=0, graph=False, x_seed=0):
    """prefill S random rows, then `steps` greedy steps; returns (ids, hidden after every step, K cache, V cache)."""
    from emu_amd import ops
    from emu_amd.llama import GreedyState
    dev = eng.device
    eng.set_decode_fused(mode, per_launch)
    g = torch.Generator(device="cpu").manual_seed(x_seed)
    x = (torch.randn(1, S, eng.cfg.hidden_size, generator=g) * 0.1).to(BF16).to(dev)
    mask = torch.ones(1, S, dtype=torch.long)
    hidden, kstart, next_pos = eng.prefill(x, mask, eng.kv_capacity(S + steps + 8))
    cur = ops.argmax(eng.logits(hidden[:, -1, :].contiguous()), suppress_id=2)
    out = torch.zeros(steps + 2, 1, device=dev, dtype=torch.int32)
    st = GreedyState(eng, 1, cur, next_pos, S, kstart, out)
    hs = []
    for _ in range(steps):
        (st.step_graph if graph else st.step)()
        hs.append(st.hidden.clone())
    torch.cuda.synchronize()
    return out.clone(), torch.stack(hs), eng.kcache.clone(), eng.vcache.clone()


def _same(a, b):
    for x, y in zip(a, b):
        assert torch.equal(x, y), float((x.float() - y.float()).abs().max())


CFGS = {
    # block forms everywhere (o_proj K = 2048 > 1024)
    "block": dict(hidden_size=2048, intermediate_size=4096, num_attention_heads=16, num_hidden_layers=3),
    # wave forms for o_proj / down_proj (HD = 1024, ffn 2048 <= 2560, H = 1024)
    "wave": dict(hidden_size=1024, intermediate_size=2048, num_attention_heads=8, num_hidden_layers=3),
}


@pytest.mark.parametrize("name", ["block", "wave"])
def test_fused_equals_launches(name):
    from emu_amd.conf.emu_conf import LlamaCfg
    cfg = LlamaCfg(**CFGS[name])
    eng = _engine(cfg, 1024)
    # 250 + 10 steps: the new token's slot crosses the 256-key boundary (a third split goes live mid-run)
    ref = _run(eng, 0, 250, 10)
    g0, f0 = eng.decode_fused_stats()
    got = _run(eng, 1, 250, 10)
    g1, f1 = eng.decode_fused_stats()
    assert f0 == 0 or f1 > f0
    assert f1 - f0 == 10 and g1 == 0, (f0, f1, g1)
    _same(ref, got)
    _same(ref, _run(eng, 1, 250, 10, per_launch=1))
    _same(ref, _run(eng, 1, 250, 10, per_launch=2))
    _same(ref, _run(eng, 1, 250, 10, graph=True))
    assert eng.decode_fused_stats()[0] == 0


def test_fused_equals_launches_llama33b_width():
    """Two layers at the true LLaMA-33B width (6656 / 52 heads / 17920): the head-form qkv / gate-up, PRE-4 o_proj and the
    single-round-trip down_proj launches against their roles."""
    from emu_amd.conf.emu_conf import LlamaCfg
    cfg = LlamaCfg(num_hidden_layers=2)
    eng = _engine(cfg, 2048)
    ref = _run(eng, 0, 380, 6)
    got = _run(eng, 1, 380, 6)
    _same(ref, got)
    _same(ref, _run(eng, 1, 380, 6, graph=True))
    assert eng.decode_fused_stats()[0] == 0


@pytest.mark.parametrize("tp", [2, 8])
def test_fused_tp_shard_modes(tp):
    """Rank 0's shard of a TP decoder with a one-rank P2P comm block: launches + all-reduce kernel (mode 0), layers cut at the
    all-reduces (1), the all-reduce inside the launch (2) and in the tail of single-role launches (3) give the same bits; TP = 8 takes the wave forms (7 heads, K = 896)."""
    from emu_amd.conf.emu_conf import LlamaCfg
    cfg = LlamaCfg(num_hidden_layers=2)
    eng = _engine(cfg, 2048, tp=tp, p2p=True)
    ref = _run(eng, 0, 200, 5)
    _same(ref, _run(eng, 1, 200, 5))
    f0 = eng.decode_fused_stats()[1]
    _same(ref, _run(eng, 2, 200, 5))
    g, f1 = eng.decode_fused_stats()
    assert g == 0 and f1 - f0 == 5
    _same(ref, _run(eng, 2, 200, 5, graph=True))
    # mode 3: the all-reduce in the tail of the single-role o_proj / down_proj launches, the split merge in the attention launch
    f2 = eng.decode_fused_stats()[1]
    _same(ref, _run(eng, 3, 200, 5))
    _same(ref, _run(eng, 3, 200, 5, graph=True))
    g, f3 = eng.decode_fused_stats()
    assert g == 0 and f3 > f2
    eng.ctx.check_p2p()


@pytest.mark.parametrize("tp", [1, 8])
def test_merged_o_proj_equals_combine_launch(tp):
    """A short attention output (<= 1024 values: a TP = 8 rank's 7 heads, or the 8-head "wave" configuration) lets the o_proj launch merge
    the decode attention's splits in its own prologue (csrc/gemv_merge.hip, opt-in: emu_gemm_tune bit 19) instead of a combine launch
    ahead of it: same bits as the two launches, across a 128-key split boundary (contexts 250 ... 262), eager and replayed from a hipGraph."""
    from emu_amd._lib import lib
    from emu_amd.conf.emu_conf import LlamaCfg
    cfg = LlamaCfg(num_hidden_layers=2) if tp > 1 else LlamaCfg(**CFGS["wave"])
    eng = _engine(cfg, 2048, tp=tp, p2p=tp > 1)
    try:
        lib().emu_gemm_tune(0)
        ref = _run(eng, 0, 250, 12)
        lib().emu_gemm_tune(1 << 19)
        _same(ref, _run(eng, 0, 250, 12))
        _same(ref, _run(eng, 0, 250, 12, graph=True))
    finally:
        lib().emu_gemm_tune(0)
    if tp > 1:
        eng.ctx.check_p2p()


def test_mode_switch_drops_graphs():
    """a captured decode graph must not survive a mode switch (fp8 weights, decode tail, fused layers)."""
    from emu_amd.conf.emu_conf import LlamaCfg
    from emu_amd.llama import GreedyState
    cfg = LlamaCfg(**CFGS["block"])
    eng = _engine(cfg, 1024)
    e0 = eng.mode_epoch
    eng.set_decode_tail(True)
    eng.set_decode_tail(False)
    eng.set_decode_fused(0)
    assert eng.mode_epoch == e0 + 3
    ref = _run(eng, 0, 100, 4)
    # one GreedyState across a switch: the graph is re-captured under the new mode
    from emu_amd import ops
    dev = eng.device
    g = torch.Generator(device="cpu").manual_seed(0)
    x = (torch.randn(1, 100, cfg.hidden_size, generator=g) * 0.1).to(BF16).to(dev)
    hidden, kstart, next_pos = eng.prefill(x, torch.ones(1, 100, dtype=torch.long), eng.kv_capacity(100 + 4 + 8))
    cur = ops.argmax(eng.logits(hidden[:, -1, :].contiguous()), suppress_id=2)
    out = torch.zeros(6, 1, device=dev, dtype=torch.int32)
    st = GreedyState(eng, 1, cur, next_pos, 100, kstart, out)
    st.step_graph(); st.step_graph()
    f0 = eng.decode_fused_stats()[1]
    eng.set_decode_fused(1)
    st.step_graph(); st.step_graph()
    torch.cuda.synchronize()
    assert eng.decode_fused_stats()[1] > f0          # the later steps ran fused (warm-up step of the re-capture at least)
    assert torch.equal(out, ref[0])


def test_release_kv_frees_and_reallocates():
    """the persistent main / beam caches can be handed back (LlamaEngine.release_kv); the next prefill allocates anew."""
    from emu_amd.conf.emu_conf import LlamaCfg
    cfg = LlamaCfg(**CFGS["wave"])
    eng = _engine(cfg, 1024)
    a = _run(eng, 0, 100, 3)
    assert eng.kcache is not None
    eng.release_kv()
    assert eng.kcache is None and not eng.__dict__.get("_kv_slots")
    b = _run(eng, 0, 100, 3)
    _same(a, b)
