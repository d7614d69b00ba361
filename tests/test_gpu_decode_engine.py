"""GPU: the persistent weight-streaming engine (csrc/decode_engine.hip) against the launches it replaces -- bit for bit.

emu_gemv_chain_bf16 runs a chain of one-row projections in ONE launch: loader waves stream every op's weights through an LDS ring ahead
of the activations, op outputs cross the chip as 4-byte granules.  The consumers keep the launch kernels' column-to-lane assignment and
summation order (emu_gemv_partition mirrors launch_gemv's dispatch), so every output must be IDENTICAL to the same ops through
emu_linear_bf16 -- single projections of a tensor-parallel shard (fused RMSNorm, SwiGLU pairs, residuals, wave-form and block-form
partitions, ragged row counts per CU), chains with one and three hand-offs, hipGraph replay -- and emu_llama_set_decode_fused mode 4
(attention launches + one engine launch per layer with the tensor-parallel all-reduces inside, one-rank comm block: the protocol runs,
the peer is this rank) must reproduce the launch path's ids, hidden states and KV caches.  The path is opt-in (it measured slower:
profiles/r06_decode_engine_probe_trace_emulate.log) and needs the device to itself.
Replaces: the LlamaDecoderLayer linears reached from Emu2/emu/emu.py:133-138, :213-229 under the SURVEY 8e shard plan."""
import pytest
import torch

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16


def _ctx():
    from emu_amd.llama import EmuHipContext
    return EmuHipContext(torch.device("cuda", 0), 0, 1)


def _r(*shape, scale=1.0, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return (torch.randn(*shape, device="cuda", generator=g) * scale).to(BF16)


# (N, K, epi, fused norm): a TP = 8 / TP = 4 shard's projections, a ragged row count, short rows (nine rows per fill), a wave-form shape
SINGLE = [(2688, 6656, 0, True), (6656, 896, 1, False), (4480, 6656, 2, True), (6656, 2240, 1, False), (4992, 6656, 0, True),
          (6656, 1664, 1, False), (1000, 512, 0, False), (1026, 2560, 1, False), (3000, 4096, 2, True), (6656, 4480, 1, False)]


@pytest.mark.parametrize("N,K,epi,norm", SINGLE)
def test_single_projection_equals_launch(N, K, epi, norm):
    from emu_amd import ops
    ctx = _ctx()
    w, x = _r(N, K, scale=0.02, seed=1), _r(1, K, seed=2)
    g = (1 + 0.1 * torch.randn(K, device="cuda")).to(BF16) if norm else None
    res = _r(1, N, seed=3) if epi == ops.EPI_RESID else None
    want = ops.linear(x, w, res=res, norm_w=g, eps=1e-6, epi=epi)
    outs, err, _ = ops.gemv_chain(ctx.handle, [dict(w=w, x=x, gain=g, eps=1e-6, epi=epi, res=res)])
    torch.cuda.synchronize()
    assert int(err.item()) == 0
    assert torch.equal(outs[0], want), float((outs[0].float() - want.float()).abs().max())


def test_chains_equal_launches_and_replay_from_a_graph():
    """gate/up -> down (one hand-off) and o_proj -> gate/up -> down -> next qkv (three) of a TP = 8 shard, eager and replayed."""
    from emu_amd import ops
    ctx = _ctx()
    H, HD, Fl = 6656, 896, 2240
    wo, wg, wd, wq = _r(H, HD, scale=0.03, seed=4), _r(2 * Fl, H, scale=0.02, seed=5), _r(H, Fl, scale=0.02, seed=6), _r(3 * HD, H, scale=0.02, seed=7)
    xa, res0, g2, g1 = _r(1, HD, seed=8), _r(1, H, seed=9), (1 + 0.1 * torch.randn(H, device="cuda")).to(BF16), (1 + 0.1 * torch.randn(H, device="cuda")).to(BF16)
    hb = ops.linear(xa, wo, res=res0, epi=ops.EPI_RESID)
    act = ops.linear(hb, wg, norm_w=g2, eps=1e-6, epi=ops.EPI_SWIGLU)
    ha = ops.linear(act, wd, res=hb, epi=ops.EPI_RESID)
    qkv = ops.linear(ha, wq, norm_w=g1, eps=1e-6)
    two = [dict(w=wg, x=hb, gain=g2, eps=1e-6, epi=ops.EPI_SWIGLU), dict(w=wd, x=None, epi=ops.EPI_RESID, res=hb)]
    four = [dict(w=wo, x=xa, epi=ops.EPI_RESID, res=res0), dict(w=wg, x=None, gain=g2, eps=1e-6, epi=ops.EPI_SWIGLU),
            dict(w=wd, x=None, epi=ops.EPI_RESID, res=hb), dict(w=wq, x=None, gain=g1, eps=1e-6, epi=ops.EPI_NONE)]
    o2, err, _ = ops.gemv_chain(ctx.handle, two)
    o4, err4, _ = ops.gemv_chain(ctx.handle, four)
    torch.cuda.synchronize()
    assert int(err.item()) == 0 and int(err4.item()) == 0
    assert o2[0] is None and torch.equal(o2[1], ha)
    assert o4[:3] == [None, None, None] and torch.equal(o4[3], qkv)
    gr = torch.cuda.CUDAGraph()
    cap = torch.cuda.Stream()
    keep = []
    with torch.cuda.graph(gr, stream=cap):
        keep.append(ops.gemv_chain(ctx.handle, four, err=err4))
    for _ in range(3):
        keep[0][0][3].zero_()
        gr.replay()
        torch.cuda.synchronize()
        assert torch.equal(keep[0][0][3], qkv) and int(err4.item()) == 0


def test_shapes_outside_the_engine_are_refused():
    """Rows longer than 13 KiB (K > 6656) answer -95 (the caller keeps the launches); a chain whose op input is not the previous op's
    output is an argument error."""
    from emu_amd import ops
    from emu_amd._lib import EmuHipError
    ctx = _ctx()
    with pytest.raises(EmuHipError):
        ops.gemv_chain(ctx.handle, [dict(w=_r(6656, 8960, scale=0.02), x=_r(1, 8960), epi=ops.EPI_NONE)])
    with pytest.raises(EmuHipError):
        ops.gemv_chain(ctx.handle, [dict(w=_r(1024, 512, scale=0.02), x=_r(1, 512)), dict(w=_r(1024, 768, scale=0.02), x=None)])


def test_mode4_decode_equals_the_launch_path():
    """emu_llama_set_decode_fused(4) on a shard-shaped model with a one-rank comm block: ids, hidden states after every step and the KV
    caches of the launch path, eager and replayed from a hipGraph; the engine path is the one that ran (forward counter), nothing
    timed out."""
    from emu_amd.conf.emu_conf import LlamaCfg
    from tests.test_gpu_decode_fused import _engine, _run, _same
    cfg = LlamaCfg(hidden_size=4096, intermediate_size=4 * 2048, num_attention_heads=4 * 8, num_hidden_layers=3)
    eng = _engine(cfg, 1024, tp=4, p2p=True)                      # rank 0's quarter: 8 heads (K = 1024), ffn 2048, hidden 4096
    ref = _run(eng, 0, 250, 10)
    f0 = eng.decode_fused_stats()[1]
    got = _run(eng, 4, 250, 10)
    g1, f1 = eng.decode_fused_stats()
    assert g1 == 0 and f1 - f0 == 10, (g1, f1 - f0)                # every step took the engine path, no wait gave up
    _same(ref, got)
    got_g = _run(eng, 4, 250, 10, graph=True)
    assert eng.decode_fused_stats()[0] == 0
    _same(ref, got_g)
    eng.set_decode_fused(0)
