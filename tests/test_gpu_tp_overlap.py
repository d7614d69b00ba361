"""Tensor-parallel prefill as two concurrent lanes (the prompt's rows cut in two halves, each half's all-reduces behind the other
half's GEMMs: include/emu_hip.h: emu_llama_set_tp_overlap; replaces nothing in the reference, whose multi-GPU scheme is layer
placement -- Emu2/emu/mixin.py:44-81 -- and is what SURVEY 8e / north_star ask of the tensor-parallel design).

One process, one GPU: rank 0's shard of a 2-layer decoder at the TRUE LLaMA-33B width with a 1-rank communicator in the loop (every
all-reduce launch is real, its peers are not), the two-lane schedule against the serial one:
  * TP = 8 shard (7 of 56 heads: the qkv projection runs the unfused RoPE / KV append / V^T launches per lane),
  * TP = 2 shard (26 heads: the fused qkv epilogue per lane, the second lane's V^T = the first's keys out of the cache + its own),
  * a comm block (peer-to-peer all-reduces of both lanes through the context's all-reduce stream) and a 1-rank RCCL communicator
    (all-reduces inside each lane's stream).
The yardstick is the oracle: layer 0 of the shard restated in fp32 torch on the packed shard tensors (tests/tp_ref.py, the engine's
tensor-parallel data flow; oracle/emu2_ref.py arithmetic) -- the two-lane schedule must sit as close to it as the serial schedule does
(measured: both 0.0165-0.0167 for a shard's partial layer at TP = 8 and TP = 2; from each other 0.0003 at TP = 8 -- single roundings
moved by another K-slice order -- and 0.004 at TP = 2, where launches of other shapes pick other tile / attention-block configurations).  Over two layers of random-init weights those roundings
amplify like any bf16 noise does here, so the whole-stack bound between the schedules is the suite's usual 2e-2, with a per-row bound
that an indexing error (a wrong row offset, slot or V^T column) cannot meet.  The layer-0
K / V planes (RoPE, slots, append) agree to single roundings; the schedule is the one that ran (counter); its hipGraph replay
reproduces the eager run bit for bit.  Two rank processes sharing this GPU: tests/tp_overlap_worker.py."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BF16 = torch.bfloat16
REL1, REL, ROW = 8e-3, 2e-2, 6e-2     # rel-L2 between the schedules after one layer / the stack, and the worst single row of the stack


def _rel(a, b):
    a, b = a.float(), b.float()
    return float((a - b).norm() / b.norm().clamp_min(1e-12))


class _ShardView:                     # the engine plans its shard from (tp_rank, tp_size); the comm block is the 1-rank context's
    def __init__(self, ctx, size):
        self.__dict__.update(ctx=ctx, tp_rank=0, tp_size=size)

    def __getattr__(self, k):
        return getattr(self.ctx, k)


def _run(eng, x, mask, cap):
    hidden, kstart, next_pos = eng.prefill(x, mask, cap)
    lg = eng.logits(hidden[:, -1, :].contiguous())
    torch.cuda.synchronize()
    return hidden.clone(), eng.kcache.clone(), eng.vcache.clone(), lg.clone()


def _oracle_layer0(cfg, V, tp, x, dev):
    """Layer 0 of rank 0's shard in fp32 torch on the device (the 1-rank all-reduce is the identity): tests/tp_ref.py on the packed
    tensors of emu_amd.tp.ShardPlan, weights = the bf16 values the engine loaded."""
    from emu_amd import synth
    from emu_amd.tp import ShardPlan
    from oracle import emu2_ref as R
    from tests.tp_ref import sharded_layer_partial
    shapes = synth.llama_param_shapes(cfg, V)
    pre = "decoder.lm.model.layers.0."
    W = {k: synth.synth_tensor(k, shapes[k], 3, dev, BF16).float() for k in shapes if k.startswith(pre)}
    plan = ShardPlan(cfg.hidden_size, cfg.num_attention_heads, cfg.head_dim, cfg.intermediate_size, tp, 0)
    packed = plan.pack_layer(*(W[pre + k] for k in ("self_attn.q_proj.weight", "self_attn.k_proj.weight", "self_attn.v_proj.weight",
                                                     "self_attn.o_proj.weight", "mlp.gate_proj.weight", "mlp.up_proj.weight",
                                                     "mlp.down_proj.weight")))
    S = x.shape[1]
    cos, sin = (t.to(dev) for t in R.rope_cos_sin(torch.arange(S)[None], cfg.head_dim, cfg.rope_theta, torch.float32))
    mask = torch.zeros(1, 1, S, S, device=dev).masked_fill(torch.ones(S, S, device=dev, dtype=torch.bool).triu(1), torch.finfo(torch.float32).min)
    rcfg = R.LlamaCfg(hidden=cfg.hidden_size, heads=cfg.num_attention_heads, layers=1, ffn=cfg.intermediate_size, vocab=V)
    return sharded_layer_partial(x.float(), packed, W[pre + "input_layernorm.weight"], W[pre + "post_attention_layernorm.weight"],
                                 plan, rcfg, cos, sin, mask, lambda t: t)


def _row_rel(a, b):
    a, b = a.float().reshape(-1, a.shape[-1]), b.float().reshape(-1, b.shape[-1])
    return float(((a - b).norm(dim=1) / b.norm(dim=1).clamp_min(1e-12)).max())


@pytest.mark.parametrize("tp,S,comm", [(8, 1544, "p2p"), (2, 1100, "p2p"), (8, 1544, "rccl"), (2, 1300, "rccl")])
def test_two_half_schedule_matches_serial(tp, S, comm):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from emu_amd import synth
    from emu_amd.conf.emu_conf import LlamaCfg
    from emu_amd.llama import EmuHipContext, LlamaEngine
    dev = torch.device("cuda", 0)
    real = EmuHipContext(dev, 0, 1)
    if comm == "p2p":
        real.init_tp(lambda b: b, force=True, allgather_bytes=lambda b: [b], rccl=False)
    else:
        real.init_tp(lambda b: b, force=True)
    cfg = LlamaCfg(num_hidden_layers=2)
    V = 2048
    eng = LlamaEngine(cfg, V, _ShardView(real, tp))
    eng.load_weights(synth.iter_synth(synth.llama_param_shapes(cfg, V), seed=3, device=dev, dtype=BF16))
    assert eng.tp_overlap_rows == 0                       # opt-in since round 6 (EMU_TP_OVERLAP / set_tp_overlap)
    g = torch.Generator().manual_seed(7)
    x = (torch.randn(1, S, cfg.hidden_size, generator=g) * 0.1).to(BF16).to(dev)
    mask = torch.ones(1, S, dtype=torch.long, device=dev)          # on the device: the capture below must not copy from the host
    cap = eng.kv_capacity(S + 8)
    with torch.no_grad():
        eng.set_tp_overlap(0)
        n0 = eng.tp_overlap_count()
        h_ser, k_ser, v_ser, lg_ser = _run(eng, x, mask, cap)
        assert eng.tp_overlap_count() == n0               # serial schedule
        eng.set_tp_overlap(1024)
        # poison what the halves must overwrite themselves: the cache planes and the workspace (V^T, slices)
        eng.kcache.fill_(float("nan")); eng.vcache.fill_(float("nan")); eng._ws.fill_(0xFF)
        h_ov, k_ov, v_ov, lg_ov = _run(eng, x, mask, cap)
        assert eng.tp_overlap_count() == n0 + 1           # the two-half schedule ran
        assert torch.isfinite(h_ov.float()).all() and torch.isfinite(lg_ov.float()).all()
        assert _rel(h_ov, h_ser) < REL and _row_rel(h_ov, h_ser) < ROW, (_rel(h_ov, h_ser), _row_rel(h_ov, h_ser))
        # layer 0's K / V: common input, so only tile choices of the qkv GEMM differ (RoPE, slot and row offsets of both lanes)
        assert _rel(k_ov[0, :, :, :S], k_ser[0, :, :, :S]) < 2e-3 and _rel(v_ov[0, :, :, :S], v_ser[0, :, :, :S]) < 2e-3
        assert _row_rel(k_ov[0, :, :, :S], k_ser[0, :, :, :S]) < 2e-2 and _row_rel(v_ov[0, :, :, :S], v_ser[0, :, :, :S]) < 2e-2
        assert _rel(k_ov[:, :, :, :S], k_ser[:, :, :, :S]) < REL and _rel(v_ov[:, :, :, :S], v_ser[:, :, :, :S]) < REL
        assert _rel(lg_ov, lg_ser) < 3e-2
        # one layer alone: (attention over lane B's V^T, both all-reduces, the MLP) on a common input
        eng.set_layer_range(0, 1)
        eng.set_tp_overlap(0)
        h1_ser = eng.prefill(x, mask, cap)[0].clone()
        eng.set_tp_overlap(1024)
        h1_ov = eng.prefill(x, mask, cap)[0].clone()
        eng.set_layer_range(0, -1)
        assert _rel(h1_ov, h1_ser) < REL1 and _row_rel(h1_ov, h1_ser) < 2e-2, (_rel(h1_ov, h1_ser), _row_rel(h1_ov, h1_ser))
        n0 += 1
        ref1 = _oracle_layer0(cfg, V, tp, x, dev)
        r_ser, r_ov = _rel(h1_ser, ref1), _rel(h1_ov, ref1)
        print(f"tp{tp} S={S} {comm}: layer 0 vs the fp32 oracle: serial {r_ser:.4f}, two-lane {r_ov:.4f}; between the schedules {_rel(h1_ov, h1_ser):.4f}")
        assert r_ov < 2.5e-2 and r_ov < 1.1 * r_ser + 5e-4, (r_ov, r_ser)
        assert _row_rel(h1_ov, ref1) < 1.5 * _row_rel(h1_ser, ref1) + 1e-3
        del ref1
        # a prompt below the threshold keeps the serial schedule
        xs, ms = x[:, :600].contiguous(), mask[:, :600].contiguous()
        eng.prefill(xs, ms, cap)
        assert eng.tp_overlap_count() == n0 + 1
        # replayed from a hipGraph: the second stream joins the capture through the events, same bits as eager
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            hg, _, _ = eng.prefill(x, mask, cap)
        eng.kcache.fill_(float("nan")); eng.vcache.fill_(float("nan"))
        gr.replay()
        torch.cuda.synchronize()
        assert torch.equal(hg, h_ov) and torch.equal(eng.kcache[:, :, :, :S], k_ov[:, :, :, :S])
        assert torch.equal(eng.vcache[:, :, :, :S], v_ov[:, :, :, :S])
    if comm == "p2p":
        real.check_p2p()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def test_two_half_schedule_two_ranks_sharing_the_gpu():
    """Two rank processes on this device, every all-reduce through the peer-to-peer kernels on each rank's second stream: the
    sharded two-half prefill reproduces the unsharded engine's residual stream and the serial sharded schedule's greedy ids."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", EMU_TP_SHARED_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "tp_overlap_worker.py")]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "two-half prefill ok" in r.stdout, r.stdout[-3000:]
