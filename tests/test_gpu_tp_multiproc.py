"""Tensor parallelism on real GPUs: the ENGINE (not a torch restatement) sharded over 2 / 4 ranks, one process per GPU over
RCCL, must generate the golden fixture's greedy ids (the real reference's) with eager launches and under hipGraph replay.
The multi-GPU cases need >= 2 GPUs on the node and skip on the 1-GPU runner; the shared-GPU cases run there: 2 / 4 rank
processes on ONE device, every all-reduce through the one-shot peer-to-peer path over HIP IPC (RCCL refuses ranks that share
a device), which is the engine's real TP data flow and the real P2P protocol minus the xGMI hop."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


@pytest.mark.parametrize("world", [2, 4])
def test_engine_tp_generates_reference_ids(world):
    if not torch.cuda.is_available() or torch.cuda.device_count() < world:
        pytest.skip(f"needs {world} GPUs on this node (found {torch.cuda.device_count() if torch.cuda.is_available() else 0})")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "tp_engine_worker.py")]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]


@pytest.mark.parametrize("world", [2, 4])
def test_engine_tp_ranks_sharing_one_gpu(world):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", EMU_TP_SHARED_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "tp_engine_worker.py")]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("p2p all-reduce on") == world, r.stdout[-3000:]


def test_engine_tp8_true_width_shards():
    """Eight ranks sharing this GPU, a 2-layer decoder at the true LLaMA-33B width (52 -> 56 heads, 7 per rank; ffn 2240 per rank):
    greedy ids of the unsharded engine on the same weights at every step whose top-2 logit margin is clear, eager and replayed
    from a hipGraph, with the stand-alone all-reduce launches and with the all-reduce in the projections' tails (mode 3)
    (tests/tp_truewidth_worker.py)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", EMU_TP_SHARED_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "tp_truewidth_worker.py")]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("ids match") == 4 and "DIFFER" not in r.stdout, r.stdout[-3000:]


def test_bench_two_ranks_sharing_one_gpu():
    """`bench.py --gpus 2` end to end (what the driver launches on a multi-GPU node), at reduced depth, with both ranks on this one
    device (EMU_TP_SHARED_GPU=1: gloo rendezvous, every all-reduce through the peer-to-peer kernels): one JSON line from rank 0
    carrying the contract's fields, the TP parallelism label, the all-reduce path, a run marked invalid (reduced depth), and the opt-in
    prefill-schedule leg (serial vs two concurrent lanes)."""
    import json
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", EMU_TP_SHARED_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
           "--layers", "2", "--vit-layers", "2", "--no-legs", "--no-denoise", "--no-fp8", "--no-beam", "--no-cpu-baseline",
           "--tp-prefill-leg", "1100"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["warmup"] == 2 and d["value"] > 0 and d["scaling"] == "strong"
    assert d["config"]["parallelism"].startswith("tp2") and "p2p" in d["config"]["allreduce"] and d["config"]["valid"] is False
    assert {"roofline", "metric", "unit", "ms_per_step", "higher_is_better", "dtype", "data"} <= set(d)
    # the opt-in prefill-schedule leg: an 1100-row prompt under the serial all-reduce schedule and as two concurrent lanes
    ps = d["config"]["tp"]["prefill_schedules"]
    assert ps["S"] == 1100 and ps["serial_ms"] > 0 and ps["two_lane_ms"] > 0, ps
    assert ps["serial_forwards_on_two_lanes"] == 0 and ps["two_lane_forwards_on_two_lanes"] == 4, ps


def test_bench_spawns_its_own_ranks():
    """Exactly `python bench.py --gpus 2 --steps 4 --warmup 1` (no torch.distributed.run in front: the driver's BENCH form applied to
    N > 1): bench.py becomes the launcher, rank 0 prints the ONE JSON line, the exit code is the job's.  Both ranks on this one device
    (EMU_TP_SHARED_GPU=1), reduced depth so the test stays short; the line says which form of the peer-to-peer exchange the soak
    selected."""
    import json
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", EMU_TP_SHARED_GPU="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
           "--layers", "2", "--vit-layers", "2", "--no-legs", "--no-denoise", "--no-fp8", "--no-beam", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["warmup"] == 1 and d["value"] > 0
    tp = d["config"]["tp"]
    assert len(tp["per_rank_ms_per_token"]) == 2 and tp["allreduces_per_token"] == 2 * 2 and tp["p2p_form"] in ("fence-free", "fenced")


def test_p2p_setup_failure_is_refused_not_hung():
    """Corrupted IPC handles: the peer-to-peer path must come up disabled on every rank without hanging; with no RCCL communicator
    to fall back on (ranks sharing one GPU) init_tp raises."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", EMU_TP_SHARED_GPU="1", EMU_TP_BREAK_P2P="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "tp_engine_worker.py")]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("refused as expected") == 2, r.stdout[-3000:]
