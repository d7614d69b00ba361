"""GPU: the UNet's classifier-free-guidance pair split over two rank processes (replaces nothing in the reference, whose
pair is one batch: Emu2/emu/diffusion.py:131-145).  Both ranks share this runner's one GPU (gloo rendezvous, host-staged exchange);
on a multi-GPU node the same code runs one rank per GPU over RCCL."""
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


def test_cfg_pair_split_over_two_ranks():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "cfg_split_worker.py")]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("identical latents on both ranks: True") == 2, r.stdout[-2000:]
