"""CPU, world size 2 (gloo): the rank-pair plumbing of the CFG split (emu_amd/tp.py::CfgPair) -- halves, all_gather order, broadcast
from rank 0 -- without a GPU (tensors on the CPU take the same code path as the host-staged exchange of the shared-GPU runs)."""
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["EMU_ROOT"])
dist.init_process_group("gloo")
from emu_amd.tp import CfgPair
p = CfgPair()
r = dist.get_rank()
assert p.half == r and p.host
parts = p.all_gather(torch.full((3, 4), float(r + 1)))
assert [float(x[0, 0]) for x in parts] == [1.0, 2.0]
b = p.broadcast(torch.full((2,), float(10 + r)))
assert b.tolist() == [10.0, 10.0]
dist.barrier(); dist.destroy_process_group()
print("ok", r)
'''


def test_cfg_pair_world2_gloo(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    w = tmp_path / "w.py"
    w.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", EMU_ROOT=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(w)]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("ok") == 2
