#!/usr/bin/env python
"""Latency of the one-shot peer-to-peer all-reduce (csrc/p2p.hip) with N rank processes sharing ONE GPU: 120 all-reduces of
6656 bf16 (one decode token's worth) captured in a hipGraph and replayed.  No xGMI hop is paid here, so this is the protocol's
floor (launch + publish + flag round trip + N reads), not the multi-GPU number.
Usage: python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/p2p_time.py"""
import os, sys, time
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emu_amd.llama import EmuHipContext

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("gloo", rank=rank, world_size=world)


def allgather(b):
    box = [None] * world
    dist.all_gather_object(box, b)
    return box


ctx = EmuHipContext(dev, rank, world)
ctx.init_tp(None, force=True, allgather_bytes=allgather, rccl=False, p2p_timeout_ms=3000)   # 1 rank: the kernel's own cost
for n in (6656, 6656 * 8, 131072):
    x = torch.zeros(n, device=dev, dtype=torch.bfloat16)
    st = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(st):
        for _ in range(4):
            ctx.allreduce(x)
    st.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st):
        for _ in range(120):
            ctx.allreduce(x)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize(); dist.barrier(); t = time.perf_counter()
    reps = 20
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    us = (time.perf_counter() - t) / (reps * 120) * 1e6
    ctx.check_p2p()
    if rank == 0:
        print(f"p2p all-reduce, {world} ranks on one GPU, {n * 2} bytes: {us:.2f} us per all-reduce (hipGraph replay)", flush=True)
dist.barrier()
dist.destroy_process_group()
