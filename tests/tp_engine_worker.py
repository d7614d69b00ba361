"""Worker of tests/test_gpu_tp_multiproc.py: one process per GPU (torch.distributed.run), the LLaMA ENGINE sharded over the
ranks (emu_amd/tp.py plan, all-reduce inside emu_llama_forward) generates the same greedy token ids as the unsharded
engine and as the golden fixture of the real reference.  Exit code 0 = pass.

EMU_TP_SHARED_GPU=1: every rank uses cuda:0 (the 1-GPU runner).  RCCL refuses two ranks on one device, so the rendezvous is
gloo and EVERY all-reduce goes through the one-shot peer-to-peer path (csrc/p2p.hip) over IPC-mapped comm blocks -- the same
kernels, flags and IPC mappings a multi-GPU node uses, minus the xGMI hop.  Without it (>= 2 GPUs): RCCL for the large
prefill messages, P2P for the decode-sized ones when its self-test passes on every rank."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def p2p_checks(ctx, dev, rank, world):
    """The all-reduce itself: ragged lengths (tail vector), more than one slot (chunks), many back-to-back calls (slot reuse)
    and hipGraph replay (the sequence number lives on the device).  Inputs are small integers, so the fp32 sum is exact and
    the expected bf16 result is known without a second implementation."""
    ok = True
    for n in (1, 7, 8, 6656, 6656 * 3 + 5, 131072, 131072 * 2 + 77):
        base = (torch.arange(n, device=dev) % 11).float()
        want = sum(base * (r + 1) + r for r in range(world)).to(torch.bfloat16)
        for it in range(3):
            x = (base * (rank + 1) + rank).to(torch.bfloat16)
            ctx.allreduce(x)
            ok &= bool(torch.equal(x, want))
    n = 6656
    base = (torch.arange(n, device=dev) % 11).float()
    x = torch.empty(n, device=dev, dtype=torch.bfloat16)
    src = (base * (rank + 1) + rank).to(torch.bfloat16)
    want = sum(base * (r + 1) + r for r in range(world)).to(torch.bfloat16)
    st = torch.cuda.Stream(device=dev)
    st.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(st):
        for _ in range(3):
            x.copy_(src); ctx.allreduce(x); ctx.allreduce(x.copy_(src))
    torch.cuda.current_stream(dev).wait_stream(st)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st):
        x.copy_(src); ctx.allreduce(x)
    for _ in range(50):
        g.replay()
    torch.cuda.synchronize(dev)
    ok &= bool(torch.equal(x, want))
    ctx.check_p2p()
    if not ok:
        print(f"rank {rank}: p2p all-reduce check FAILED", flush=True)
    return ok


def main():
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    shared = os.environ.get("EMU_TP_SHARED_GPU") == "1"
    if shared:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if shared:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from emu_amd import EmuModel, TextDecoderCfg
    from emu_amd.llama import EmuHipContext
    from tests import tiny
    z = tiny.load(os.path.join(ROOT, "tests", "golden"), "generate_tiny.npz")
    v, l, vocab, W = tiny.weights_from(z)
    t = lambda a: torch.from_numpy(np.asarray(a))

    def bcast(b):
        box = [b]
        dist.broadcast_object_list(box, src=0)
        return box[0]

    def allgather(b):
        box = [None] * world
        dist.all_gather_object(box, b)
        return box

    ctx = EmuHipContext(dev, rank, world)
    if os.environ.get("EMU_TP_BREAK_P2P") == "1":
        # failure path: every peer handle arrives corrupted -> the mapping (or the self-test) fails on every rank, nothing hangs,
        # and without an RCCL communicator to fall back on init_tp must raise
        def broken(b):
            got = allgather(b)
            return [x if (i == rank or len(x) != 64) else bytes(64) for i, x in enumerate(got)]
        try:
            ctx.init_tp(bcast, allgather_bytes=broken, rccl=not shared, p2p_timeout_ms=500)
        except RuntimeError as e:
            print(f"rank {rank}: refused as expected: {e}", flush=True)
            dist.barrier()
            dist.destroy_process_group()
            sys.exit(0)
        print(f"rank {rank}: p2p={ctx.p2p} (expected a refusal)", flush=True)
        sys.exit(0 if (not shared and not ctx.p2p) else 1)
    ctx.init_tp(bcast, allgather_bytes=allgather, rccl=not shared, p2p_timeout_ms=3000)
    print(f"rank {rank}: p2p all-reduce {'on' if ctx.p2p else 'OFF (RCCL only)'}", flush=True)
    ok = True
    if shared or ctx.p2p:
        ok &= p2p_checks(ctx, dev, rank, world)
    m = EmuModel(v, TextDecoderCfg(instruct=True), llama_cfg=l, device=dev, ctx=ctx)
    m.load_state_dict(W, strict=True)
    # fused decode layers (cut at the all-reduces) as well where the ranks' grids are small enough to be resident together: two ranks
    # of the tiny model (a launch that waits inside must not starve the other rank's producers of CU slots: one GPU per rank otherwise)
    for use_graph, fused in ((False, 0), (True, 0)) + (((False, 1), (True, 1)) if world == 2 else ()):
        m.use_graph = use_graph
        m.decoder.lm.set_decode_fused(fused)
        got1 = m.generate_ids(t(z["ids1"]), t(z["mask1"]), t(z["image"]).to(dev), max_new_tokens=8).cpu()
        got2 = m.generate_ids(t(z["ids2"]), t(z["mask2"]), None, max_new_tokens=6).cpu()
        ok &= got1.tolist() == z["new1"].tolist() and got2.tolist() == z["new2"].tolist()
        if not ok:
            print(f"rank {rank} graph={use_graph} fused={fused}: {got1.tolist()} vs {z['new1'].tolist()}; {got2.tolist()} vs {z['new2'].tolist()}",
                  flush=True)
    # every rank holds the same ids (the all-reduced hidden state is identical on all ranks)
    flag = torch.tensor([1 if ok else 0], device="cpu" if shared else dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if int(flag.item()) == 1 else 1)


if __name__ == "__main__":
    main()
