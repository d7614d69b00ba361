"""Worker of tests/test_gpu_tp_multiproc.py::test_engine_tp8_true_width_shards: EIGHT rank processes on one GPU (gloo rendezvous,
every all-reduce through the peer-to-peer kernels) run a 2-layer decoder at the TRUE LLaMA-33B width -- 52 heads padded to 56,
seven per rank, ffn 17920 / 8 = 2240, the wave-form o_proj / down_proj of the shards -- and must generate the greedy ids of the
unsharded engine on the same weights wherever the unsharded run's top-2 logit margin is clear (lm_head x 8; steps behind the first
unclear margin are not compared).  The launches, and mode 3 of emu_llama_set_decode_fused (the all-reduce in the tail of the o_proj /
down_proj launches: only that one workgroup waits, for its peers); the fused layers of modes 1 / 2 wait inside a launch, and
rank processes that SHARE a device can starve each other's producer workgroups of CU slots (a 20 s time-out and garbage, observed
with eight ranks here) -- that path needs the device to itself and is covered per shard in tests/test_gpu_decode_fused.py.
Exit code 0 = pass."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
BF16 = torch.bfloat16
S, NEW, V, MARGIN = 96, 12, 2048, 0.5


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from emu_amd import ops, synth
    from emu_amd.conf.emu_conf import LlamaCfg
    from emu_amd.llama import EmuHipContext, GreedyState, LlamaEngine

    def bcast(b):
        box = [b]
        dist.broadcast_object_list(box, src=0)
        return box[0]

    def allgather(b):
        box = [None] * world
        dist.all_gather_object(box, b)
        return box

    cfg = LlamaCfg(num_hidden_layers=2)
    shapes = synth.llama_param_shapes(cfg, V)
    weights = lambda: synth.iter_synth(shapes, seed=11, device=dev, dtype=BF16, lm_head_scale=8.0)
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(1, S, cfg.hidden_size, generator=g) * 0.1).to(BF16).to(dev)
    mask = torch.ones(1, S, dtype=torch.long)
    # ---- the unsharded reference run (rank 0): ids and the top-2 margin of every step's logits
    ref = None
    if rank == 0:
        e1 = LlamaEngine(cfg, V, EmuHipContext(dev, 0, 1))
        e1.load_weights(weights())
        with torch.no_grad():
            hidden, kstart, next_pos = e1.prefill(x, mask, e1.kv_capacity(S + NEW + 8))
            lg = e1.logits(hidden[:, -1, :].contiguous())
            ids, margins = [], []
            cur = ops.argmax(lg, suppress_id=2)
            top2 = lg.float().topk(2).values[0]
            ids.append(int(cur)); margins.append(float(top2[0] - top2[1]))
            out = torch.zeros(NEW + 2, 1, device=dev, dtype=torch.int32)
            st = GreedyState(e1, 1, cur, next_pos, S, kstart, out)
            for _ in range(NEW - 1):
                st.step()
                top2 = st.logits.float().topk(2).values[0]
                ids.append(int(st.cur)); margins.append(float(top2[0] - top2[1]))
        ref = (ids, margins)
        del e1, st
        torch.cuda.empty_cache()
    ids_ref, margins = bcast(ref)
    decided = 0
    while decided < NEW and margins[decided] >= MARGIN:
        decided += 1
    # ---- the sharded engine
    ctx = EmuHipContext(dev, rank, world)
    ctx.init_tp(bcast, allgather_bytes=allgather, rccl=False, p2p_timeout_ms=20000)
    eng = LlamaEngine(cfg, V, ctx)
    eng.load_weights(weights())
    ok = decided >= 6
    for mode in (0, 3):
        eng.set_decode_fused(mode)
        for use_graph in (False, True):
            with torch.no_grad():
                got = eng.greedy_generate(x, mask, NEW, use_graph=use_graph, stop_on_eos=False)[0].tolist()
            same = got[:decided] == ids_ref[:decided]
            ok &= same
            if rank == 0 or not same:
                print(f"rank {rank}: tp{world} heads/rank {eng.plan.heads_local} fused mode {mode} graph {use_graph}: first {decided} of {NEW} "
                      f"steps have margin >= {MARGIN}; ids {'match' if same else 'DIFFER'} {got[:decided]} vs {ids_ref[:decided]}", flush=True)
    ctx.check_p2p()
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if int(flag.item()) == 1 else 1)


if __name__ == "__main__":
    main()
