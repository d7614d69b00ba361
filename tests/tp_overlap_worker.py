"""Worker of tests/test_gpu_tp_overlap.py::test_two_half_schedule_two_ranks_sharing_the_gpu: TWO rank processes on one GPU (gloo
rendezvous, every all-reduce through the peer-to-peer kernels over IPC-mapped comm blocks) run a 3-layer decoder (hidden 1024,
8 heads of 128 -> 4 per rank, so each half's qkv projection takes the fused RoPE / KV append / V^T epilogue) over a 1100-row prompt:
  * the two-half schedule (all-reduces on each rank's second stream) against the unsharded engine on rank 0: residual stream rel-L2,
  * against the serial sharded schedule: residual stream rel-L2, the first token where its margin is clear, and eager == graph-replayed
    decode behind the two-half prefill,
  * the time of both schedules (printed; two processes time-slice one device, so this is not a measurement of the overlap).
Exit code 0 = pass."""
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
BF16 = torch.bfloat16
S, NEW, V = 1100, 8, 2048


def rel(a, b):
    a, b = a.float(), b.float()
    return float((a - b).norm() / b.norm().clamp_min(1e-12))


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from emu_amd import synth
    from emu_amd.conf.emu_conf import LlamaCfg
    from emu_amd.llama import EmuHipContext, LlamaEngine

    def bcast(b):
        box = [b]
        dist.broadcast_object_list(box, src=0)
        return box[0]

    def allgather(b):
        box = [None] * world
        dist.all_gather_object(box, b)
        return box

    cfg = LlamaCfg(hidden_size=1024, intermediate_size=2816, num_attention_heads=8, num_hidden_layers=3)
    shapes = synth.llama_param_shapes(cfg, V)
    weights = lambda: synth.iter_synth(shapes, seed=13, device=dev, dtype=BF16, lm_head_scale=8.0)
    g = torch.Generator().manual_seed(9)
    x = (torch.randn(1, S, cfg.hidden_size, generator=g) * 0.1).to(BF16).to(dev)
    mask = torch.ones(1, S, dtype=torch.long)
    ref = None
    if rank == 0:                                          # the unsharded engine's residual stream
        e1 = LlamaEngine(cfg, V, EmuHipContext(dev, 0, 1))
        e1.load_weights(weights())
        with torch.no_grad():
            ref = e1.prefill(x, mask, e1.kv_capacity(S + NEW + 8))[0].float().cpu()
        del e1
        torch.cuda.empty_cache()
    ctx = EmuHipContext(dev, rank, world)
    ctx.init_tp(bcast, allgather_bytes=allgather, rccl=False, p2p_timeout_ms=20000)
    eng = LlamaEngine(cfg, V, ctx)
    eng.load_weights(weights())
    ok = True
    out = {}
    for name, rows in (("serial", 0), ("two-half", 1024)):
        eng.set_tp_overlap(rows)
        n0 = eng.tp_overlap_count()
        with torch.no_grad():
            cap = eng.kv_capacity(S + NEW + 8)
            hidden = eng.prefill(x, mask, cap)[0].clone()
            lg = eng.logits(hidden[:, -1, :].contiguous()).float()
            torch.cuda.synchronize(); dist.barrier(); t = time.perf_counter()
            for _ in range(3):
                eng.prefill(x, mask, cap)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t) / 3 * 1e3
            ids = [eng.greedy_generate(x, mask, NEW, use_graph=ug, stop_on_eos=False)[0].tolist() for ug in (False, True)]
        took = eng.tp_overlap_count() - n0
        ok &= (took > 0) == (rows > 0)
        out[name] = (hidden, ids, lg)
        if rank == 0:
            print(f"rank 0: tp{world} {name} prefill S={S}: {ms:.2f} ms (forwards on the two-half schedule: {took}); ids {ids[0]}", flush=True)
    r_sched = rel(out["two-half"][0], out["serial"][0])
    ok &= r_sched < 2e-2
    ok &= out["two-half"][1][0] == out["two-half"][1][1]                 # eager == graph decode behind the two-half prefill
    # the ids behind the two prefills: the first token must agree where the serial run's top-2 margin clears the bf16 noise; later
    # ones are reported (random-init weights put many steps on near-ties, and a flipped token changes everything after it)
    same_ids = out["two-half"][1][0] == out["serial"][1][0]
    top2 = out["serial"][2].topk(2).values[0]
    margin = float(top2[0] - top2[1])
    if margin >= 0.25:
        ok &= int(out["two-half"][2].argmax()) == int(out["serial"][2].argmax())
    if rank == 0:
        r_ref = rel(out["two-half"][0].cpu(), ref)
        r_ser = rel(out["serial"][0].cpu(), ref)
        ok &= r_ref < 2e-2 and r_ref < 1.5 * r_ser + 1e-3
        print(f"rank 0: residual stream rel-L2 two-half vs serial {r_sched:.2e}; vs the unsharded engine {r_ref:.2e} (serial: {r_ser:.2e}); "
              f"first-token margin {margin:.3f}; greedy ids {'equal' if same_ids else 'differ behind a near-tie'}", flush=True)
    ctx.check_p2p()
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0 and int(flag.item()) == 1:
        print("two-half prefill ok", flush=True)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if int(flag.item()) == 1 else 1)


if __name__ == "__main__":
    main()
