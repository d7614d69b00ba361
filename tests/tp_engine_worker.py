"""Worker of tests/test_gpu_tp_multiproc.py: one process per GPU (torch.distributed.run), the LLaMA ENGINE sharded over the
ranks (emu_amd/tp.py plan, RCCL all-reduce inside emu_llama_forward) generates the same greedy token ids as the unsharded
engine and as the golden fixture of the real reference.  Exit code 0 = pass."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
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

    ctx = EmuHipContext(dev, rank, world)
    ctx.init_tp(bcast)
    m = EmuModel(v, TextDecoderCfg(instruct=True), llama_cfg=l, device=dev, ctx=ctx)
    m.load_state_dict(W, strict=True)
    ok = True
    for use_graph in (False, True):
        m.use_graph = use_graph
        got1 = m.generate_ids(t(z["ids1"]), t(z["mask1"]), t(z["image"]).to(dev), max_new_tokens=8).cpu()
        got2 = m.generate_ids(t(z["ids2"]), t(z["mask2"]), None, max_new_tokens=6).cpu()
        ok &= got1.tolist() == z["new1"].tolist() and got2.tolist() == z["new2"].tolist()
        if not ok:
            print(f"rank {rank} graph={use_graph}: {got1.tolist()} vs {z['new1'].tolist()}; {got2.tolist()} vs {z['new2'].tolist()}",
                  flush=True)
    # every rank holds the same ids (the all-reduced hidden state is identical on all ranks)
    flag = torch.tensor([1 if ok else 0], device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if int(flag.item()) == 1 else 1)


if __name__ == "__main__":
    main()
