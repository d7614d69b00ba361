"""CPU, world_size=2 over gloo: the tensor-parallel data flow (shard plan + residual-once + all-reduce placement)
run as two real processes reproduces the unsharded oracle layer."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from emu_amd import synth
        from emu_amd.conf.emu_conf import LlamaCfg
        from emu_amd.tp import ShardPlan
        from oracle import emu2_ref as R
        from tests.tp_ref import sharded_layer_partial
        heads, D = 5, 16                                        # 5 heads over 2 ranks -> padded to 6
        l = LlamaCfg(hidden_size=heads * D, intermediate_size=128, num_attention_heads=heads, num_hidden_layers=1)
        shapes = {k: s for k, s in synth.llama_param_shapes(l, 64).items() if ".layers.0." in k}
        W = synth.synth_state_dict(shapes, seed=11)
        pre = "decoder.lm.model.layers.0."
        cfg = R.LlamaCfg(hidden=l.hidden_size, heads=heads, layers=1, ffn=128, vocab=64)
        x = torch.randn(2, 9, l.hidden_size, generator=torch.Generator().manual_seed(4))
        pos = torch.arange(9)[None].expand(2, -1)
        cos, sin = R.rope_cos_sin(pos, D, 10000.0, torch.float32)
        mask = R.build_mask(torch.ones(2, 9, dtype=torch.long), 9, torch.float32)
        plan = ShardPlan(l.hidden_size, heads, D, 128, world, rank)
        packed = plan.pack_layer(*(W[pre + k] for k in ("self_attn.q_proj.weight", "self_attn.k_proj.weight",
                                                         "self_attn.v_proj.weight", "self_attn.o_proj.weight",
                                                         "mlp.gate_proj.weight", "mlp.up_proj.weight",
                                                         "mlp.down_proj.weight")))

        def allreduce(t):
            t = t.clone()
            dist.all_reduce(t)
            return t
        out = sharded_layer_partial(x, packed, W[pre + "input_layernorm.weight"],
                                    W[pre + "post_attention_layernorm.weight"], plan, cfg, cos, sin, mask, allreduce)
        want = R.llama_layer(x, W, 0, cfg, cos, sin, mask, None)
        q.put((rank, float((out - want).abs().max())))
    finally:
        dist.destroy_process_group()


def test_tp2_gloo_layer_matches_unsharded():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = [q.get(timeout=180) for _ in procs]
    [p.join(60) for p in procs]
    assert sorted(r for r, _ in res) == [0, 1]
    assert all(err < 1e-4 for _, err in res), res


def _lane_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from emu_amd import synth
        from emu_amd.conf.emu_conf import LlamaCfg
        from emu_amd.tp import ShardPlan
        from oracle import emu2_ref as R
        from tests.tp_ref import sharded_two_lane_prefill
        heads, D, S, Ma, L = 5, 16, 23, 8, 2                    # 5 heads over 2 ranks -> padded to 6; lanes of 8 and 15 rows
        l = LlamaCfg(hidden_size=heads * D, intermediate_size=128, num_attention_heads=heads, num_hidden_layers=L)
        W = synth.synth_state_dict(synth.llama_param_shapes(l, 64), seed=12)
        cfg = R.LlamaCfg(hidden=l.hidden_size, heads=heads, layers=L, ffn=128, vocab=64)
        x = torch.randn(1, S, l.hidden_size, generator=torch.Generator().manual_seed(6))
        pos = torch.arange(S)[None]
        cos, sin = R.rope_cos_sin(pos, D, 10000.0, torch.float32)
        plan = ShardPlan(l.hidden_size, heads, D, 128, world, rank)
        layers = []
        for i in range(L):
            pre = f"decoder.lm.model.layers.{i}."
            packed = plan.pack_layer(*(W[pre + k] for k in ("self_attn.q_proj.weight", "self_attn.k_proj.weight",
                                                             "self_attn.v_proj.weight", "self_attn.o_proj.weight",
                                                             "mlp.gate_proj.weight", "mlp.up_proj.weight", "mlp.down_proj.weight")))
            layers.append((packed, W[pre + "input_layernorm.weight"], W[pre + "post_attention_layernorm.weight"]))
        n_ar = [0]

        def allreduce(t):
            t = t.clone()
            dist.all_reduce(t)
            n_ar[0] += 1
            return t
        out = sharded_two_lane_prefill(x, layers, plan, cfg, cos, sin, allreduce, Ma)
        mask = R.build_mask(torch.ones(1, S, dtype=torch.long), S, torch.float32)
        want = x
        for i in range(L):
            want = R.llama_layer(want, W, i, cfg, cos, sin, mask, None)
        q.put((rank, float((out - want).abs().max()), n_ar[0]))
    finally:
        dist.destroy_process_group()


def test_tp2_gloo_two_lane_prefill_matches_unsharded():
    """The two-lane prefill schedule (emu_llama_set_tp_overlap) as two real processes over gloo: the prompt's rows cut in two chains,
    lane B reading lane A's keys / values of the same layer, four all-reduces per layer in the engine's host order (A's two, then
    B's two) -- reproduces the unsharded oracle stack on all rows."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_lane_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = [q.get(timeout=180) for _ in procs]
    [p.join(60) for p in procs]
    assert sorted(r for r, _, _ in res) == [0, 1]
    assert all(err < 1e-4 for _, err, _ in res), res
    assert all(n == 8 for _, _, n in res), res                  # 2 layers x 2 lanes x 2 all-reduces


def _img_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from emu_amd.tp import image_parallel_encode
        g = torch.Generator().manual_seed(9)
        res = {}
        for n in (1, 2, 3, 4, 5):
            imgs = torch.randn(n, 3, 8, 8, generator=g)
            calls = []

            def enc(x):                                              # stand-in for ViT + pooling: a fixed map per image
                calls.append(x.shape[0])
                return torch.stack([x[i].reshape(-1)[:12].reshape(4, 3) * 2.0 + 1.0 for i in range(x.shape[0])])

            def gather(t):
                parts = [torch.empty_like(t) for _ in range(world)]
                dist.all_gather(parts, t)
                return parts
            out = image_parallel_encode(imgs, enc, rank, world, gather)
            want = torch.stack([imgs[i].reshape(-1)[:12].reshape(4, 3) * 2.0 + 1.0 for i in range(n)])
            res[n] = (bool(torch.equal(out, want)), calls[0])
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


def test_vit_image_parallel_over_two_ranks():
    """BASELINE configs[2] (several images per prompt) under tensor parallelism: every rank encodes ceil(n / world) images and
    the all-gather restores the tokens of all n images, in order, on every rank."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_img_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = [q.get(timeout=180) for _ in procs]
    [p.join(60) for p in procs]
    for rank, r in res:
        assert all(ok for ok, _ in r.values()), (rank, r)
        assert {n: k for n, (_, k) in r.items()} == {1: 1, 2: 1, 3: 2, 4: 2, 5: 3}, r       # images encoded per rank


def _vocab_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from emu_amd.tp import ShardPlan
        V, H, M = 37, 48, 3                                      # 37 rows over 2 ranks: 19 + 18 (the last rank is short, as 32 274 / 8 is)
        g = torch.Generator().manual_seed(21)
        head = (torch.randn(V, H, generator=g) * 0.3).to(torch.bfloat16)
        x = torch.randn(M, H, generator=g).to(torch.bfloat16)
        plan = ShardPlan(80, 5, 16, 128, world, rank)
        r0, r1 = plan.vocab_range(V)
        # what emu_llama_logits does with a sharded head: own columns of zeroed [M, V] rows, then one sum all-reduce
        rows = torch.zeros(M, V, dtype=torch.bfloat16)
        rows[:, r0:r1] = (x.float() @ head[r0:r1].float().t()).to(torch.bfloat16)
        full32 = rows.float()
        dist.all_reduce(full32)                                  # gloo has no bf16 sum; every column has ONE non-zero addend: exact
        want = (x.float() @ head.float().t()).to(torch.bfloat16)
        q.put((rank, (r0, r1), bool(torch.equal(full32.to(torch.bfloat16), want)), int(full32.argmax(-1)[0]), int(want.float().argmax(-1)[0])))
    finally:
        dist.destroy_process_group()


def test_tp2_gloo_vocab_sharded_head_is_bit_identical():
    """SURVEY 8e's vocabulary-sharded lm_head (ShardPlan.vocab_range + the zero-fill / all-reduce of emu_llama_logits) as two real
    processes: every logit is computed by exactly one rank and summed with zeros, so the rows come out bit-identical to the
    replicated head on every rank -- arg-max, beam scorer and samplers stay as they are."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_vocab_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=180) for _ in procs)
    [p.join(60) for p in procs]
    assert [r[1] for r in res] == [(0, 19), (19, 37)]
    assert all(r[2] and r[3] == r[4] for r in res), res
