"""Parity at BASELINE.json's OWN sizes (configs[1]): the full Emu2-Chat model -- EVA-CLIP ViT 64 blocks x 1792, LLaMA-33B 60
layers x 6656 -- on the bench's synthetic prompt (1 x 448^2 image + 512 tokens, S = 770), against the CPU oracle
(oracle/emu2_ref.py) on the GPU box's host cores.

The oracle cannot hold 65 GB of fp32 weights twice, and does not have to: the synthetic weights are regenerable per tensor from
the seed (emu_amd/synth.py), so the CPU pass STREAMS -- one ViT block / one decoder layer resident at a time, regenerated on
the GPU (the same generator the engine's packer consumed, hence bit-identical bf16 values), copied to the host, used, freed.

What is checked (tolerances stated per assertion):
  * ViT forward_features of the bench image, all 64 blocks: relative L2 of the [1025, 1792] features;
  * LLaMA: the GPU prefills the 770-slot prompt and decodes 8 greedy tokens; ONE teacher-forced fp32 oracle pass over
    prompt + the GPU's tokens gives the reference hidden states of every position and the reference logits of the 8 decoding
    positions: (a) relative L2 of the final residual stream, (b) the GPU's token == the oracle's argmax at every step whose
    oracle top-2 margin is > 0.05 (lm_head scaled x8 as the tiny fixtures do, so margins are not degenerate), (c) the cached
    decode path and a one-shot prefill over the same 777 rows agree.
Slow by design (about a minute of host GEMMs); not skipped."""
import time

import pytest
import torch

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16
SEED = 0


def rel_err(got, want):
    got, want = got.float().cpu(), want.float().cpu()
    return float((got - want).norm() / want.norm().clamp_min(1e-12))


@pytest.fixture(scope="module")
def full_model():
    from emu_amd import CLIPVisionCfg, LlamaCfg, TextDecoderCfg, synth
    from emu_amd.constants import VOCAB_EMU2_CHAT
    from emu_amd.emu import EmuModel
    dev = torch.device("cuda", 0)
    vcfg = CLIPVisionCfg(n_query=256, v_query=64)
    lcfg = LlamaCfg()
    assert (vcfg.layers, vcfg.width, lcfg.num_hidden_layers, lcfg.hidden_size) == (64, 1792, 60, 6656)
    m = EmuModel(vcfg, TextDecoderCfg(instruct=True), llama_cfg=lcfg, device=dev)
    shapes = synth.emu_param_shapes(vcfg, lcfg, VOCAB_EMU2_CHAT)
    m.load_weights(synth.iter_synth(shapes, seed=SEED, device=dev, dtype=BF16, lm_head_scale=8.0), strict=True)
    torch.cuda.synchronize()
    yield m, vcfg, lcfg, shapes
    del m
    torch.cuda.empty_cache()


def _host(name, shapes, lm_head_scale=1.0):
    """One synthetic tensor exactly as the engine received it (generated on the GPU in bf16), as fp32 on the host."""
    from emu_amd import synth
    return synth.synth_tensor(name, shapes[name], SEED, "cuda", BF16, lm_head_scale=lm_head_scale).float().cpu()


def _bench_inputs(vcfg):
    from emu_amd.constants import IMAGE_TOKEN_ID, IMG_END_TOKEN_ID, IMG_TOKEN_ID, OPENAI_DATASET_MEAN, OPENAI_DATASET_STD
    g = torch.Generator().manual_seed(2)
    text_ids = torch.randint(3, 32000, (512,), generator=g)
    block = torch.tensor([IMG_TOKEN_ID] + [IMAGE_TOKEN_ID] * vcfg.n_query + [IMG_END_TOKEN_ID])
    ids = torch.cat([text_ids[:256], block, text_ids[256:]])[None]
    gi = torch.Generator().manual_seed(1)
    img = torch.rand(1, 3, 448, 448, generator=gi)
    img = (img - torch.tensor(OPENAI_DATASET_MEAN)[None, :, None, None]) / torch.tensor(OPENAI_DATASET_STD)[None, :, None, None]
    return ids, img


def test_full_depth_vit_matches_streamed_oracle(full_model):
    """EVAVisionTransformer.forward_features at the true configuration (eva_vit.py:402-431), 64 post-norm blocks: relative
    L2 < 4e-2 against the fp32 oracle on the same bf16-valued weights (the 2-block tiny model is held to 2e-2; 64 blocks of
    bf16 rounding points accumulate, the post-norm LayerNorms keep the growth sub-linear)."""
    from oracle import emu2_ref as R
    m, vcfg, lcfg, shapes = full_model
    _, img = _bench_inputs(vcfg)
    img_b = img.to(BF16)
    got = m.visual(img_b.cuda())
    cfg = R.VitCfg(image_size=448, patch_size=14, width=1792, layers=64, head_width=112, mlp_hidden=vcfg.mlp_hidden)
    t0 = time.time()
    with torch.no_grad():
        W = {k: _host(k, shapes) for k in ("visual.cls_token", "visual.pos_embed", "visual.patch_embed.proj.weight",
                                           "visual.patch_embed.proj.bias")}
        x = R.vit_patch_embed(img_b.float(), W)
        x = torch.cat((W["visual.cls_token"].expand(1, -1, -1), x), dim=1) + W["visual.pos_embed"]
        for i in range(cfg.layers):
            Wi = {k: _host(k, shapes) for k in shapes if k.startswith(f"visual.blocks.{i}.")}
            x = R.vit_block(x, Wi, i, cfg)
            del Wi
    err = rel_err(got, x)
    print(f"full-depth ViT: rel L2 {err:.4f}, host pass {time.time() - t0:.1f} s")
    assert got.shape == x.shape == (1, 1025, 1792)
    assert err < 4e-2, err


def test_full_depth_llama_prefill_and_greedy_match_streamed_oracle(full_model):
    from oracle import emu2_ref as R
    m, vcfg, lcfg, shapes = full_model
    lm = m.decoder.lm
    ids, img = _bench_inputs(vcfg)
    S, n_new = ids.shape[1], 8
    assert S == 770
    mask = torch.ones(1, S, dtype=torch.long)
    x = m._prompt_embeds(ids, img.to(BF16).cuda(), vcfg.n_query).view(1, S, -1)
    new = lm.greedy_generate(x, mask, n_new, stop_on_eos=False)                  # [1, 8] cached decode, eager launches
    m.use_graph = True
    try:
        new_g = m.generate_ids(ids, mask, img.to(BF16).cuda(), max_new_tokens=n_new, stop_on_eos=False)
    finally:
        m.use_graph = False
    assert new_g.tolist() == new.tolist()                                        # hipGraph replay == eager at full depth
    # teacher-forced sequence: prompt + the first 7 generated tokens -> positions 769 .. 776 predict tokens 0 .. 7
    full = torch.cat((x, lm.embed_tokens(new[:, : n_new - 1]).view(1, n_new - 1, -1)), dim=1).contiguous()
    T = S + n_new - 1
    hidden, _, _ = lm.prefill(full, torch.ones(1, T, dtype=torch.long), lm.kv_capacity(T + 8))
    got_h = hidden[0].float().cpu()                                              # residual stream after layer 60, not normed
    got_logits = lm.logits(hidden[0, S - 1:, :].contiguous()).float().cpu()      # [8, V] one-shot prefill path

    cfg = R.LlamaCfg(hidden=lcfg.hidden_size, heads=lcfg.num_attention_heads, layers=lcfg.num_hidden_layers,
                     ffn=lcfg.intermediate_size, vocab=lm.vocab, rms_eps=lcfg.rms_norm_eps, rope_theta=lcfg.rope_theta,
                     max_pos=lcfg.max_position_embeddings)
    t0 = time.time()
    with torch.no_grad():
        h = full.float().cpu()
        pos = torch.arange(T)[None]
        cos, sin = R.rope_cos_sin(pos, cfg.head_dim, cfg.rope_theta, torch.float32)
        amask = R.build_mask(torch.ones(1, T, dtype=torch.long), T, torch.float32)
        for i in range(cfg.layers):
            Wi = {k: _host(k, shapes) for k in shapes if k.startswith(f"decoder.lm.model.layers.{i}.")}
            h = R.llama_layer(h, Wi, i, cfg, cos, sin, amask, None)
            del Wi
        want_h = h[0]
        hn = R.rms_norm(want_h[S - 1:], _host("decoder.lm.model.norm.weight", shapes), cfg.rms_eps)
        want_logits = torch.nn.functional.linear(hn, _host("decoder.lm.lm_head.weight", shapes, lm_head_scale=8.0))
    err = rel_err(got_h, want_h)
    err_tail = rel_err(got_h[S - 1:], want_h[S - 1:])
    top2 = want_logits.topk(2, dim=-1).values
    margin = (top2[:, 0] - top2[:, 1])
    want_ids = want_logits.argmax(-1)
    print(f"full-depth LLaMA: rel L2 of the final stream {err:.4f} (decoding rows {err_tail:.4f}), oracle margins "
          f"{[round(float(v), 3) for v in margin]}, gpu ids {new[0].tolist()}, oracle ids {want_ids.tolist()}, "
          f"host pass {time.time() - t0:.1f} s")
    # (a) 60 layers x 2 residual adds of bf16 rounding against an fp32 oracle: stated tolerance 4e-2 (2-layer models: 2e-2)
    assert err < 4e-2 and err_tail < 4e-2, (err, err_tail)
    # (b) token ids: exact wherever the oracle's decision is not a near-tie
    checked = 0
    for j in range(n_new):
        if float(margin[j]) > 0.05:
            assert int(new[0, j]) == int(want_ids[j]), (j, new[0].tolist(), want_ids.tolist(), float(margin[j]))
            checked += 1
    assert checked >= n_new // 2, f"only {checked} of {n_new} steps had a usable margin: {margin.tolist()}"
    # (c) cached decode vs the one-shot prefill over the same rows: the same logits up to bf16 accumulation order
    assert rel_err(got_logits, want_logits) < 4e-2
    for j in range(n_new):
        if float(margin[j]) > 0.05:
            assert int(got_logits[j].argmax()) == int(want_ids[j]), (j, float(margin[j]))
