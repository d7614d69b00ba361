"""GPU parity tests of the Emu1 caption path (BASELINE.json configs[0]): the ViT-g tower and the CausalFormer against outputs
of the REAL Emu1 modules (tests/golden/emu1_tiny.npz, oracle/make_golden_emu1.py) and against oracle/emu1_ref.py, which that
fixture pins; Emu.generate at the id level against the oracle."""
import pytest
import torch

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16


def rel_err(got, want):
    got, want = got.float().cpu(), want.float().cpu()
    return float((got - want).norm() / want.norm().clamp_min(1e-12))


@pytest.fixture(scope="module")
def tiny_emu1():
    from emu_amd import synth
    from emu_amd.emu1 import Emu, T5DecoderCfg, emu1_llama_cfg, emu1_param_shapes, emu1_vision_cfg
    from oracle import emu1_ref as E, emu2_ref as R
    v = emu1_vision_cfg(image_size=56, width=176, layers=2, head_width=88, mlp_ratio=2.0)     # 2 heads x 88 (padded to 128)
    l = emu1_llama_cfg(hidden_size=256, intermediate_size=512, num_attention_heads=2, num_hidden_layers=2)
    t5 = T5DecoderCfg(d_model=128, num_layers=2, num_heads=2, d_ff=256, n_causal=8)
    W = synth.synth_state_dict(emu1_param_shapes(v, t5, l, 32006), seed=4, lm_head_scale=8.0)
    m = Emu(v, l, t5, vocab=32006, device="cuda")
    m.load_state_dict(W, strict=True)
    cfg = E.Emu1Cfg(vit=R.VitCfg(image_size=56, patch_size=14, width=176, layers=2, head_width=88, mlp_hidden=v.mlp_hidden),
                    t5=E.T5Cfg(d_model=128, layers=2, heads=2, d_ff=256, n_causal=8),
                    llama=R.LlamaCfg(hidden=256, heads=2, layers=2, ffn=512, vocab=32006))
    return m, R.bf16_round(W), cfg


def test_vit_g_prenorm_and_causal_former(tiny_emu1):
    """EVA-CLIP-g pre-norm blocks + ln_visual + CausalFormer (T5 decoder, relative-position bias, unscaled attention):
    relative L2 error < 3e-2 on the 8 visual tokens handed to the LLaMA."""
    from oracle import emu1_ref as E
    m, W, cfg = tiny_emu1
    img = torch.randn(2, 3, 56, 56, generator=torch.Generator().manual_seed(1)).to(BF16)
    feats = m.visual(img.cuda())
    want_f = E.vit_g_forward(img.float(), W, cfg.vit)
    assert rel_err(feats, want_f) < 2e-2, rel_err(feats, want_f)
    got = m.encode_image(img.cuda())
    want = E.encode_image(img.float(), W, cfg)
    assert got.shape == want.shape == (2, 8, 256)
    assert rel_err(got, want) < 3e-2, rel_err(got, want)


def test_vit_g_and_causal_former_match_real_reference(golden_dir):
    """The HIP engines against outputs of the REAL Emu1 modules (tests/golden/emu1_tiny.npz): ViT-g features and the 8
    visual tokens after ln_visual + CausalFormer, relative L2 error < 3e-2 (bf16 engine vs fp32 reference)."""
    from tests import tiny
    from emu_amd.emu1 import CausalFormer
    from emu_amd.llama import EmuHipContext
    from emu_amd.vit import VitEngine
    from emu_amd import ops
    z = tiny.load(golden_dir, "emu1_tiny.npz")
    v, t5, out_dim, W, cfg = tiny.emu1_from(z)
    ctx = EmuHipContext(torch.device("cuda", 0))
    vit = VitEngine(v, ctx)
    for k, t in W.items():
        if k.startswith("visual."):
            vit.load_tensor(k[len("visual."):], t)
    assert vit.ready
    img = torch.from_numpy(z["image"])
    feats = vit(img.to(BF16).cuda())
    want_f = torch.from_numpy(z["feats"])
    assert rel_err(feats, want_f) < 2e-2, rel_err(feats, want_f)
    cf = CausalFormer(t5, v.width, out_dim, ctx)
    for k, t in W.items():
        if k.startswith("cformer."):
            cf.load_tensor(k, t)
    assert cf.ready
    B, N, C = feats.shape
    lnv = ops.layernorm(feats.reshape(B * N, C).contiguous(), W["ln_visual.weight"].to("cuda", BF16),
                        W["ln_visual.bias"].to("cuda", BF16), 1e-6).view(B, N, C)
    assert rel_err(lnv, torch.from_numpy(z["ln_visual"])) < 2e-2
    got = cf.forward(lnv.contiguous())
    want = torch.from_numpy(z["cformer"])
    assert got.shape == want.shape
    assert rel_err(got, want) < 3e-2, rel_err(got, want)


def test_emu1_generate_follows_real_reference(golden_dir):
    """Emu.generate on the HIP engines against ids of the REAL Emu1 class (tests/golden/emu1_generate_tiny.npz): greedy
    exact (the fixture's top-2 margins are > 0.06), default 5-beam search compared by sequence log-probability."""
    from tests import tiny
    from emu_amd.emu1 import Emu
    from oracle import emu1_ref as E, emu2_ref as R
    z = tiny.load(golden_dir, "emu1_generate_tiny.npz")
    v, t5, l, vocab, W, cfg = tiny.emu1_generate_from(z)
    m = Emu(v, l, t5, vocab=vocab, device="cuda")
    m.load_state_dict(W, strict=True)
    ids, mask, img = torch.from_numpy(z["ids"]), torch.from_numpy(z["mask"]), torch.from_numpy(z["image"])
    got = m.generate_ids(ids, mask, img.cuda(), num_beams=1, max_new_tokens=6)
    assert got.cpu().tolist() == z["greedy"].tolist()
    beam = m.generate_ids(ids, mask, img.cuda(), num_beams=5, max_new_tokens=6, length_penalty=0.0).cpu()
    Wb = R.bf16_round(W)

    def seq_lp(seq):
        x = R.embed_tokens(ids, Wb)
        e = E.encode_image(img, Wb, cfg)
        x = R.scatter_image_embeds(x, ids, e.reshape(-1, e.shape[-1]))
        full = torch.cat([x, R.embed_tokens(seq[:, :-1], Wb)], dim=1)
        am = torch.ones(1, full.shape[1], dtype=torch.long)
        h = R.llama_model(full, am, Wb, cfg.llama)
        lp = torch.log_softmax(torch.nn.functional.linear(h[:, ids.shape[1] - 1:], Wb["decoder.lm.lm_head.weight"]).float(), -1)
        return float(lp[0, torch.arange(seq.shape[1]), seq[0]].sum())
    if beam.tolist() != z["beam"].tolist():                          # near-tie pruning under bf16: compare quality instead
        assert seq_lp(beam) > seq_lp(torch.from_numpy(z["beam"])) - 0.15
    # num_captions = 2 with a bigram ban (modeling_emu.py:110,115): two distinct captions, the best one the real reference's
    cap = m.generate_ids(ids, mask, img.cuda(), num_beams=5, max_new_tokens=8, length_penalty=0.0, no_repeat_ngram_size=2,
                         num_return_sequences=2).cpu()
    assert cap.shape == (2, 8) and cap[0].tolist() != cap[1].tolist()
    if cap[0].tolist() != z["beam_cap2_ngram2"][0].tolist():         # this prompt's pruning margins are ~0.003 nat
        assert seq_lp(cap[:1]) > seq_lp(torch.from_numpy(z["beam_cap2_ngram2"][:1])) - 0.15
    for row in cap.tolist():
        big = list(zip(row, row[1:]))
        assert len(big) == len(set(big)), row


def test_emu1_generate_greedy_and_beam(tiny_emu1):
    """Emu.generate at the id level: greedy ids exact vs the oracle when its top-2 margins allow it; beam search (the
    reference default, 5 beams, length_penalty 0) returns a sequence of the right shape through the same engine path."""
    from oracle import emu1_ref as E
    m, W, cfg = tiny_emu1
    img = torch.randn(1, 3, 56, 56, generator=torch.Generator().manual_seed(2)).to(BF16)
    ids = torch.tensor([[1, 32001] + [32003] * 8 + [32002, 500, 600, 700]])
    mask = torch.ones_like(ids)
    want, margins = E.emu1_generate(ids, mask, img.float(), W, cfg, 6, num_beams=1, return_margins=True)
    got = m.generate_ids(ids, mask, img.cuda(), num_beams=1, max_new_tokens=6)
    n_ok = 0
    for t in range(want.shape[1]):                       # compare up to the first near-tie
        if float(margins[0, t]) < 0.05:
            break
        n_ok += 1
    assert n_ok >= 3, margins
    assert got.cpu()[0, :n_ok].tolist() == want[0, :n_ok].tolist()
    beam = m.generate_ids(ids, mask, img.cuda(), num_beams=5, max_new_tokens=6)
    assert beam.shape[0] == 1 and 1 <= beam.shape[1] <= 6
    with pytest.raises(ValueError):
        m.generate_ids(ids[:, :-5], mask[:, :-5], torch.cat([img, img]).cuda(), num_beams=1, max_new_tokens=2)
