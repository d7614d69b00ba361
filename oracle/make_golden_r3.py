"""Round-3 fixtures frozen from the REAL reference -- TEST INFRASTRUCTURE ONLY.

Run in the build container:  python -m oracle.make_golden_r3
Same tiny seeded reference model as oracle/make_golden.py.  Adds what the earlier fixtures lacked:

* ``generate_multi_image_tiny.npz`` -- prompts with SEVERAL ``[<IMG_PLH>]`` blocks and ``image=[n,3,H,W]``
  (Emu2/emu/emu.py:196-203: every block is n_query ``<image>`` slots, the rows of ``project_up(encode_image(image))`` fill
  them in row-major (batch, position) order): one prompt with two images, and a ragged batch whose rows carry one and two
  images.  Greedy ids; the oracle's top-2 logit margin is > 0.05 at every step (asserted here and in the CPU test).
* ``generate_margin_tiny.npz`` -- beam-search modes on prompts picked so that EVERY pruning decision has a margin
  (>= 0.08 nat) in fp32: penalised 3-beam search (repetition_penalty 1.5) on a ragged batch of two prompts, and the
  reference's default 5-beam search with an image.  A bf16 engine must then reproduce the ids exactly, which lets the GPU
  tests assert id equality instead of "one of the rows matches".  (Contrastive search has no such fixture: its selection
  scores differ by ~1e-3 on random-init weights whatever the prompt, so the GPU test replays the GPU's own token path
  on the CPU stand-in engine and compares the candidates' probabilities and degeneration penalties number by number.)

Margins are measured with the product's own host logic (``trace``) running on tests/fake_engine.FakeEngine in fp32.
"""
import itertools
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import emu2_ref as R  # noqa: E402
from oracle import ref_import  # noqa: E402
from oracle.make_golden import OUT, TINY, tiny_cfgs  # noqa: E402
from emu_amd import synth  # noqa: E402

WORDS = ["a photo of", "the quick brown fox", "describe", "what is", "in the garden", "two dogs and", "please compare",
         "a red car next to", "the weather today", "on the table there is", "tell me about", "why does", "an old map of",
         "write a poem about", "how many", "the colour of", "count the", "is there a", "look at", "between these"]


def capture_ids(m, tok, **kw):
    captured = {}
    orig = tok.batch_decode

    def hook(ids, **k2):
        captured["ids"] = ids.clone()
        return orig(ids, **k2)
    tok.batch_decode = hook
    try:
        with torch.no_grad():
            m.generate(**kw)
    finally:
        tok.batch_decode = orig
    return captured["ids"]


def main():
    from emu_amd import llama as L, ops
    from tests import tiny
    from tests.fake_engine import FakeEngine
    t = TINY
    v, l, vocab = tiny_cfgs(t)
    sd = synth.synth_state_dict(synth.emu_param_shapes(v, l, vocab), seed=t["seed"], lm_head_scale=t["lm_head_scale"])
    d = ref_import.tiny_llama_dir(t["hidden"], t["ffn"], t["heads"], t["llayers"])
    vk = dict(image_size=v.image_size, patch_size=v.patch_size, width=v.width, layers=v.layers, head_width=v.head_width,
              mlp_ratio=v.mlp_ratio, n_query=v.n_query, v_query=v.v_query)
    m = ref_import.build_reference(vk, d, t["instruct"], sd)
    tok = m.decoder.tokenizer
    meta = {"cfg_" + k: np.array(val) for k, val in t.items()}
    W = {k: x.float() for k, x in sd.items()}
    cfg = tiny.oracle_cfg(v, l, vocab)
    expand = lambda x: x.replace("[<IMG_PLH>]", m.image_placeholder)

    # the product's host logic on the CPU stand-in engine (fp32)
    L.BF16 = torch.float32
    ops.embed_gather = lambda ids, table, out=None: out.copy_(table[ids.long()])
    new_eng = lambda: FakeEngine(l, vocab, W, cfg.llama)

    g = torch.Generator().manual_seed(505)
    imgs = torch.randn(3, 3, v.image_size, v.image_size, generator=g)

    # ---------------------------------------------------------------- 1. several images in one prompt / ragged rows
    out = dict(images=imgs.numpy(), **meta)
    found = 0
    for a, b, c in itertools.permutations(WORDS[:10], 3):
        if found == 2:
            break
        if found == 0:
            text, image = [f"{a} [<IMG_PLH>] {b} [<IMG_PLH>] {c}"], imgs[:2]
            tag, n_new = "a", 8
        else:
            text, image = [f"[<IMG_PLH>]{a}", f"{b} [<IMG_PLH>] and [<IMG_PLH>] {c}:"], imgs
            tag, n_new = "b", 6
        enc = tok([expand(x) for x in text], padding="longest", return_tensors="pt")
        ids, margins = R.emu_generate(enc.input_ids, enc.attention_mask, image, W, cfg, max_new_tokens=n_new, return_margins=True)
        if float(margins.min()) <= 0.08 or min(len(set(r)) for r in ids.tolist()) < 4:
            continue                                            # usable margins, and not one token repeated
        ref = capture_ids(m, tok, text=text, image=image, num_beams=1, max_new_tokens=n_new)
        assert ref.tolist() == ids.tolist(), (ref.tolist(), ids.tolist())
        out.update({f"ids_{tag}": enc.input_ids.numpy(), f"mask_{tag}": enc.attention_mask.numpy(), f"new_{tag}": ref.numpy(),
                    f"n_img_{tag}": np.array(image.shape[0])})
        print("multi-image", tag, text, "margin", float(margins.min()), ref.tolist())
        found += 1
    assert found == 2
    np.savez(os.path.join(OUT, "generate_multi_image_tiny.npz"), **out)

    # ---------------------------------------------------------------- 2. margin-controlled beam modes
    out = dict(image=imgs[:1].numpy(), **meta)
    # 2a. penalised 3-beam search, ragged batch of two text prompts.  Near-ties are the norm on random-init weights
    # (typical pruning margin 0.02 nat), so prompts are screened one by one and then paired.
    def singles(run):
        for (a, b), suffix in itertools.product(itertools.permutations(WORDS, 2), (" that", ":", " of the", "?")):
            text = [f"{a} {b}{suffix}"]
            enc = tok(text, padding="longest", return_tensors="pt")
            tr = {}
            run(enc, tr)
            if tr.get("margin", 0.0) >= 0.1:
                yield text[0], int(enc.input_ids.shape[1])

    def pair(run, what):
        good = []
        for text, n in singles(run):
            for other, n2 in good:
                if n2 == n:
                    continue
                texts = [other, text] if n2 < n else [text, other]          # shorter row first: it is the left-padded one
                enc = tok(texts, padding="longest", return_tensors="pt")
                tr = {}
                got = run(enc, tr)
                if tr.get("margin", 0.0) >= 0.08:
                    print(what, texts, "margin", tr["margin"], got.tolist())
                    return texts, enc, got, tr["margin"]
            good.append((text, n))
        raise RuntimeError("no usable pair for " + what)

    run_pen = lambda enc, tr: L.LlamaEngine.beam_search_generate(new_eng(), R.embed_tokens(enc.input_ids, W), enc.attention_mask,
                                                                 3, 8, repetition_penalty=1.5, trace=tr)
    text, enc, got, mg = pair(run_pen, "penalised beam")
    assert not bool(enc.attention_mask.all())                               # ragged
    ref = capture_ids(m, tok, text=text, num_beams=3, repetition_penalty=1.5, max_new_tokens=8)
    assert ref.tolist() == got.tolist(), (ref.tolist(), got.tolist())
    out.update(pen_ids=enc.input_ids.numpy(), pen_mask=enc.attention_mask.numpy(), pen_new=ref.numpy(), pen_margin=np.array(mg))
    # the same prompts with no_repeat_ngram_size = 2 and num_return_sequences = 2 (EmuModel.generate forwards **kwargs to
    # transformers' generate, emu.py:175,228): 4 rows, prompt-major
    ref2 = capture_ids(m, tok, text=text, num_beams=3, max_new_tokens=8, no_repeat_ngram_size=2, num_return_sequences=2)
    got2 = L.LlamaEngine.beam_search_generate(new_eng(), R.embed_tokens(enc.input_ids, W), enc.attention_mask, 3, 8,
                                              no_repeat_ngram_size=2, num_return_sequences=2)
    assert ref2.tolist() == got2.tolist(), (ref2.tolist(), got2.tolist())
    out.update(ngram_new=ref2.numpy())
    print("ngram2 / 2 sequences", ref2.tolist())
    # 2b. the default decoding mode: 5 beams, 10 tokens, length_penalty -1, one image
    best = None
    e = R.encode_image(imgs[:1], W, cfg)
    e = torch.nn.functional.linear(e.reshape(-1, e.shape[-1]), W["project_up.weight"])
    cands = [w + sfx for w, sfx in itertools.product(WORDS, ("", ":", " this?", " that"))]
    cands += [f"{a} {b}" for a, b in itertools.permutations(WORDS, 2)]
    for a in cands:
        text = [f"[<IMG_PLH>]{a}"]
        enc = tok([expand(x) for x in text], padding="longest", return_tensors="pt")
        x = R.scatter_image_embeds(R.embed_tokens(enc.input_ids, W), enc.input_ids, e)
        for n_new in (10, 8):
            tr = {}
            got = L.LlamaEngine.beam_search_generate(new_eng(), x, enc.attention_mask, 5, n_new, trace=tr)
            if best is None or tr["margin"] > best[0]:
                best = (tr["margin"], text, enc, got, n_new)
        if best[0] >= 0.08:
            break
    mg, text, enc, got, n_new = best
    print("5-beam", text, "n_new", n_new, "margin", mg)
    assert mg >= 0.08, mg
    ref = capture_ids(m, tok, text=text, image=imgs[:1], num_beams=5, max_new_tokens=n_new)
    assert ref.tolist() == got.tolist(), (ref.tolist(), got.tolist())
    out.update(b5_ids=enc.input_ids.numpy(), b5_mask=enc.attention_mask.numpy(), b5_new=ref.numpy(), b5_margin=np.array(mg),
               b5_n_new=np.array(n_new))
    np.savez(os.path.join(OUT, "generate_margin_tiny.npz"), **out)
    for f in ("generate_multi_image_tiny.npz", "generate_margin_tiny.npz"):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
