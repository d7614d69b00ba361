"""Freeze the REAL reference's default decoding mode on prompts where hypotheses END ON EOS before the length limit -- TEST
INFRASTRUCTURE ONLY.   Run in the build container:  python -m oracle.make_golden_beam_eos

``EmuModel.generate`` (Emu2/emu/emu.py:155-235: num_beams=5, length_penalty=-1 by default) forwards ``**kwargs`` to
``lm.generate`` (:213-229), so ``eos_token_id=<id>`` declares another token the end-of-sequence id for one call.  A random-init
model never emits the real EOS, so the fixture declares tokens the search DOES produce to be EOS -- the finished-hypothesis
branch of the scorer (BeamHypotheses.add / is_done, the length penalty on a short hypothesis, EOS padding of the returned rows)
then runs in the real library (the installed transformers 5.x: what ``hf_semantics="5.x"`` is pinned to; the reference's own pin,
4.31, is not installable here).  Stored: prompts, the declared EOS ids, beams, limits and the returned ids of every case."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_import  # noqa: E402
from oracle.make_golden import OUT, TINY, tiny_cfgs  # noqa: E402
from emu_amd import synth  # noqa: E402


def main():
    t = TINY
    v, l, vocab = tiny_cfgs(t)
    sd = synth.synth_state_dict(synth.emu_param_shapes(v, l, vocab), seed=t["seed"], lm_head_scale=t["lm_head_scale"])
    d = ref_import.tiny_llama_dir(t["hidden"], t["ffn"], t["heads"], t["llayers"])
    vk = dict(image_size=v.image_size, patch_size=v.patch_size, width=v.width, layers=v.layers, head_width=v.head_width,
              mlp_ratio=v.mlp_ratio, n_query=v.n_query, v_query=v.v_query)
    m = ref_import.build_reference(vk, d, t["instruct"], sd)
    tok = m.decoder.tokenizer
    g = torch.Generator().manual_seed(101)
    img = torch.randn(1, 3, v.image_size, v.image_size, generator=g)

    def run(text, image, **kw):
        exp = [x.replace("[<IMG_PLH>]", m.image_placeholder) for x in text]
        enc = tok(exp, padding="longest", return_tensors="pt")
        cap = {}
        orig = tok.batch_decode

        def hook(ids, **k2):
            cap["ids"] = ids.clone()
            return orig(ids, **k2)
        tok.batch_decode = hook
        try:
            with torch.no_grad():
                m.generate(text=text, image=image, **kw)
        finally:
            tok.batch_decode = orig
        return enc.input_ids, enc.attention_mask, cap["ids"]

    cases = [("a", ["[<IMG_PLH>]describe the image in detail:"], img, dict(num_beams=5, max_new_tokens=10)),
             ("b", ["a photo of", "an image of a very large dog that"], None, dict(num_beams=5, max_new_tokens=10)),
             ("c", ["[<IMG_PLH>]where was this taken?"], img, dict(num_beams=3, max_new_tokens=8))]
    out = {"image": img.numpy()}
    out.update({"cfg_" + k: np.array(val) for k, val in t.items()})
    n_cases = 0
    for name, text, image, kw in cases:
        ids, mask, base = run(text, image, **kw)
        # tokens the unconstrained search produces at steps 1..4 of any row: each in turn is declared EOS
        cand = sorted({int(x) for x in base[:, 1:5].reshape(-1).tolist() if int(x) not in (2, 32000)})
        kept = 0
        for eos in cand:
            _, _, got = run(text, image, eos_token_id=eos, **kw)
            early = bool((got == eos).any()) and got.shape[1] <= kw["max_new_tokens"]
            if not early:
                continue
            key = f"{name}{kept}"
            out[key + "_ids"], out[key + "_mask"], out[key + "_out"] = ids.numpy(), mask.numpy(), got.numpy()
            out[key + "_eos"], out[key + "_nb"], out[key + "_n_new"] = np.array(eos), np.array(kw["num_beams"]), np.array(kw["max_new_tokens"])
            out[key + "_has_image"] = np.array(image is not None)
            print(key, "eos", eos, "->", got.tolist())
            kept += 1
            n_cases += 1
            if kept == 3:
                break
    out["n_cases"] = np.array(n_cases)
    np.savez(os.path.join(OUT, "generate_beam_eos_tiny.npz"), **out)
    print("wrote generate_beam_eos_tiny.npz with", n_cases, "cases")


if __name__ == "__main__":
    main()
