"""Freeze the REAL reference's beam-search multinomial sampling and penalised beam search -- TEST INFRASTRUCTURE ONLY.

Run in the build container:  python -m oracle.make_golden_beam_sample
Same tiny seeded reference model as oracle/make_golden.py (Emu2/emu/emu.py:155-235 forwards num_beams / do_sample / top_k /
top_p / temperature / repetition_penalty to transformers' generate).  Sampling draws come from torch's global CPU generator,
seeded right before each call; tests/test_host_logic.py seeds it the same way and must reproduce the ids.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_import  # noqa: E402
from oracle.make_golden import OUT, TINY, tiny_cfgs  # noqa: E402
from emu_amd import synth  # noqa: E402

CASES = {                                   # name -> generate kwargs (all on the ragged text-only batch of 2 prompts)
    "bs_sample": dict(num_beams=3, do_sample=True, top_k=40, top_p=0.9, temperature=0.7, max_new_tokens=8),
    "bs_sample_plain": dict(num_beams=4, do_sample=True, max_new_tokens=6),
    "bs_penalty": dict(num_beams=3, repetition_penalty=1.5, max_new_tokens=8),
}
SEED = 4242


def main():
    t = TINY
    v, l, vocab = tiny_cfgs(t)
    sd = synth.synth_state_dict(synth.emu_param_shapes(v, l, vocab), seed=t["seed"], lm_head_scale=t["lm_head_scale"])
    d = ref_import.tiny_llama_dir(t["hidden"], t["ffn"], t["heads"], t["llayers"])
    vk = dict(image_size=v.image_size, patch_size=v.patch_size, width=v.width, layers=v.layers, head_width=v.head_width,
              mlp_ratio=v.mlp_ratio, n_query=v.n_query, v_query=v.v_query)
    m = ref_import.build_reference(vk, d, t["instruct"], sd)
    tok = m.decoder.tokenizer
    text = ["a photo of", "an image of a very large dog that"]
    enc = tok(text, padding="longest", return_tensors="pt")
    out = {"ids": enc.input_ids.numpy(), "mask": enc.attention_mask.numpy(), "seed": np.array(SEED)}
    out.update({"cfg_" + k: np.array(val) for k, val in t.items()})
    for name, kw in CASES.items():
        captured = {}
        orig = tok.batch_decode

        def hook(ids, **k2):
            captured["ids"] = ids.clone()
            return orig(ids, **k2)
        tok.batch_decode = hook
        try:
            torch.manual_seed(SEED)
            with torch.no_grad():
                m.generate(text=text, **kw)
        finally:
            tok.batch_decode = orig
        out[name] = captured["ids"].numpy()
        print(name, kw, captured["ids"].tolist())
    np.savez(os.path.join(OUT, "generate_beam_sample_tiny.npz"), **out)


if __name__ == "__main__":
    main()
