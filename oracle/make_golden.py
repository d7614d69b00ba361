"""Freeze outputs of the REAL reference into tests/golden/  -- TEST INFRASTRUCTURE ONLY.

Run in the build container:  python -m oracle.make_golden
The reference (``/root/reference/Emu2/emu``) is imported on CPU in fp32 with tiny seeded
configs, loaded with ``emu_amd.synth`` weights (regenerable from the seed, so the fixtures hold
only inputs and outputs), and its ``encode_image`` / ``lm.model`` / ``generate`` /
``generate_image`` outputs are stored.  ``tests/test_oracle_golden.py`` then pins
``oracle/emu2_ref.py`` against these files on any machine.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_import  # noqa: E402
from emu_amd import synth  # noqa: E402
from emu_amd.conf.emu_conf import CLIPVisionCfg, LlamaCfg  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

# tiny config shared with the tests (tests/golden_cfg.py reads the same numbers from the npz)
TINY = dict(image_size=56, patch_size=14, width=224, layers=2, head_width=112, mlp_ratio=2.0,
            n_query=4, v_query=4,
            hidden=256, ffn=512, heads=2, llayers=2, instruct=True, seed=7, lm_head_scale=8.0)


def tiny_cfgs(t=TINY):
    v = CLIPVisionCfg(image_size=t["image_size"], patch_size=t["patch_size"], width=t["width"],
                      layers=t["layers"], head_width=t["head_width"], mlp_ratio=t["mlp_ratio"],
                      n_query=t["n_query"], v_query=t["v_query"])
    l = LlamaCfg(hidden_size=t["hidden"], intermediate_size=t["ffn"], num_attention_heads=t["heads"],
                 num_hidden_layers=t["llayers"])
    vocab = 32274 if t["instruct"] else 32272
    return v, l, vocab


def main():
    os.makedirs(OUT, exist_ok=True)
    t = TINY
    v, l, vocab = tiny_cfgs(t)
    sd = synth.synth_state_dict(synth.emu_param_shapes(v, l, vocab), seed=t["seed"],
                                lm_head_scale=t["lm_head_scale"])
    d = ref_import.tiny_llama_dir(t["hidden"], t["ffn"], t["heads"], t["llayers"])
    vk = dict(image_size=v.image_size, patch_size=v.patch_size, width=v.width, layers=v.layers,
              head_width=v.head_width, mlp_ratio=v.mlp_ratio, n_query=v.n_query, v_query=v.v_query)
    m = ref_import.build_reference(vk, d, t["instruct"], sd)
    tok = m.decoder.tokenizer
    meta = {"cfg_" + k: np.array(val) for k, val in t.items()}

    g = torch.Generator().manual_seed(101)
    # --- 1. ViT + encode_image (emu.py:77-90, eva_vit.py:402-445)
    image = torch.randn(2, 3, v.image_size, v.image_size, generator=g)
    with torch.no_grad():
        feats = m.visual(image)
        enc = m.encode_image(image)
        enc1 = m.encode_image(image, n_query=1)
    np.savez(os.path.join(OUT, "vit_tiny.npz"), image=image.numpy(), feats=feats.numpy(),
             encode=enc.numpy(), encode_nq1=enc1.numpy(), **meta)

    # --- 2. lm.model on inputs_embeds (emu.py:133-138), ragged (left-padded) batch + logits
    S = 12
    embeds = torch.randn(2, S, l.hidden_size, generator=g) * 0.5
    mask = torch.ones(2, S, dtype=torch.long)
    mask[1, :3] = 0
    with torch.no_grad():
        out = m.decoder.lm.model(inputs_embeds=embeds, attention_mask=mask, output_hidden_states=True,
                                 return_dict=True)
        hidden = out.hidden_states[-1]
        logits = m.decoder.lm.lm_head(hidden[:, -1, :])
    np.savez(os.path.join(OUT, "llama_tiny.npz"), embeds=embeds.numpy(), mask=mask.numpy(),
             hidden=hidden.numpy(), logits=logits.numpy(), **meta)

    # --- 3. EmuModel.generate greedy (emu.py:155-235), B=1 with image and B=2 ragged text-only
    img1 = torch.randn(1, 3, v.image_size, v.image_size, generator=g)
    text1 = ["[<IMG_PLH>]describe the image in detail:"]
    text2 = ["a photo of", "an image of a very large dog that"]

    def run_generate(text, image, n_new, num_beams=1, video=None):
        exp = [x.replace("[<IMG_PLH>]", m.image_placeholder).replace("[<VID_PLH>]", m.video_placeholder) for x in text]
        enc_ = tok(exp, padding="longest", return_tensors="pt")
        # the reference returns decoded strings; capture ids by calling lm.generate the same way
        # (emu.py:184-229) through a thin hook on batch_decode
        captured = {}
        orig = tok.batch_decode

        def hook(ids, **kw):
            captured["ids"] = ids.clone()
            return orig(ids, **kw)
        tok.batch_decode = hook
        try:
            with torch.no_grad():
                strs = m.generate(text=text, image=image, video=video, num_beams=num_beams, max_new_tokens=n_new)
        finally:
            tok.batch_decode = orig
        return enc_.input_ids, enc_.attention_mask, captured["ids"], strs

    ids1, am1, new1, s1 = run_generate(text1, img1, 8)
    ids2, am2, new2, s2 = run_generate(text2, None, 6)
    # the reference's DEFAULT decoding: beam search, num_beams=5, max_new_tokens=10, length_penalty=-1 (emu.py:163-172)
    _, _, beam1, sb1 = run_generate(text1, img1, 10, num_beams=5)
    _, _, beam2, sb2 = run_generate(text2, None, 10, num_beams=5)
    # a beam-search case whose pruning decisions have a usable margin (0.1 nat; random-init models make most beam
    # decisions near-ties, see tests/test_oracle_golden.py): 3 beams, 6 tokens -- used for the bf16 GPU comparison
    text3 = ["[<IMG_PLH>]where was this taken?"]
    ids3, am3, beam3, sb3 = run_generate(text3, img1, 6, num_beams=3)
    np.savez(os.path.join(OUT, "generate_tiny.npz"), image=img1.numpy(),
             ids1=ids1.numpy(), mask1=am1.numpy(), new1=new1.numpy(),
             ids2=ids2.numpy(), mask2=am2.numpy(), new2=new2.numpy(),
             beam1=beam1.numpy(), beam2=beam2.numpy(),
             ids3=ids3.numpy(), mask3=am3.numpy(), beam3=beam3.numpy(), **meta)
    print("beam3 (nb=3):", beam3.tolist(), sb3)
    print("beam B=1:", beam1.tolist(), sb1)
    print("beam B=2:", beam2.tolist(), sb2)
    print("generate B=1:", new1.tolist(), s1)
    print("generate B=2:", new2.tolist(), s2)

    # --- 3b. video frames ([gIMG] slots, v_query tokens per frame, emu.py:205-211): frames only, and an image + frames
    vid = torch.randn(2, 3, v.image_size, v.image_size, generator=torch.Generator().manual_seed(303))
    text4 = ["[<VID_PLH>][<VID_PLH>]what happens next?"]
    text5 = ["[<IMG_PLH>]and then[<VID_PLH>][<VID_PLH>]compare them:"]
    ids4, am4, new4, s4 = run_generate(text4, None, 6, video=vid)
    ids5, am5, new5, s5 = run_generate(text5, img1, 6, video=vid)
    np.savez(os.path.join(OUT, "generate_video_tiny.npz"), image=img1.numpy(), video=vid.numpy(),
             ids4=ids4.numpy(), mask4=am4.numpy(), new4=new4.numpy(),
             ids5=ids5.numpy(), mask5=am5.numpy(), new5=new5.numpy(), **meta)
    print("generate video:", new4.tolist(), s4)
    print("generate image+video:", new5.tolist(), s5)

    # --- 4. EmuModel.generate_image (emu.py:92-153), text-only and with an image prompt
    with torch.no_grad():
        gi_text = m.generate_image(text=["a dog on the grass"])
        gi_img = m.generate_image(text=["[<IMG_PLH>]make it red"], image=img1)
    p_text = tok(["a dog on the grass"], return_tensors="pt").input_ids
    p_img = tok([f"{m.image_placeholder}make it red"], return_tensors="pt").input_ids
    # sanity: appending special tokens to the text == appending their ids (used by the restatement)
    chk = tok(["a dog on the grass[IMG]<image><image>"], return_tensors="pt").input_ids
    assert chk[0, -3:].tolist() == [32001, 32003, 32003] and torch.equal(chk[:, :-3], p_text), chk
    np.savez(os.path.join(OUT, "generate_image_tiny.npz"), image=img1.numpy(),
             prompt_text=p_text.numpy(), out_text=gi_text.numpy(),
             prompt_img=p_img.numpy(), out_img=gi_img.numpy(), **meta)
    # --- 5. chat prompt templates (chat.py:121-195): the real reference functions with a torchvision stub (the module
    #        only needs the names at import time; the stub transform returns a zero tensor, strings are what is pinned)
    import json
    import types
    from PIL import Image
    tv, tvt = types.ModuleType("torchvision"), types.ModuleType("torchvision.transforms")
    tvt.Compose = lambda fs: (lambda x: torch.zeros(3, 2, 2))
    tvt.Resize = tvt.ToTensor = tvt.Normalize = lambda *a, **k: None
    tvt.InterpolationMode = types.SimpleNamespace(BICUBIC=3)
    tv.transforms = tvt
    sys.modules.setdefault("torchvision", tv)
    sys.modules.setdefault("torchvision.transforms", tvt)
    from emu.chat import EmuChatGeneration as RefChat
    fake = types.SimpleNamespace(transform=lambda x: torch.zeros(3, 2, 2))
    fake._prepare_inputs = lambda *a, **k: RefChat._prepare_inputs(fake, *a, **k)
    img = Image.new("RGB", (8, 8))
    cases = {
        "plain": ["describe ", img, " and ", img, " please"],
        "video": ["[VIDEO]", img, img, "[/VIDEO]", "what happens?"],
        "chat1": [["hello ", img]],
        "chat3": [[img, "what is this?"], ["a dog."], ["and this? ", img]],
        "ground": [["find the dog ", img]],
    }
    enc = lambda seq: ["<IMG>" if not isinstance(x, str) else x for x in seq]
    out = {}
    for name, inp in cases.items():
        if isinstance(inp[0], list):
            t, im, vd, _, _ = RefChat._prepare_chat_inputs(fake, inp, name == "ground")
            key_in = [enc(m) for m in inp]
        else:
            t, im, vd, _, _ = RefChat._prepare_inputs(fake, inp)
            key_in = enc(inp)
        out[name] = {"inputs": key_in, "text": t[0], "n_image": 0 if im is None else im.shape[0],
                     "n_video": 0 if vd is None else vd.shape[0]}
    json.dump(out, open(os.path.join(OUT, "chat_templates.json"), "w"), indent=1)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
