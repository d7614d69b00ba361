import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from tests import tiny
from emu_amd import EmuModel, TextDecoderCfg
BF16 = torch.bfloat16
gd = "/root/repo/tests/golden"
z = tiny.load(gd, "generate_tiny.npz")
v, l, vocab, W = tiny.weights_from(z)
m = EmuModel(v, TextDecoderCfg(instruct=True), llama_cfg=l, device="cuda")
m.load_state_dict(W, strict=True)
lm = m.decoder.lm
ids, mask = torch.from_numpy(z["ids2"]), torch.from_numpy(z["mask2"])
n_new = 8
b = m.generate_ids(ids, mask, None, max_new_tokens=n_new, stop_on_eos=False).cpu()
lm.quantize_fp8(); lm.use_fp8(True)
f = m.generate_ids(ids, mask, None, max_new_tokens=n_new, stop_on_eos=False).cpu()
lm.use_fp8(False)
print("bf16", b.tolist()); print("fp8 ", f.tolist())
# teacher-forced logits: feed bf16 ids, compare logits per step
x = m._prompt_embeds(ids, None, m.n_query)
S = ids.shape[1]
def run(fp8):
    lm.use_fp8(fp8)
    hidden, kstart, pos = lm.prefill(x.view(ids.shape[0], S, -1), mask)
    out = [lm.logits(hidden[:, -1, :].contiguous()).float().cpu()]
    cur_pos = pos
    for i in range(n_new - 1):
        e = lm.embed_tokens(b[:, i:i+1].cuda()).view(ids.shape[0], -1)
        h = lm.decode_embeds(e, cur_pos, S + i, kstart)
        cur_pos = cur_pos + 1
        out.append(lm.logits(h).float().cpu())
    lm.use_fp8(False)
    return torch.stack(out, 1)
lb, lf = run(False), run(True)
d = (lf - lb)
print("rel l2 per step", [(float(d[:, i].norm() / lb[:, i].norm())) for i in range(n_new)])
t2 = lb.topk(2, -1).values
print("bf16 margins", (t2[..., 0] - t2[..., 1]).tolist())
print("max abs dlogit", d.abs().amax(-1).tolist())
print("argmax agree", (lf.argmax(-1) == lb.argmax(-1)).tolist())
