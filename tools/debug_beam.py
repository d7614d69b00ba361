import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import tiny
from oracle import emu2_ref as R
from emu_amd import EmuModel, TextDecoderCfg, ops
BF16 = torch.bfloat16
gd = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
z = tiny.load(gd, "generate_tiny.npz")
v, l, vocab, W = tiny.weights_from(z)
m = EmuModel(v, TextDecoderCfg(instruct=True), llama_cfg=l, device="cuda")
m.load_state_dict(W, strict=True)
lm = m.decoder.lm
t = lambda a: torch.from_numpy(np.asarray(a))
# decode consistency: 5 identical rows vs 1 row
g = torch.Generator().manual_seed(5)
x = (torch.randn(1, 40, l.hidden_size, generator=g) * 0.5).to(BF16).cuda()
mask = torch.ones(1, 40, dtype=torch.long)
part, kstart, pos = lm.prefill(x[:, :39].contiguous(), mask[:, :39])
step1 = lm.decode_embeds(x[:, 39, :].contiguous(), pos, 39, kstart).clone()
lg1 = lm.logits(step1).float().cpu()
# now 5 rows
x5 = x.expand(5, -1, -1).contiguous()
mask5 = torch.ones(5, 40, dtype=torch.long)
part5, kstart5, pos5 = lm.prefill(x5[:, :39].contiguous(), mask5[:, :39])
step5 = lm.decode_embeds(x5[:, 39, :].contiguous(), pos5, 39, kstart5).clone()
lg5 = lm.logits(step5).float().cpu()
print("prefill rows equal:", bool(torch.equal(part5[0], part5[4])), float((part5[0].float()-part.float()[0]).abs().max()))
print("decode 5 vs 1 max abs diff:", float((step5.float().cpu() - step1.float().cpu()).abs().max()), "rows equal:", bool(torch.equal(step5[0], step5[3])))
print("logits 5 vs 1 max abs diff:", float((lg5 - lg1).abs().max()))
# replicate-kv path used by beam search
part, kstart, pos = lm.prefill(x[:, :39].contiguous(), mask[:, :39])
k_old, v_old = lm.kcache, lm.vcache
lm.kcache = lm.vcache = None
lm.alloc_kv(5, lm.cfg.max_position_embeddings)
rep = torch.zeros(5, dtype=torch.long, device="cuda")
lm.kcache[:, :, :, :39] = k_old[:, rep, :, :39]
lm.vcache[:, :, :, :39] = v_old[:, rep, :, :39]
step5b = lm.decode_embeds(x5[:, 39, :].contiguous(), pos.repeat_interleave(5).contiguous(), 39, kstart.repeat_interleave(5).contiguous())
print("replicated-cache decode vs 1-row:", float((step5b.float().cpu() - step1.float().cpu()).abs().max()))
b1 = m.generate_ids(t(z["ids1"]), t(z["mask1"]), t(z["image"]).cuda(), max_new_tokens=10, num_beams=5)
print(b1.tolist(), z["beam1"].tolist())
