"""CPU tests of the host logic: TP shard plan, EOS bookkeeping, synthetic weights, config mirrors."""
import pytest
import torch

from emu_amd import synth
from emu_amd.conf.emu_conf import CLIPVisionCfg, LlamaCfg
from emu_amd.tp import ShardPlan
from oracle import emu2_ref as R
from tests.tp_ref import sharded_layer_partial


def _layer_weights(l: LlamaCfg, seed=1):
    shapes = {k: s for k, s in synth.llama_param_shapes(l, 64).items() if ".layers.0." in k}
    return synth.synth_state_dict(shapes, seed=seed)


@pytest.mark.parametrize("heads,tp", [(4, 1), (4, 2), (4, 4), (5, 2), (13, 8), (52, 8)])
def test_shard_plan_sum_of_shards_equals_unsharded_layer(heads, tp):
    D = 16
    l = LlamaCfg(hidden_size=heads * D, intermediate_size=64 * tp, num_attention_heads=heads, num_hidden_layers=1)
    W = _layer_weights(l)
    pre = "decoder.lm.model.layers.0."
    cfg = R.LlamaCfg(hidden=l.hidden_size, heads=heads, layers=1, ffn=l.intermediate_size, vocab=64)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 7, l.hidden_size, generator=g)
    pos = torch.arange(7)[None].expand(2, -1)
    cos, sin = R.rope_cos_sin(pos, D, 10000.0, torch.float32)
    amask = torch.ones(2, 7, dtype=torch.long)
    mask = R.build_mask(amask, 7, torch.float32)
    want = R.llama_layer(x, W, 0, cfg, cos, sin, mask, None)
    plans = [ShardPlan(l.hidden_size, heads, D, l.intermediate_size, tp, r) for r in range(tp)]
    assert plans[0].heads_pad % tp == 0 and plans[0].heads_pad >= heads
    packed = [p.pack_layer(*(W[pre + k] for k in ("self_attn.q_proj.weight", "self_attn.k_proj.weight",
                                                   "self_attn.v_proj.weight", "self_attn.o_proj.weight",
                                                   "mlp.gate_proj.weight", "mlp.up_proj.weight",
                                                   "mlp.down_proj.weight"))) for p in plans]
    # lock-step simulation of the tp ranks with an in-process all-reduce
    ln1, ln2 = W[pre + "input_layernorm.weight"], W[pre + "post_attention_layernorm.weight"]
    import threading
    barrier = threading.Barrier(tp)
    box, outs = [None] * tp, [None] * tp

    def run(r):
        def allreduce(t):
            box[r] = t
            barrier.wait()
            s = sum(box)
            barrier.wait()
            return s
        outs[r] = sharded_layer_partial(x, packed[r], ln1, ln2, plans[r], cfg, cos, sin, mask, allreduce)

    ts = [threading.Thread(target=run, args=(r,)) for r in range(tp)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    for r in range(tp):
        torch.testing.assert_close(outs[r], want, rtol=1e-4, atol=1e-5)


def test_shard_plan_33b_shapes():
    for tp in (1, 2, 4, 8):
        p = ShardPlan(6656, 52, 128, 17920, tp, tp - 1)
        assert p.ffn_local * tp == 17920 and p.heads_local * tp >= 52
    p = ShardPlan(6656, 52, 128, 17920, 8, 7)
    assert p.heads_pad == 56 and p.heads_local == 7 and p.head_range == (49, 56)     # 3 real + 4 zero heads
    with pytest.raises(ValueError):
        ShardPlan(6656, 52, 128, 17920, 3, 0)


def test_apply_eos_padding_matches_hf_bookkeeping():
    from emu_amd.llama import apply_eos_padding
    ids = torch.tensor([[5, 2, 9, 9, 9], [7, 8, 2, 4, 4]])
    out = apply_eos_padding(ids, eos_id=2, pad_id=32000)
    assert out.tolist() == [[5, 2, 32000], [7, 8, 2]]
    ids = torch.tensor([[5, 6, 7], [7, 8, 2]])
    assert apply_eos_padding(ids, 2, 32000).tolist() == ids.tolist()


def test_synth_is_deterministic_and_per_tensor():
    v = CLIPVisionCfg(image_size=28, width=32, layers=1, head_width=16, mlp_ratio=2.0)
    l = LlamaCfg(hidden_size=32, intermediate_size=64, num_attention_heads=2, num_hidden_layers=1)
    shapes = synth.emu_param_shapes(v, l, 100)
    a = synth.synth_state_dict(shapes, seed=3)
    b = dict(synth.iter_synth(shapes, seed=3))
    assert list(a) == list(b) and all(torch.equal(a[k], b[k]) for k in a)
    one = synth.synth_tensor("project_up.weight", shapes["project_up.weight"], seed=3)
    assert torch.equal(one, a["project_up.weight"])
    assert abs(float(a["decoder.lm.model.norm.weight"].mean()) - 1.0) < 0.1
    assert set(k.split(".")[0] for k in a) == {"visual", "decoder", "project_up", "project_down"}


def test_config_mirrors_reference_defaults():
    v = CLIPVisionCfg()
    assert (v.width, v.layers, v.head_width, v.heads, v.mlp_hidden, v.tokens, v.n_query) == (1792, 64, 112, 16, 15360, 1025, 64)
    l = LlamaCfg.from_json(__import__("os").path.join(__import__("os").path.dirname(synth.__file__), "conf", "llama_config"))
    assert (l.hidden_size, l.num_attention_heads, l.num_hidden_layers, l.intermediate_size, l.head_dim) == (6656, 52, 60, 17920, 128)


def test_product_beam_search_host_logic_matches_reference(golden_dir, monkeypatch):
    """emu_amd.llama.LlamaEngine.beam_search_generate (the PRODUCT's host bookkeeping: 2N candidates, finished-beam merge,
    early-stop heuristic, per-beam KV replication and re-ordering) driven by a CPU stand-in engine: ids of the real
    reference for the default decoding mode (num_beams=5, max_new_tokens=10)."""
    import numpy as np
    from emu_amd import llama as L, ops
    from tests import tiny
    from tests.fake_engine import FakeEngine
    z = tiny.load(golden_dir, "generate_tiny.npz")
    v, l, vocab, W = tiny.weights_from(z)
    cfg = tiny.oracle_cfg(v, l, vocab)
    eng = FakeEngine(l, vocab, W, cfg.llama)
    monkeypatch.setattr(L, "BF16", torch.float32)
    monkeypatch.setattr(ops, "embed_gather", lambda ids, table, out=None: out.copy_(table[ids.long()]))
    t = lambda a: torch.from_numpy(np.asarray(a))
    ids, mask = t(z["ids2"]), t(z["mask2"])
    x = R.embed_tokens(ids, W)
    out = L.LlamaEngine.beam_search_generate(eng, x, mask, 5, 10, hf_semantics="5.x")     # the fixture comes from the installed library
    assert out.tolist() == z["beam2"].tolist()


@pytest.mark.parametrize("case,kw", [
    ("bs_sample", dict(num_beams=3, do_sample=True, top_k=40, top_p=0.9, temperature=0.7, max_new_tokens=8)),
    ("bs_sample_plain", dict(num_beams=4, do_sample=True, max_new_tokens=6)),
    ("bs_penalty", dict(num_beams=3, repetition_penalty=1.5, max_new_tokens=8)),
])
def test_product_beam_sampling_and_penalty_match_reference(golden_dir, monkeypatch, case, kw):
    """Beam-search multinomial sampling (num_beams > 1 with do_sample) and penalised beam search through the PRODUCT's
    beam_search_generate on the CPU stand-in engine: the ids the real reference produced for the same prompts under the
    same torch seed (oracle/make_golden_beam_sample.py; the draws come from torch's global CPU generator)."""
    import numpy as np
    from emu_amd import llama as L, ops
    from tests import tiny
    from tests.fake_engine import FakeEngine
    z = tiny.load(golden_dir, "generate_beam_sample_tiny.npz")
    v, l, vocab, W = tiny.weights_from(z)
    cfg = tiny.oracle_cfg(v, l, vocab)
    eng = FakeEngine(l, vocab, W, cfg.llama)
    monkeypatch.setattr(L, "BF16", torch.float32)
    monkeypatch.setattr(ops, "embed_gather", lambda ids, table, out=None: out.copy_(table[ids.long()]))
    t = lambda a: torch.from_numpy(np.asarray(a))
    x = R.embed_tokens(t(z["ids"]), W)
    kw = dict(kw)
    nb, n_new = kw.pop("num_beams"), kw.pop("max_new_tokens")
    torch.manual_seed(int(z["seed"]))
    # the fixture was drawn by the transformers installed here (5.x ordering of the sampling pipeline)
    out = L.LlamaEngine.beam_search_generate(eng, x, t(z["mask"]), nb, n_new, hf_semantics="5.x", **kw)
    assert out.tolist() == z[case].tolist()


def test_product_beam_modes_on_margin_fixtures(golden_dir, monkeypatch):
    """The fixtures the GPU tests hold the bf16 engine to (tests/golden/generate_margin_tiny.npz, oracle/make_golden_r3.py):
    penalised 3-beam search on a ragged batch and the default 5-beam search with an image -- ids of the REAL reference,
    reproduced by the product's host logic in fp32 with every pruning margin >= 0.08 nat."""
    import numpy as np
    from emu_amd import llama as L, ops
    from tests import tiny
    from tests.fake_engine import FakeEngine
    z = tiny.load(golden_dir, "generate_margin_tiny.npz")
    v, l, vocab, W = tiny.weights_from(z)
    cfg = tiny.oracle_cfg(v, l, vocab)
    monkeypatch.setattr(L, "BF16", torch.float32)
    monkeypatch.setattr(ops, "embed_gather", lambda ids, table, out=None: out.copy_(table[ids.long()]))
    t = lambda a: torch.from_numpy(np.asarray(a))
    tr = {}
    out = L.LlamaEngine.beam_search_generate(FakeEngine(l, vocab, W, cfg.llama), R.embed_tokens(t(z["pen_ids"]), W), t(z["pen_mask"]),
                                             3, 8, repetition_penalty=1.5, trace=tr, hf_semantics="5.x")
    assert out.tolist() == z["pen_new"].tolist() and tr["margin"] >= 0.08, tr
    assert not bool(t(z["pen_mask"]).all())
    ids = t(z["b5_ids"])
    e = R.encode_image(t(z["image"]), W, cfg)
    x = R.scatter_image_embeds(R.embed_tokens(ids, W), ids, torch.nn.functional.linear(e.reshape(-1, e.shape[-1]), W["project_up.weight"]))
    tr = {}
    out = L.LlamaEngine.beam_search_generate(FakeEngine(l, vocab, W, cfg.llama), x, t(z["b5_mask"]), 5, int(z["b5_n_new"]), trace=tr,
                                             hf_semantics="5.x")
    assert out.tolist() == z["b5_new"].tolist() and tr["margin"] >= 0.08, tr


class _Hyps431:
    """``BeamHypotheses`` of transformers 4.31 (generation/beam_search.py), early_stopping=False: ``add`` scores a hypothesis by
    sum_logprobs / len(hyp) ** length_penalty and keeps the num_beams best; ``is_done`` compares the worst kept score with
    best_sum_logprobs / cur_len ** length_penalty."""

    def __init__(self, nb, lp):
        self.nb, self.lp, self.beams, self.worst = nb, lp, [], 1e9

    def add(self, hyp, sum_logprobs):
        score = sum_logprobs / (len(hyp) ** self.lp)
        if len(self.beams) < self.nb or score > self.worst:
            self.beams.append((score, list(hyp)))
            if len(self.beams) > self.nb:
                srt = sorted((sc, i) for i, (sc, _) in enumerate(self.beams))
                del self.beams[srt[0][1]]
                self.worst = srt[1][0]
            else:
                self.worst = min(score, self.worst)

    def is_done(self, best_sum_logprobs, cur_len):
        if len(self.beams) < self.nb:
            return False
        return self.worst >= best_sum_logprobs / cur_len ** self.lp


def _beam_431(x, mask, W, cfg, nb, n_new, do_sample=False, temperature=None, top_k=None, top_p=None, eos_id=2, pad_id=32000,
              length_penalty=-1.0, min_len=1):
    """``GenerationMixin.beam_search`` / ``beam_sample`` + ``BeamSearchScorer.process`` / ``finalize`` as transformers 4.31 (the
    release the reference pins) states them, written per hypothesis with python lists and full recomputation of the decoder -- an
    independent statement of what the product's hf_semantics="4.31" mode claims:
      * sampling order: log_softmax -> MinLength -> + beam score -> temperature -> top-k -> top-p (min_tokens_to_keep 2) ->
        multinomial(2N) -> sort by score -> scorer; every beam starts at score 0 (beam_search: (0, -1e9, ...));
      * scorer: with ``inputs_embeds`` the ids start EMPTY, so a hypothesis that ends with EOS is added as the cur tokens ahead of
        the EOS (``hyp.shape[-1] ** length_penalty`` = cur ** lp), only while its rank is < N; ``is_done`` gets the best of all 2N
        candidate scores and cur_len = cur + 1; a done prompt adds nothing more; when the length limit ends the loop ``finalize``
        adds every running beam (L tokens) unless the prompt is done; the result is the best kept hypothesis (+ EOS if it fits).
    Restated from memory of that release -- it cannot be installed here: this pins the product to THIS statement, not to the library."""
    B, S, _ = x.shape
    V = W["decoder.lm.lm_head.weight"].shape[0]
    seqs = [[[] for _ in range(nb)] for _ in range(B)]
    scores = torch.zeros(B, nb)
    if not do_sample:
        scores[:, 1:] = -1.0e9
    hyps = [_Hyps431(nb, length_penalty) for _ in range(B)]
    done = [False] * B
    for cur in range(n_new):
        nxt_seqs, nxt_scores = [], []
        for b in range(B):
            if done[b]:                                                 # process(): a done prompt is padded
                nxt_seqs.append([sq + [pad_id] for sq in seqs[b]]); nxt_scores.append([0.0] * nb)
                continue
            rows = []
            for j in range(nb):
                xe = torch.cat((x[b:b + 1], R.embed_tokens(torch.tensor([seqs[b][j]], dtype=torch.long).view(1, -1), W)), dim=1) \
                    if seqs[b][j] else x[b:b + 1]
                me = torch.cat((mask[b:b + 1], torch.ones(1, len(seqs[b][j]), dtype=mask.dtype)), dim=1)
                pos = (me.long().cumsum(-1) - 1).masked_fill(me == 0, 1)
                h = R.llama_model(xe, me, W, cfg.llama, position_ids=pos)
                lp = torch.log_softmax(torch.nn.functional.linear(h[0, -1], W["decoder.lm.lm_head.weight"]).float(), -1)
                if cur < min_len:
                    lp[eos_id] = -float("inf")
                rows.append(lp + scores[b, j])
            acc = torch.stack(rows)                                     # [nb, V] accumulated
            if do_sample:
                if temperature is not None and temperature != 1.0:
                    acc = acc / temperature
                if top_k:
                    kth = torch.topk(acc, max(top_k, 2))[0][:, -1:]
                    acc = acc.masked_fill(acc < kth, -float("inf"))
                if top_p is not None and top_p < 1.0:
                    srt, idx = torch.sort(acc, descending=False)
                    rm = srt.softmax(-1).cumsum(-1) <= (1 - top_p)
                    rm[:, -2:] = False
                    acc = acc.masked_fill(rm.scatter(1, idx, rm), -float("inf"))
                flat = acc.reshape(-1)
                draw = torch.multinomial(torch.softmax(flat, -1), num_samples=2 * nb)
                sc = flat[draw]
                sc, order = torch.sort(sc, descending=True)
                draw = draw[order]
            else:
                sc, draw = torch.topk(acc.reshape(-1), 2 * nb)
            run = []
            for rank in range(2 * nb):                                  # BeamSearchScorer.process
                j, tok = int(draw[rank]) // V, int(draw[rank]) % V
                if tok == eos_id:
                    if rank >= nb:
                        continue
                    hyps[b].add(seqs[b][j], float(sc[rank]))            # input_ids of the beam: WITHOUT the EOS
                else:
                    run.append((float(sc[rank]), seqs[b][j] + [tok]))
                if len(run) == nb:
                    break
            nxt_seqs.append([r[1] for r in run]); nxt_scores.append([r[0] for r in run])
            done[b] = done[b] or hyps[b].is_done(float(sc.max()), cur + 1)
        seqs, scores = nxt_seqs, torch.tensor(nxt_scores)
        if all(done):
            break
    for b in range(B):                                                  # BeamSearchScorer.finalize
        if done[b]:
            continue
        for j in range(nb):
            hyps[b].add(seqs[b][j], float(scores[b, j]))
    best = []
    for b in range(B):
        sc, hyp = sorted(hyps[b].beams, key=lambda t_: t_[0])[-1]
        best.append(hyp + ([eos_id] if len(hyp) < n_new else []))
    n = max(len(s_) for s_ in best)
    return torch.tensor([s_ + [pad_id] * (n - len(s_)) for s_ in best])


def _beam_sample_431(x, mask, W, cfg, nb, n_new, temperature, top_k, top_p, **kw):
    return _beam_431(x, mask, W, cfg, nb, n_new, do_sample=True, temperature=temperature, top_k=top_k, top_p=top_p, **kw)


@pytest.mark.parametrize("kw", [dict(temperature=0.7, top_k=40, top_p=0.9), dict(temperature=None, top_k=None, top_p=None)])
def test_product_beam_sample_431_ordering(golden_dir, monkeypatch, kw):
    """hf_semantics="4.31" (the default: the reference pins transformers 4.31.0): the product's vectorised beam sampling
    equals a per-hypothesis restatement of that release's pipeline order under the same seed (same number and order of
    multinomial calls: one per prompt row per step would differ, so the restatement is run at batch size 1)."""
    import numpy as np
    from emu_amd import llama as L, ops
    from tests import tiny
    from tests.fake_engine import FakeEngine
    z = tiny.load(golden_dir, "generate_beam_sample_tiny.npz")
    v, l, vocab, W = tiny.weights_from(z)
    cfg = tiny.oracle_cfg(v, l, vocab)
    monkeypatch.setattr(L, "BF16", torch.float32)
    monkeypatch.setattr(ops, "embed_gather", lambda ids, table, out=None: out.copy_(table[ids.long()]))
    t = lambda a: torch.from_numpy(np.asarray(a))
    ids, mask = t(z["ids"])[1:2], t(z["mask"])[1:2]                   # the unpadded row
    x = R.embed_tokens(ids, W)
    torch.manual_seed(99)
    got = L.LlamaEngine.beam_search_generate(FakeEngine(l, vocab, W, cfg.llama), x, mask, 3, 6, do_sample=True, **kw)
    torch.manual_seed(99)
    want = _beam_sample_431(x, mask, W, cfg, 3, 6, **kw)
    assert got.tolist() == want.tolist()
    torch.manual_seed(99)
    new = L.LlamaEngine.beam_search_generate(FakeEngine(l, vocab, W, cfg.llama), x, mask, 3, 6, do_sample=True,
                                             hf_semantics="5.x", **kw)
    assert new.shape[0] == 1                                            # the other ordering runs; it need not agree


def test_product_beam_search_431_scorer(golden_dir, monkeypatch):
    """Deterministic beam search under hf_semantics="4.31" (the default of EmuModel: the reference pins transformers 4.31.0):
    the product's vectorised bookkeeping equals the per-hypothesis restatement of that release's BeamSearchScorer /
    BeamHypotheses -- on prompts where hypotheses END EARLY, which is where 4.31 and 5.x differ (an EOS hypothesis scored over
    cur instead of cur + 1 tokens, is_done on the best of all candidates, finalize of the running beams).  Random-init weights
    never emit the real EOS, so tokens the searches do produce are declared EOS in turn; with length_penalty -1 (the reference's
    default), 0 and 1.  At least one case must differ from the 5.x conventions, or the test would not see the difference."""
    import numpy as np
    from emu_amd import llama as L, ops
    from tests import tiny
    from tests.fake_engine import FakeEngine
    z = tiny.load(golden_dir, "generate_tiny.npz")
    v, l, vocab, W = tiny.weights_from(z)
    cfg = tiny.oracle_cfg(v, l, vocab)
    monkeypatch.setattr(L, "BF16", torch.float32)
    monkeypatch.setattr(ops, "embed_gather", lambda ids, table, out=None: out.copy_(table[ids.long()]))
    t = lambda a: torch.from_numpy(np.asarray(a))
    ids, mask = t(z["ids2"]), t(z["mask2"])
    x = R.embed_tokens(ids, W)
    nb, n_new = 3, 6
    base = L.LlamaEngine.beam_search_generate(FakeEngine(l, vocab, W, cfg.llama), x, mask, nb, n_new, hf_semantics="5.x")
    eos_choices = sorted({int(base[0, 1]), int(base[0, 3]), int(base[-1, 2])})
    differs = 0
    for eos in eos_choices:
        for lp in (-1.0, 0.0, 1.0):
            for b in range(ids.shape[0]):                               # per prompt: rows of different progress never interact
                xb, mb = x[b:b + 1], mask[b:b + 1]
                got = L.LlamaEngine.beam_search_generate(FakeEngine(l, vocab, W, cfg.llama), xb, mb, nb, n_new, length_penalty=lp,
                                                         eos_id=eos, hf_semantics="4.31")
                want = _beam_431(xb, mb, W, cfg, nb, n_new, eos_id=eos, length_penalty=lp)
                assert got.tolist() == want.tolist(), (eos, lp, b, got.tolist(), want.tolist())
                new = L.LlamaEngine.beam_search_generate(FakeEngine(l, vocab, W, cfg.llama), xb, mb, nb, n_new, length_penalty=lp,
                                                         eos_id=eos, hf_semantics="5.x")
                differs += int(new.tolist() != got.tolist())
    assert differs > 0
    # a batch of both prompts = the rows alone (padded to the longer result)
    got2 = L.LlamaEngine.beam_search_generate(FakeEngine(l, vocab, W, cfg.llama), x, mask, nb, n_new, eos_id=eos_choices[0],
                                              hf_semantics="4.31")
    want2 = _beam_431(x, mask, W, cfg, nb, n_new, eos_id=eos_choices[0])
    assert got2.tolist() == want2.tolist()


def _contrastive_uncached(x, mask, W, cfg, n_new, alpha, k, eos_id=2, pad_id=32000):
    """Contrastive search by full recomputation (no KV cache, no candidate batching): an independent statement of the
    algorithm (transformers 4.31 ``contrastive_search`` / ``_ranking_fast``) on the oracle's decoder, left-padding positions
    left out of the similarity like the product does."""
    B, S, _ = x.shape
    lmw, nw = W["decoder.lm.lm_head.weight"], W["decoder.lm.model.norm.weight"]
    seqs = [x[b:b + 1] for b in range(B)]
    masks = [mask[b:b + 1] for b in range(B)]
    out = torch.full((B, n_new), pad_id, dtype=torch.int64)
    done = [False] * B
    for step in range(n_new):
        for b in range(B):
            if done[b]:
                continue
            h = R.llama_model(seqs[b], masks[b], W, cfg.llama)                      # final-norm hidden states [1, L, H]
            logit = torch.nn.functional.linear(h[0, -1], lmw).float()
            if step < 1:
                logit[eos_id] = -float("inf")
            p, ids = torch.topk(torch.softmax(logit, -1), k)
            ok = masks[b][0].bool()
            best, best_sc = None, None
            for j in range(k):
                xe = torch.cat((seqs[b], R.embed_tokens(ids[j].view(1, 1), W)), dim=1)
                me = torch.cat((masks[b], torch.ones(1, 1, dtype=masks[b].dtype)), dim=1)
                hh = R.llama_model(xe, me, W, cfg.llama)[0]
                cos = torch.nn.functional.cosine_similarity(hh[:-1][ok], hh[-1:], dim=-1)
                sc = (1 - alpha) * float(p[j]) - alpha * float(cos.max())
                if best is None or sc > best_sc:
                    best, best_sc = j, sc
            t = int(ids[best])
            out[b, step] = t
            seqs[b] = torch.cat((seqs[b], R.embed_tokens(torch.tensor([[t]]), W)), dim=1)
            masks[b] = torch.cat((masks[b], torch.ones(1, 1, dtype=masks[b].dtype)), dim=1)
            done[b] = t == eos_id
        if all(done):
            return out[:, :step + 1]
    return out


def test_product_contrastive_search(golden_dir, monkeypatch):
    """LlamaEngine.contrastive_generate (candidate rows sharing the prompt KV, sibling KV copy, hidden-state context) on
    the CPU stand-in engine: penalty_alpha = 0 and top_k = 1 must reproduce the REAL reference's greedy ids, and a real
    setting (alpha 0.6, k 4) must equal an uncached full-recomputation statement of the algorithm."""
    import numpy as np
    from emu_amd import llama as L, ops
    from tests import tiny
    from tests.fake_engine import FakeEngine
    z = tiny.load(golden_dir, "generate_tiny.npz")
    v, l, vocab, W = tiny.weights_from(z)
    cfg = tiny.oracle_cfg(v, l, vocab)
    monkeypatch.setattr(L, "BF16", torch.float32)
    monkeypatch.setattr(ops, "embed_gather", lambda ids, table, out=None: out.copy_(table[ids.long()]))
    t = lambda a: torch.from_numpy(np.asarray(a))
    ids, mask = t(z["ids2"]), t(z["mask2"])
    x = R.embed_tokens(ids, W)
    new_eng = lambda: FakeEngine(l, vocab, W, cfg.llama)
    assert L.LlamaEngine.contrastive_generate(new_eng(), x, mask, 6, 0.0, 4).tolist() == z["new2"].tolist()
    assert L.LlamaEngine.contrastive_generate(new_eng(), x, mask, 6, 0.6, 1).tolist() == z["new2"].tolist()
    got = L.LlamaEngine.contrastive_generate(new_eng(), x, mask, 5, 0.6, 4)
    want = _contrastive_uncached(x, mask, W, cfg, 5, 0.6, 4)
    assert got.tolist() == want.tolist()
    assert got.tolist() != z["new2"][:, :5].tolist()          # the penalty actually changes the choice on this prompt


def test_emu1_lora_merge_both_peft_layouts():
    """Emu1 instruct checkpoints carry peft LoRA adapters (Emu1/inference.py:40-51); the loader folds them into the base
    matrices: W' = W + (alpha / r) * B @ A, for both peft key layouts, and leaves adapter-free dicts untouched."""
    import pytest
    from emu_amd.emu1 import merge_lora_state_dict
    g = torch.Generator().manual_seed(0)
    W = torch.randn(32, 24, generator=g)
    A = torch.randn(4, 24, generator=g)
    B = torch.randn(32, 4, generator=g)
    other = torch.randn(5, generator=g)
    want = W + (8.0 / 4) * (B @ A)
    pre = "decoder.lm.base_model.model.model.layers.0.self_attn.q_proj."
    old = {pre + "weight": W, pre + "lora_A.weight": A, pre + "lora_B.weight": B, "ln_visual.weight": other}
    new = {pre + "base_layer.weight": W, pre + "lora_A.default.weight": A, pre + "lora_B.default.weight": B,
           "ln_visual.weight": other}
    for sd in (old, new):
        out = merge_lora_state_dict(sd, r=4, alpha=8.0)
        assert set(out) == {"decoder.lm.model.layers.0.self_attn.q_proj.weight", "ln_visual.weight"}
        assert torch.allclose(out["decoder.lm.model.layers.0.self_attn.q_proj.weight"], want, atol=1e-5)
        assert out["ln_visual.weight"] is other
    plain = {"decoder.lm.model.norm.weight": other}
    assert merge_lora_state_dict(plain) == plain
    with pytest.raises(RuntimeError):
        merge_lora_state_dict({pre + "weight": W, pre + "lora_A.weight": A}, r=4, alpha=8.0)
    with pytest.raises(RuntimeError):
        merge_lora_state_dict({pre + "lora_A.weight": A, pre + "lora_B.weight": B}, r=4, alpha=8.0)


def test_committed_bench_line_matches_the_contract():
    """The bench line committed under profiles/ (the default `python bench.py` on an MI355X) carries every field of the
    driver's contract, with the roofline / cpu_baseline objects and consistent arithmetic."""
    import glob
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(root, "profiles", "r01_bench_tp1_v*_full.json")),
                   key=lambda p: int(os.path.basename(p).split("_v")[1].split("_")[0]))
    assert files, "no committed full bench line"
    d = json.load(open(files[-1]))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "tokens/s" and d["higher_is_better"] is True and d["vs_baseline"] is None and d["dtype"] == "bf16"
    assert d["n_gpus"] == 1 and d["scaling"] == "strong" and d["data"] == "synthetic"
    assert "workload" in d["config"] and d["config"]["valid"] is True and "model" not in d["config"]
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) / d["value"] < 1e-6
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["traffic"] is not None and 0.99 < r["traffic"] / r["bytes_per_launch"] < 1.05
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "tokens/s" and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    assert d["denoise"]["roofline"]["bound"] == "mfma" and d["denoise"]["unit"] == "steps/s"


def test_logits_processing_matches_transformers_processors():
    """The sampling path's logits pipeline (repetition penalty -> min length -> temperature -> top-k -> top-p) against
    transformers' own LogitsProcessor / LogitsWarper classes, element for element."""
    from transformers.generation.logits_process import (MinLengthLogitsProcessor, RepetitionPenaltyLogitsProcessor,
                                                        TemperatureLogitsWarper, TopKLogitsWarper, TopPLogitsWarper)
    from emu_amd.llama import process_logits
    g = torch.Generator().manual_seed(5)
    B, V, eos = 3, 500, 2
    for trial in range(6):
        scores = torch.randn(B, V, generator=g) * 3
        n_gen = [0, 1, 4, 9, 2, 7][trial]
        gen = torch.randint(3, V, (B, n_gen), generator=g)
        temp, tk, tp, rp = [(0.7, 40, 0.9, 1.3), (1.0, None, 0.8, 1.0), (1.5, 5, None, 2.0), (0.3, 500, 0.5, 1.1),
                            (None, None, None, 1.2), (2.0, 1, 0.99, 1.0)][trial]
        min_len = 3
        want = scores.clone()
        ids = gen                                               # with inputs_embeds the id sequence starts empty
        if rp != 1.0 and n_gen > 0:
            want = RepetitionPenaltyLogitsProcessor(rp)(ids, want)
        want = MinLengthLogitsProcessor(min_len, eos)(ids, want)
        if temp is not None and temp != 1.0:
            want = TemperatureLogitsWarper(temp)(ids, want)
        if tk:
            want = TopKLogitsWarper(tk)(ids, want)
        if tp is not None and tp < 1.0:
            want = TopPLogitsWarper(tp)(ids, want)
        got = process_logits(scores.clone(), gen, n_gen < min_len, eos, True, temp, tk, tp, rp)
        assert torch.equal(torch.isinf(got), torch.isinf(want)), trial
        fin = ~torch.isinf(want)
        assert torch.allclose(got[fin], want[fin], rtol=0, atol=0), trial
    # greedy: warpers are not applied
    s0 = torch.randn(2, 50, generator=g)
    assert torch.equal(process_logits(s0.clone(), torch.zeros(2, 0, dtype=torch.long), False, eos, False, 0.5, 3, 0.5, 1.0), s0)


def test_no_repeat_ngram_matches_transformers_processor():
    """ban_repeated_ngrams (process_logits' no_repeat_ngram_size) against transformers' NoRepeatNGramLogitsProcessor, element for
    element, on id rows with planted repeats; and its place in the pipeline: after the repetition penalty, ahead of MinLength."""
    from transformers.generation.logits_process import (MinLengthLogitsProcessor, NoRepeatNGramLogitsProcessor,
                                                        RepetitionPenaltyLogitsProcessor)
    from emu_amd.llama import ban_repeated_ngrams, process_logits
    g = torch.Generator().manual_seed(8)
    V = 40
    for n in (1, 2, 3, 4):
        for cur in (0, 1, 2, 3, 7, 15):
            gen = torch.randint(3, 9, (4, cur), generator=g)             # a tiny alphabet: repeats are everywhere
            scores = torch.randn(4, V, generator=g)
            want = NoRepeatNGramLogitsProcessor(n)(gen, scores.clone())
            got = ban_repeated_ngrams(scores.clone(), gen, n)
            assert torch.equal(torch.isinf(got), torch.isinf(want)), (n, cur)
            assert torch.equal(got[~torch.isinf(want)], want[~torch.isinf(want)])
    gen = torch.randint(3, 9, (3, 6), generator=g)
    scores = torch.randn(3, V, generator=g)
    want = MinLengthLogitsProcessor(9, 2)(gen, NoRepeatNGramLogitsProcessor(2)(gen, RepetitionPenaltyLogitsProcessor(1.4)(gen, scores.clone())))
    got = process_logits(scores.clone(), gen, True, 2, False, repetition_penalty=1.4, no_repeat_ngram_size=2)
    assert torch.equal(torch.isinf(got), torch.isinf(want)) and torch.equal(got[~torch.isinf(want)], want[~torch.isinf(want)])


def test_product_ngram_ban_and_several_returned_sequences_match_reference(golden_dir, monkeypatch):
    """no_repeat_ngram_size = 2 with num_return_sequences = 2 through the PRODUCT's beam search on the CPU stand-in engine: ids of
    the real reference (Emu2: EmuModel.generate forwards **kwargs, emu.py:175,228; fixture generate_margin_tiny.npz), 4 rows
    prompt-major; the library's argument checks are mirrored."""
    import numpy as np
    from emu_amd import llama as L, ops
    from tests import tiny
    from tests.fake_engine import FakeEngine
    z = tiny.load(golden_dir, "generate_margin_tiny.npz")
    v, l, vocab, W = tiny.weights_from(z)
    cfg = tiny.oracle_cfg(v, l, vocab)
    monkeypatch.setattr(L, "BF16", torch.float32)
    monkeypatch.setattr(ops, "embed_gather", lambda ids, table, out=None: out.copy_(table[ids.long()]))
    t = lambda a: torch.from_numpy(np.asarray(a))
    x, mask = R.embed_tokens(t(z["pen_ids"]), W), t(z["pen_mask"])
    out = L.LlamaEngine.beam_search_generate(FakeEngine(l, vocab, W, cfg.llama), x, mask, 3, 8, no_repeat_ngram_size=2,
                                             num_return_sequences=2, hf_semantics="5.x")
    assert out.tolist() == z["ngram_new"].tolist() and out.shape[0] == 4
    for row in out.tolist():                                             # no bigram twice
        big = list(zip(row, row[1:]))
        assert len(big) == len(set(big)), row
    with pytest.raises(ValueError):
        L.LlamaEngine.beam_search_generate(FakeEngine(l, vocab, W, cfg.llama), x, mask, 3, 8, num_return_sequences=4)
    with pytest.raises(ValueError):
        L.LlamaEngine.sample_generate(FakeEngine(l, vocab, W, cfg.llama), x, mask, 4, do_sample=False, num_return_sequences=2)


def test_emu1_num_captions_and_ngram_ban_match_real_reference(golden_dir, monkeypatch):
    """Emu1's ``num_captions`` / ``no_repeat_ngram_size`` (Emu1/models/modeling_emu.py:110,115,173,176): the product's beam search
    on the CPU stand-in engine, fed the oracle's visual embeddings, against the two captions of the REAL Emu1 class
    (tests/golden/emu1_generate_tiny.npz: 5 beams, 8 tokens, bigram ban, length_penalty 0, whole model in bf16 under autocast).
    The BEST caption must be the reference's.  The runner-up is decided by a pruning margin of 0.003 nat on this prompt (measured
    with ``trace``), which the reference's autocast arithmetic and a plain bf16 / fp32 evaluation resolve differently, so for it
    the library's guarantees are checked instead: distinct from the best, same length, no bigram twice, and the exact-id pin of
    this mode is the margin-screened Emu2 fixture above."""
    import numpy as np
    from emu_amd import llama as L, ops
    from oracle import emu1_ref as E
    from tests import tiny
    from tests.fake_engine import FakeEngine
    z = tiny.load(golden_dir, "emu1_generate_tiny.npz")
    v, t5, l, vocab, W, cfg = tiny.emu1_generate_from(z)
    Wb = R.cast_weights(W, torch.bfloat16)
    ids, mask, img = (torch.from_numpy(np.asarray(z[k])) for k in ("ids", "mask", "image"))
    x = R.embed_tokens(ids, Wb)
    e = E.encode_image(img.to(torch.bfloat16), Wb, cfg)
    x = R.scatter_image_embeds(x, ids, e.reshape(-1, e.shape[-1]))
    monkeypatch.setattr(ops, "embed_gather", lambda i_, table, out=None: out.copy_(table[i_.long()]))
    eng = FakeEngine(l, vocab, Wb, cfg.llama, dtype=torch.bfloat16)      # bf16 arithmetic, as the reference runs the model
    tr = {}
    out = L.LlamaEngine.beam_search_generate(eng, x, mask, 5, 8, length_penalty=0.0, no_repeat_ngram_size=2, num_return_sequences=2,
                                             trace=tr, hf_semantics="5.x")     # Emu1 leaves transformers unpinned
    want = z["beam_cap2_ngram2"].tolist()
    assert out.shape == (2, 8) and out[0].tolist() == want[0]
    if tr["margin"] >= 0.05:
        assert out[1].tolist() == want[1]
    assert out[1].tolist() != out[0].tolist()
    for row in out.tolist() + want:
        big = list(zip(row, row[1:]))
        assert len(big) == len(set(big)), row


def test_stream_handle_refuses_another_device_without_switching(monkeypatch):
    """``ops.stream`` is a getter: an operand on another GPU than the current one is refused loudly, the current device is never
    switched behind the caller's back (one process drives one GPU; EmuHipContext is the place that selects it)."""
    from emu_amd import ops
    switched = []
    monkeypatch.setattr(torch.cuda, "current_device", lambda: 0)
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: switched.append(d))
    with pytest.raises(ValueError, match="current device"):
        ops.stream(torch.device("cuda", 1))
    assert switched == []


def _eos_cases(z):
    return sorted(k[:-4] for k in z if k.endswith("_eos"))


def _prompt_x(z, key, W, cfg):
    import numpy as np
    t = lambda a: torch.from_numpy(np.asarray(a))
    ids = t(z[key + "_ids"])
    x = R.embed_tokens(ids, W)
    if bool(z[key + "_has_image"]):
        e = R.encode_image(t(z["image"]), W, cfg, None)
        e = torch.nn.functional.linear(e.reshape(-1, e.shape[-1]), W["project_up.weight"])
        x = R.scatter_image_embeds(x, ids, e)
    return ids, t(z[key + "_mask"]), x


def test_product_beam_search_finished_hypotheses_real_reference(golden_dir, monkeypatch):
    """The reference's default decoding mode (5 beams, length_penalty -1) on prompts where hypotheses END ON EOS before the length
    limit, against the ids of the REAL reference (tests/golden/generate_beam_eos_tiny.npz, oracle/make_golden_beam_eos.py: the
    installed transformers 5.x with ``eos_token_id`` forwarded through ``generate``'s **kwargs).  Pins, under
    hf_semantics="5.x": the scorer's finished-hypothesis branch (length penalty of a short hypothesis, early stop, EOS / PAD
    fill of the returned rows of a ragged batch) and that min_length = 1 enforces nothing with inputs_embeds (EOS first).
    And the "4.31" conventions differ from these ids only where the restatement of that release says they must: on the cases
    where EOS would be the FIRST token (4.31 masks it: min_length counts generated tokens there)."""
    from emu_amd import llama as L, ops
    from tests import tiny
    from tests.fake_engine import FakeEngine
    z = tiny.load(golden_dir, "generate_beam_eos_tiny.npz")
    v, l, vocab, W = tiny.weights_from(z)
    cfg = tiny.oracle_cfg(v, l, vocab)
    monkeypatch.setattr(L, "BF16", torch.float32)
    monkeypatch.setattr(ops, "embed_gather", lambda ids, table, out=None: out.copy_(table[ids.long()]))
    n_first, n_cases = 0, 0
    for key in _eos_cases(z):
        ids, mask, x = _prompt_x(z, key, W, cfg)
        nb, n_new, eos = int(z[key + "_nb"]), int(z[key + "_n_new"]), int(z[key + "_eos"])
        want = z[key + "_out"]
        S = ids.shape[1]
        got = L.LlamaEngine.beam_search_generate(FakeEngine(l, vocab, W, cfg.llama), x, mask, nb, n_new, min_len=max(1 - S, 0),
                                                 eos_id=eos, hf_semantics="5.x")
        assert got.tolist() == want.tolist(), key
        # 4.31: product == restatement; and == the 5.x ids unless EOS leads a returned row
        got431 = L.LlamaEngine.beam_search_generate(FakeEngine(l, vocab, W, cfg.llama), x, mask, nb, n_new, min_len=1, eos_id=eos,
                                                    hf_semantics="4.31")
        want431 = torch.cat([_beam_431(x[b:b + 1], mask[b:b + 1], W, cfg, nb, n_new, eos_id=eos) for b in range(x.shape[0])]) \
            if x.shape[0] == 1 else None
        if want431 is not None:
            assert got431.tolist() == want431.tolist(), key
        eos_first = bool((torch.from_numpy(want)[:, 0] == eos).any())
        n_first += eos_first
        n_cases += 1
        if eos_first:
            assert not bool((got431[:, 0] == eos).any()), key          # 4.31 never returns EOS as the first token (min_length 1)
    assert n_cases >= 5 and 0 < n_first < n_cases                         # both kinds of case are present


def test_effective_min_len_per_release():
    """min_length under inputs_embeds: transformers 5.x subtracts the prompt length (GenerationMixin._prepare_generated_length), 4.31
    counts generated tokens (restated); pinned by the real library's EOS-first rows in tests/golden/generate_beam_eos_tiny.npz."""
    assert R.effective_min_len(1, 12, "5.x") == 0 and R.effective_min_len(1, 12, "4.31") == 1
    assert R.effective_min_len(20, 12, "5.x") == 8 and R.effective_min_len(0, 12, "4.31") == 0
    import inspect
    try:
        from transformers.generation import utils
        src = inspect.getsource(utils.GenerationMixin._prepare_generated_length)
    except Exception:
        pytest.skip("transformers internals not importable")
    if "min_length - inputs_tensor.shape[1]" not in src.replace("generation_config.", ""):
        pytest.skip("the installed transformers words the correction differently")


def test_default_beam_mode_is_announced_and_its_cur0_eos_case_is_refused():
    """The DEFAULT decoding mode (emu.py:163-172: 5 beams, length_penalty -1) runs under hf_semantics='4.31', a restatement with no
    library vector: EmuModel.generate_ids says so once per process, and refuses the one case the restatement does not define -- an
    EOS hypothesis that ends at cur == 0 (4.31 scores it over 0 ** length_penalty; reachable only with min_len < 1) -- before any
    device work.  '5.x' (pinned by tests/golden/generate_beam_eos_tiny.npz) takes min_len 0."""
    import types
    import warnings
    from emu_amd.emu import EmuModel
    calls = []

    class _LM:
        def beam_search_generate(self, x, mask, nb, max_new, min_len, lp, **kw):
            calls.append((nb, min_len, kw["hf_semantics"]))
            return torch.zeros(x.shape[0], 1, dtype=torch.long)

    me = types.SimpleNamespace(hf_semantics="4.31", n_query=4, v_query=4, use_graph=False, decoder=types.SimpleNamespace(lm=_LM()),
                               _prompt_embeds=lambda ids, image, nq, tok, embeds=None: torch.zeros(ids.numel(), 8))
    ids, mask = torch.ones(1, 5, dtype=torch.long), torch.ones(1, 5, dtype=torch.long)
    with pytest.raises(ValueError, match="min_len >= 1"):
        EmuModel.generate_ids(me, ids, mask, num_beams=5, min_len=0)
    assert not calls                                                    # refused before the prompt was even embedded
    EmuModel._warned_431 = False
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        EmuModel.generate_ids(me, ids, mask, num_beams=5)
        EmuModel.generate_ids(me, ids, mask, num_beams=5)
    assert sum("RESTATEMENT" in str(x.message) for x in w) == 1          # once per process
    assert calls == [(5, 1, "4.31")] * 2
    EmuModel.generate_ids(me, ids, mask, num_beams=5, min_len=0, hf_semantics="5.x")
    assert calls[-1] == (5, 0, "5.x")


def test_round6_bench_line_has_no_stale_traffic_and_every_leg_its_objects():
    """The round-6 line (`python bench.py` on the final sources, profiles/r06_bench_tp1_final.json): the contract's fields, `roofline.traffic` and
    `prefill_roofline.traffic` from PMC passes that hash-match the sources they were quoted for (neither None nor STALE), and `roofline` +
    `cpu_baseline` on the headline, the denoise leg and the VAE / generate_image legs."""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = json.load(open(os.path.join(root, "profiles", "r06_bench_tp1_final.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["dtype"] == "bf16" and d["vs_baseline"] is None and d["config"]["valid"] is True
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) / d["value"] < 1e-6
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["traffic"] is not None and "STALE" not in r["traffic_source"] and "r06_gemv_pmc_traffic" in r["traffic_source"]
    assert 0.99 < r["traffic"] / r["bytes_per_launch"] < 1.05 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    pr = d["extra"]["prefill_roofline"]
    assert pr["traffic"] is not None and "STALE" not in pr["traffic_source"] and "r06_prefill_gemm_pmc_traffic" in pr["traffic_source"]
    assert d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["cores"] >= 1
    assert d["denoise"]["roofline"]["bound"] == "mfma" and d["denoise"]["cpu_baseline"]["value"] > 0
    for leg, bound in (("vae_decode", "mfma"), ("generate_image", "hbm")):
        L = d["legs"][leg]
        assert L["roofline"]["bound"] == bound and 0 < L["roofline"]["frac"] < 1, leg
        assert L["cpu_baseline"]["value"] > 0 and L["cpu_baseline"]["sample"], leg
    assert abs(d["legs"]["vae_decode"]["roofline"]["flops"] - 10.47e12) < 0.01e12
