"""GPU tests of the one-launch beam-search step (csrc/beam.hip, emu_beam_step_bf16) against the torch pipeline of
emu_amd.llama.beam_search_generate, which restates transformers' vectorised beam search and is itself pinned to the real reference's
ids (tests/test_host_logic.py, tests/test_gpu_model.py).  Run on an MI355X with `-m gpu`."""
import pytest
import torch

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16
NEG = -1.0e9


def topk_stable(x, k):
    """top-k with the kernel's tie rule: among equal values the lower index first.  (torch.topk leaves the order of ties unspecified,
    and ties are the rule here, not the exception: bf16 logits take few distinct values, and every entry that carries the -1e9
    offset is the same fp32 number.)"""
    v, i = torch.sort(x, dim=-1, descending=True, stable=True)
    return v[..., :k], i[..., :k]


def torch_step(st, lp_rows, cur, min_len, max_len, length_penalty, eos_id, v431=False):
    """One iteration of the torch pipeline (emu_amd/llama.py::beam_search_generate, deterministic branch), on a dict of state, with
    its three top-k calls made stable.  v431: the scorer conventions of transformers 4.31 (hf_semantics="4.31")."""
    B, nb, V = lp_rows.shape
    dev = lp_rows.device
    gather = lambda t, idx: torch.gather(t, 1, idx.reshape(B, -1, *([1] * (t.dim() - 2))).expand(-1, -1, *t.shape[2:]))
    log_probs = torch.log_softmax(lp_rows, dim=-1)
    if cur < min_len:
        log_probs = log_probs.clone()
        log_probs[..., eos_id] = -float("inf")
    acc = (log_probs + st["running_scores"][:, :, None]).reshape(B, nb * V)
    top_lp, top_idx = topk_stable(acc, 2 * nb)
    src_beam, tok = top_idx // V, top_idx % V
    cand_seq = gather(st["running_seq"], src_beam)
    cand_seq[:, :, cur] = tok
    hits = (tok == eos_id) if v431 else (tok == eos_id) | (cur + 1 >= max_len)
    run_lp = top_lp + hits.float() * NEG
    nxt = topk_stable(run_lp, nb)[1]
    st["running_seq"] = gather(cand_seq, nxt)
    st["running_scores"] = torch.gather(run_lp, 1, nxt)
    beam_idx = torch.gather(src_beam, 1, nxt)
    fin_lp = top_lp / float((cur if (v431 and cur > 0) else cur + 1) ** length_penalty)
    fin_lp = fin_lp + (~st["open"]).float() * NEG
    top_mask = torch.cat([torch.ones(nb, dtype=torch.bool), torch.zeros(nb, dtype=torch.bool)]).to(dev)
    just = hits & top_mask[None, :]
    fin_lp = fin_lp + (~just).float() * NEG
    m_seq = torch.cat((st["sequences"], cand_seq), dim=1)
    m_sc = torch.cat((st["beam_scores"], fin_lp), dim=1)
    m_fin = torch.cat((st["finished"], just), dim=1)
    m_len = torch.cat((st["seq_len"], torch.full((B, 2 * nb), cur + 1, dtype=torch.int64, device=dev)), dim=1)
    keep = topk_stable(m_sc, nb)[1]
    st["sequences"] = gather(m_seq, keep)
    st["beam_scores"] = torch.gather(m_sc, 1, keep)
    st["finished"] = torch.gather(m_fin, 1, keep)
    st["seq_len"] = torch.gather(m_len, 1, keep)
    best_run = (top_lp[:, :1] if v431 else st["running_scores"][:, :1]) / float((cur + 1) ** length_penalty)
    worst_fin = torch.where(st["finished"], st["beam_scores"].min(dim=1, keepdim=True)[0], torch.full_like(st["beam_scores"], NEG))
    st["open"] = st["open"] & (best_run > worst_fin).any(dim=-1, keepdim=True)
    toks = st["running_seq"][:, :, cur].reshape(-1)
    if v431 and cur + 1 >= max_len:                # finalize: the running beams join at L ** length_penalty unless the prompt is done
        fin2 = st["running_scores"] / float((cur + 1) ** length_penalty) + (~st["open"]).float() * NEG
        m_sc = torch.cat((st["beam_scores"], fin2), dim=1)
        keep = topk_stable(m_sc, nb)[1]
        st["sequences"] = gather(torch.cat((st["sequences"], st["running_seq"]), dim=1), keep)
        st["beam_scores"] = torch.gather(m_sc, 1, keep)
        st["finished"] = torch.gather(torch.cat((st["finished"], torch.ones_like(st["finished"])), dim=1), 1, keep)
        st["seq_len"] = torch.gather(torch.cat((st["seq_len"], torch.full((B, nb), cur + 1, dtype=torch.int64, device=dev)), dim=1), 1, keep)
    flat = (beam_idx + torch.arange(B, device=dev)[:, None] * nb).reshape(-1)
    return toks, flat


@pytest.mark.parametrize("v431", [False, True])
@pytest.mark.parametrize("B,nb,V,max_len,min_len,lp,eos_boost", [
    (1, 5, 32274, 10, 1, -1.0, 0.0),           # the reference's defaults
    (2, 5, 32274, 12, 1, -1.0, 6.0),           # EOS often among the best: results finish, the heuristic closes
    (3, 3, 1000, 8, 4, 1.0, 8.0),              # min_length masks EOS for four steps; positive length penalty
    (2, 8, 4096, 6, 1, 0.0, 5.0),              # 8 beams, length_penalty 0 (Emu1's captions)
    (1, 2, 64, 5, 1, -1.0, 3.0),               # tiny vocabulary
])
def test_beam_step_kernel_equals_the_torch_pipeline(B, nb, V, max_len, min_len, lp, eos_boost, v431):
    """Random bf16 logits (exact ties included: bf16 takes few values), the EOS logit raised so that finished hypotheses, the merge
    with the kept results and the early-stopping heuristic all happen: after every step the kernel's state -- running and kept
    sequences, both score sets, finished flags, lengths, the heuristic flag -- and its outputs (tokens to feed, cache rows to
    continue) equal the torch pipeline's under the same tie rule, for both scorer conventions (hf431 = 0: transformers 5.x, 1: the
    4.31 the reference pins).  Scores are compared to 1e-4 (the log-sum-exp is summed in another order; a near-tie closer than that
    between two beams could legitimately flip -- none does on these seeds).  Under v431 the step index comes from a DEVICE counter
    (the hipGraph-replay form of the call), and a call beyond the length limit must leave every piece of state untouched."""
    from emu_amd import ops
    from emu_amd._lib import lib, check
    dev = torch.device("cuda", 0)
    eos = 2
    g = torch.Generator().manual_seed(B * 1000 + nb * 10 + max_len)
    ref = dict(running_seq=torch.full((B, nb, max_len), 32000, dtype=torch.int64, device=dev),
               running_scores=torch.zeros(B, nb, device=dev), beam_scores=torch.full((B, nb), NEG, device=dev),
               finished=torch.zeros(B, nb, dtype=torch.bool, device=dev), seq_len=torch.zeros(B, nb, dtype=torch.int64, device=dev),
               open=torch.ones(B, 1, dtype=torch.bool, device=dev))
    ref["sequences"] = ref["running_seq"].clone()
    ref["running_scores"][:, 1:] = NEG
    i32 = dict(dtype=torch.int32, device=dev)
    k_run = torch.full((B, nb, max_len), 32000, **i32)
    k_seq = k_run.clone()
    k_rs = ref["running_scores"].clone()
    k_bs = ref["beam_scores"].clone()
    k_fin = torch.zeros(B, nb, dtype=torch.uint8, device=dev)
    k_len = torch.zeros(B, nb, **i32)
    k_open = torch.ones(B, dtype=torch.uint8, device=dev)
    k_tok = torch.zeros(B * nb, **i32)
    k_flat = torch.zeros(B * nb, dtype=torch.int64, device=dev)
    L = lib()
    ws = torch.empty(L.emu_beam_step_workspace_bytes(B, nb, V), dtype=torch.uint8, device=dev)
    cur_dev = torch.zeros(1, **i32)

    def call(cur):
        check(L.emu_beam_step_bf16(lg.data_ptr(), ld_prompt, ld_beam, V, B, nb, max_len, -7 if v431 else cur,
                                   cur_dev.data_ptr() if v431 else None, min_len, eos, lp, int(v431),
                                   k_run.data_ptr(), k_seq.data_ptr(), k_rs.data_ptr(), k_bs.data_ptr(), k_fin.data_ptr(),
                                   k_len.data_ptr(), k_open.data_ptr(), k_tok.data_ptr(), k_flat.data_ptr(), ws.data_ptr(), ws.numel(),
                                   ops.stream(dev)), "emu_beam_step_bf16")

    for cur in range(max_len):
        if cur == 0:
            lg = (torch.randn(B, V, generator=g) * 3.0).to(BF16).to(dev)
            lg[:, eos] += eos_boost
            lp_rows = lg.float()[:, None, :].expand(B, nb, V)
            ld_prompt, ld_beam = lg.stride(0), 0
        else:
            lg = (torch.randn(B * nb, V, generator=g) * 3.0).to(BF16).to(dev)
            lg[:, eos] += eos_boost
            lp_rows = lg.float().view(B, nb, V)
            ld_prompt, ld_beam = nb * lg.stride(0), lg.stride(0)
        toks, flat = torch_step(ref, lp_rows, cur, min_len, max_len, lp, eos, v431)
        cur_dev.fill_(cur)
        call(cur)
        torch.cuda.synchronize()
        what = f"step {cur}"
        assert k_run.long().tolist() == ref["running_seq"].tolist(), what
        assert k_seq.long().tolist() == ref["sequences"].tolist(), what
        assert k_tok.long().tolist() == toks.tolist() and k_flat.tolist() == flat.tolist(), what
        assert k_fin.bool().tolist() == ref["finished"].tolist() and k_len.long().tolist() == ref["seq_len"].tolist(), what
        assert k_open.bool().tolist() == ref["open"].view(-1).tolist(), what
        for a_, b_ in ((k_rs, ref["running_scores"]), (k_bs, ref["beam_scores"])):
            live = b_ > -1.0e8                                              # (entries at -1e9 are compared as such)
            assert torch.equal(a_ > -1.0e8, live), what
            if bool(live.any()):
                assert float((a_[live] - b_[live]).abs().max()) < 1e-4, what
    assert bool(ref["finished"].any())                                       # the scenario did finish hypotheses
    if v431:                                                                 # a replay beyond the limit is a no-op
        snap = [t_.clone() for t_ in (k_run, k_seq, k_rs, k_bs, k_fin, k_len, k_open, k_tok, k_flat)]
        cur_dev.fill_(max_len)
        call(max_len)
        torch.cuda.synchronize()
        assert all(torch.equal(a_, b_) for a_, b_ in zip(snap, (k_run, k_seq, k_rs, k_bs, k_fin, k_len, k_open, k_tok, k_flat)))


def test_beam_step_rejects_shapes_outside_its_range():
    from emu_amd import ops
    from emu_amd._lib import lib
    dev = torch.device("cuda", 0)
    z = torch.zeros(1 << 18, device=dev)
    lg = torch.zeros(1, 64, dtype=BF16, device=dev)
    p = z.data_ptr()
    for nb, L_, cur, V in ((9, 8, 0, 64), (2, 300, 0, 64), (2, 8, 8, 64), (5, 8, 0, 8)):
        st = lib().emu_beam_step_bf16(lg.data_ptr(), 64, 0, V, 1, nb, L_, cur, None, 0, 2, 1.0, 0, p, p, p, p, p, p, p, p, p, p, 1 << 20,
                                      ops.stream(dev))
        assert st == -22, (nb, L_, cur, V, st)
