// One step of transformers' beam search, on the device, in one launch.
//
// The reference's default decoding mode is lm.generate(num_beams=5, length_penalty=-1, early_stopping=False)
// (Emu2/emu/emu.py:163-172,213-229).  emu_amd/llama.py::beam_search_generate restates the library's vectorised search in ~70
// small torch kernels per step; on a slow host their dispatch alone is 2-3 ms of a 13 ms step (bench leg beam_search_5: 13.2 ms
// per step on one box, 15.8 on another with the same GPU time).  This kernel is that step for the deterministic mode (no sampling,
// repetition penalty or n-gram ban -- those keep the torch pipeline), in two launches (the first one-workgroup version walked
// global memory in every arg-max round and took 485 us; a register-resident one spilled):
//   1. per (1024-entry vocabulary chunk, beam, prompt): the chunk's softmax statistics and its 2N best logits;
//   2. per prompt: the beams' log-sum-exp from the chunk statistics, the 2N best continuations over beams x vocabulary of
//      log p + running score  (EOS masked while min_length is not reached; ties go to the lower flat index), and
//      the scorer's bookkeeping, statement for statement the host's: the N best non-finished candidates keep running, the
//      finished ones among the first N candidates (EOS or the length limit) compete with the kept results at
//      score / (len ** length_penalty), and the early-stopping heuristic compares the best running beam with the worst kept one.
// Two scorer conventions (BeamStepArgs::hf431): 0 = the vectorised search of transformers 5.x (what is installed here and what the
// golden fixtures pin): a hypothesis that ends -- by EOS or at the length limit -- is scored over cur + 1 tokens, only the first N
// candidates may end, the heuristic looks at the best RUNNING beam.  1 = BeamSearchScorer / BeamHypotheses of transformers 4.31 (the
// release the reference pins, Emu2/requirements.txt:2; restated, not runnable here): an EOS hypothesis is scored over the cur tokens
// before the EOS (`hyp.shape[-1] ** length_penalty`), the heuristic looks at the best of all 2N candidates, and at the length limit
// `finalize` adds every running beam at L ** length_penalty unless the prompt is already done.
// The step index and everything derived from it (EOS suppression below min_len, the length divisors) are read from a device counter
// when one is given, so a captured hipGraph replays the step for every token.
// Out: the N tokens to feed next, the beam each of them extends (flat cache row, for the KV re-order), the updated state.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int BEAM_MAXN = 8;                           // beams per prompt
constexpr int BEAM_MAXL = 256;                         // generated tokens
constexpr int BEAM_T = 2 * BEAM_MAXN;                  // candidates a chunk hands up (>= 2N)
constexpr int CHUNK = 1024;                            // vocabulary entries per workgroup of the first kernel
constexpr int PART = 2 + 2 * BEAM_T;                   // floats per chunk partial: max, sum exp, T values, T indices
constexpr float BEAM_NEG = -1.0e9f;
constexpr int NT = 256, NW = NT / 64;

// "a ranks before b": higher value first, then lower index
__device__ __forceinline__ bool before(float av, int ai, float bv, int bi) { return av > bv || (av == bv && ai < bi); }

// block-wide arg-max of (v, i) under before(); every thread gets the winner
__device__ __forceinline__ void block_argmax(float& v, int& i, float* red_v, int* red_i) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int sh = 1; sh < 64; sh <<= 1) {
        const float ov = __shfl_xor(v, sh, 64);
        const int oi = __shfl_xor(i, sh, 64);
        if (before(ov, oi, v, i)) { v = ov; i = oi; }
    }
    __syncthreads();
    if (lane == 0) { red_v[wave] = v; red_i[wave] = i; }
    __syncthreads();
    v = red_v[0]; i = red_i[0];
#pragma unroll
    for (int w = 1; w < NW; ++w)
        if (before(red_v[w], red_i[w], v, i)) { v = red_v[w]; i = red_i[w]; }
}

// Kernel 1, one workgroup per (vocabulary chunk, beam, prompt): the chunk's softmax statistics and its 2N best logits (within a beam
// the order of the logits is the order of the log-probabilities), EOS masked while min_length is not reached.  4 values per thread,
// held in registers through the 2N arg-max rounds.
__global__ __launch_bounds__(NT) void beam_chunk_kernel(const BeamStepArgs a, float* part, int nchunk) {
    __shared__ float red_v[NW];
    __shared__ int red_i[NW];
    const int c = blockIdx.x, j = blockIdx.y, b = blockIdx.z, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cur = a.cur_dev ? *a.cur_dev : a.cur;
    if (cur >= a.L) return;                            // graph replays beyond the length limit are no-ops
    const bool suppress_eos = cur < a.min_len;
    const bf16_t* r = a.logits + (size_t)b * a.ld_prompt + (size_t)j * a.ld_beam;
    float x[4]; int vi[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        vi[e] = c * CHUNK + tid + NT * e;
        x[e] = vi[e] < a.V ? bf2f(r[vi[e]]) : -INFINITY;
    }
    float m = fmaxf(fmaxf(x[0], x[1]), fmaxf(x[2], x[3]));
    m = wave_max(m);
    __syncthreads();
    if (lane == 0) red_v[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red_v[0], red_v[1]), fmaxf(red_v[2], red_v[3]));
    float sum = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) sum += (x[e] == -INFINITY) ? 0.f : __expf(x[e] - m);
    sum = wave_sum(sum);
    __syncthreads();
    if (lane == 0) red_v[wave] = sum;
    __syncthreads();
    sum = (red_v[0] + red_v[1]) + (red_v[2] + red_v[3]);
    float* out = part + (((size_t)b * a.nb + j) * nchunk + c) * PART;
    if (tid == 0) { out[0] = m; out[1] = sum; }
    // (the statistics include the EOS logit: the library masks its LOG-PROBABILITY, after the softmax)
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (suppress_eos && vi[e] == a.eos_id) x[e] = -INFINITY;
    const int T = 2 * a.nb;
    for (int k = 0; k < T; ++k) {
        float bv = -INFINITY; int bi = 0x7fffffff;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (vi[e] < a.V && before(x[e], vi[e], bv, bi)) { bv = x[e]; bi = vi[e]; }
        block_argmax(bv, bi, red_v, red_i);
        if (tid == 0) { out[2 + k] = bv; out[2 + BEAM_T + k] = __int_as_float(bi); }
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (vi[e] == bi) vi[e] = 0x7fffffff;       // taken (its owner retires it)
    }
}

// Kernel 2, one workgroup per prompt: log-sum-exp of every beam from the chunk statistics, the 2N best of the N x chunks x 2N handed-up
// candidates of (log p + running score) -- ties to the lower flat index --, then the scorer's bookkeeping.
__global__ __launch_bounds__(NT) void beam_step_kernel(const BeamStepArgs a, const float* part, int nchunk) {
    __shared__ float red_v[NW];
    __shared__ int red_i[NW];
    __shared__ float s_lse[BEAM_MAXN], s_rs[BEAM_MAXN];
    __shared__ float top_lp[2 * BEAM_MAXN];
    __shared__ int top_idx[2 * BEAM_MAXN];
    __shared__ int cand[2 * BEAM_MAXN][BEAM_MAXL];     // candidate sequences
    __shared__ int kept[BEAM_MAXN][BEAM_MAXL];         // kept results before this step
    const int b = blockIdx.x, tid = threadIdx.x;
    const int nb = a.nb, V = a.V, L = a.L, cur = a.cur_dev ? *a.cur_dev : a.cur;
    if (cur >= L) return;
    int* run_seq = a.running_seq + (size_t)b * nb * L;
    int* res_seq = a.sequences + (size_t)b * nb * L;
    float* run_sc = a.running_scores + (size_t)b * nb;
    float* res_sc = a.beam_scores + (size_t)b * nb;
    unsigned char* fin = a.finished + (size_t)b * nb;
    int* res_len = a.seq_len + (size_t)b * nb;
    const float* pb = part + (size_t)b * nb * nchunk * PART;

    if (tid < nb) {                                    // log-sum-exp of beam tid over its chunks (fixed order)
        float m = -INFINITY;
        for (int c = 0; c < nchunk; ++c) m = fmaxf(m, pb[((size_t)tid * nchunk + c) * PART]);
        float sum = 0.f;
        for (int c = 0; c < nchunk; ++c) {
            const float* q = pb + ((size_t)tid * nchunk + c) * PART;
            sum += q[0] == -INFINITY ? 0.f : q[1] * __expf(q[0] - m);
        }
        s_lse[tid] = m + __logf(sum);
        s_rs[tid] = run_sc[tid];
    }
    __syncthreads();
    const int T = 2 * nb, ncand = nb * nchunk * T;
    float pv = INFINITY; int pi = -1;                  // last pick (everything taken ranks before or at it)
    for (int k = 0; k < T; ++k) {
        float bv = -INFINITY; int bi = 0x7fffffff;
        for (int q = tid; q < ncand; q += NT) {
            const int t = q % T, jc = q / T, j = jc / nchunk;
            const float* pq = pb + (size_t)jc * PART;
            const int v = __float_as_int(pq[2 + BEAM_T + t]);
            if (v == 0x7fffffff) continue;             // the chunk had fewer than T live entries
            const float x = (pq[2 + t] - s_lse[j]) + s_rs[j];
            const int idx = j * V + v;
            const bool taken = pi >= 0 && !before(pv, pi, x, idx);
            if (!taken && before(x, idx, bv, bi)) { bv = x; bi = idx; }
        }
        block_argmax(bv, bi, red_v, red_i);
        if (tid == 0) { top_lp[k] = bv; top_idx[k] = bi; }
        pv = bv; pi = bi;
    }
    __syncthreads();
    // ---- bookkeeping.  Stage the candidate sequences and the kept results, then one thread decides, then all write back.
    const int n2 = 2 * nb;
    for (int q = tid; q < n2 * L; q += NT) {
        const int k = q / L, p = q - k * L;
        const int src = top_idx[k] / V;
        cand[k][p] = p == cur ? top_idx[k] - src * V : run_seq[src * L + p];
    }
    for (int q = tid; q < nb * L; q += NT) kept[q / L][q % L] = res_seq[q];
    __shared__ int nxt[BEAM_MAXN], keep[BEAM_MAXN];
    __shared__ float new_rs[BEAM_MAXN], new_sc[BEAM_MAXN];
    __shared__ int new_len[BEAM_MAXN];
    __shared__ unsigned char new_fin[BEAM_MAXN];
    __syncthreads();
    if (tid == 0) {
        bool hits[2 * BEAM_MAXN];
        float run_lp[2 * BEAM_MAXN], fin_lp[2 * BEAM_MAXN];
        const bool open = a.heuristic_open[b] != 0;
        const bool v431 = a.hf431 != 0, at_limit = cur + 1 >= L;
        // (cur + 1) ** length_penalty as the host's double-precision power rounded to fp32 (the torch pipeline's divisor)
        const float len_div = (float)pow((double)(cur + 1), (double)a.length_penalty);
        const float eos_div = (v431 && cur > 0) ? (float)pow((double)cur, (double)a.length_penalty) : len_div;
        for (int k = 0; k < n2; ++k) {
            const int tok = top_idx[k] % V;
            const bool eos = tok == a.eos_id;
            hits[k] = eos || (!v431 && at_limit);      // 4.31: at the limit the non-EOS candidates still become running beams
            run_lp[k] = top_lp[k] + (hits[k] ? 1.f : 0.f) * BEAM_NEG;
            float f = top_lp[k] / (eos ? eos_div : len_div);
            f = f + (open ? 0.f : 1.f) * BEAM_NEG;
            const bool just = hits[k] && k < nb;       // only the first N candidates may finish
            f = f + (just ? 0.f : 1.f) * BEAM_NEG;
            fin_lp[k] = f;
        }
        // the N best running candidates (stable: equal scores keep candidate order)
        bool used[2 * BEAM_MAXN] = {};
        for (int j = 0; j < nb; ++j) {
            int best = -1;
            for (int k = 0; k < n2; ++k)
                if (!used[k] && (best < 0 || run_lp[k] > run_lp[best])) best = k;
            used[best] = true;
            nxt[j] = best;
            new_rs[j] = run_lp[best];
        }
        // merge the finished candidates with the kept results: N best of (kept 0..N-1, candidates 0..2N-1)
        bool usedm[3 * BEAM_MAXN] = {};
        for (int j = 0; j < nb; ++j) {
            int best = -1; float bs = 0.f;
            for (int e = 0; e < nb + n2; ++e) {
                if (usedm[e]) continue;
                const float sc = e < nb ? res_sc[e] : fin_lp[e - nb];
                if (best < 0 || sc > bs) { best = e; bs = sc; }
            }
            usedm[best] = true;
            keep[j] = best;
            new_sc[j] = bs;
            new_fin[j] = best < nb ? fin[best] : (unsigned char)(hits[best - nb] && best - nb < nb);
            new_len[j] = best < nb ? res_len[best] : cur + 1;
        }
        // early-stopping heuristic (early_stopping=False): can the best running beam (4.31: the best of all 2N candidates, an
        // EOS one included) still beat the worst kept result?
        const float best_run = (v431 ? top_lp[0] : new_rs[0]) / len_div;
        float worst = INFINITY;
        for (int j = 0; j < nb; ++j) worst = fminf(worst, new_sc[j]);
        bool still = false;
        for (int j = 0; j < nb; ++j) still |= best_run > (new_fin[j] ? worst : BEAM_NEG);
        const bool open_now = open && still;
        a.heuristic_open[b] = (unsigned char)open_now;
        // 4.31 `finalize` at the length limit: the N running beams (now L tokens long) join the kept results at L ** length_penalty
        // unless the prompt is done; entries nb + n2 + j of the merge below are running beam j (= candidate nxt[j])
        if (v431 && at_limit) {
            float k_sc[BEAM_MAXN]; int k_src[BEAM_MAXN]; int k_len[BEAM_MAXN]; unsigned char k_fin[BEAM_MAXN];
            for (int j = 0; j < nb; ++j) { k_sc[j] = new_sc[j]; k_src[j] = keep[j]; k_len[j] = new_len[j]; k_fin[j] = new_fin[j]; }
            bool used2[2 * BEAM_MAXN] = {};
            for (int j = 0; j < nb; ++j) {
                int best = -1; float bs = 0.f;
                for (int e = 0; e < 2 * nb; ++e) {
                    if (used2[e]) continue;
                    const float sc = e < nb ? k_sc[e] : new_rs[e - nb] / len_div + (open_now ? 0.f : 1.f) * BEAM_NEG;
                    if (best < 0 || sc > bs) { best = e; bs = sc; }
                }
                used2[best] = true;
                new_sc[j] = bs;
                if (best < nb) { keep[j] = k_src[best]; new_fin[j] = k_fin[best]; new_len[j] = k_len[best]; }
                else { keep[j] = nb + nxt[best - nb]; new_fin[j] = 1; new_len[j] = cur + 1; }
            }
        }
    }
    __syncthreads();
    for (int q = tid; q < nb * L; q += NT) {
        const int j = q / L, p = q - j * L;
        run_seq[q] = cand[nxt[j]][p];
        const int e = keep[j];
        res_seq[q] = e < nb ? kept[e][p] : cand[e - nb][p];
    }
    if (tid < nb) {
        run_sc[tid] = new_rs[tid];
        res_sc[tid] = new_sc[tid];
        fin[tid] = new_fin[tid];
        res_len[tid] = new_len[tid];
        a.next_tok[b * nb + tid] = cand[nxt[tid]][cur];
        a.beam_flat[b * nb + tid] = (long)b * nb + top_idx[nxt[tid]] / V;
    }
}

// ---- the rest of a beam step that used to be host work: with these two the whole step {advance, re-order, embed, 60 layers,
// logits, beam_step_kernel} reads its step index from a device counter and is replayed from one hipGraph.
// Step bookkeeping.  phase 0 (ahead of the decoder step that feeds token cur - 1): slot = slot0 + cur - 1, pos = pos0 + cur - 1;
// phase 1 (behind beam_step_kernel): cur += 1.  Steps at the length limit change nothing (replays beyond it are no-ops).
__global__ void beam_advance_kernel(int32_t* cur_dev, int32_t* pos, int32_t* slot, const int32_t* pos0, int slot0, int rows, int L,
                                    int phase) {
    const int c = *cur_dev;
    if (c >= L) return;
    if (phase == 0) {
        for (int i = threadIdx.x; i < rows; i += blockDim.x) { slot[i] = slot0 + c - 1; pos[i] = pos0[i] + c - 1; }
    } else if (threadIdx.x == 0) {
        *cur_dev = c + 1;
    }
}

// transformers' _reorder_cache for the GENERATED slots [slot0, slot0 + cur - 1) (the prompt's slots are shared by the beams of a
// group and never move): cache row r takes the slots of row beam_flat[r].  In place: one workgroup owns a (layer, head, slot) of
// all the rows of a group -- it loads the nb source vectors, then stores them (a barrier in between).  grid = (layers * heads,
// slot chunks, groups); D / 8 lanes per row.
template <int D>
__global__ __launch_bounds__(256) void beam_reorder_kernel(bf16_t* kc, bf16_t* vc, const long* beam_flat, const int32_t* cur_dev,
                                                           int rows, int Hl, int s_max, int nb, int slot0, int L, int chunk) {
    constexpr int LPR = D / 8;                         // lanes per row (16-byte pieces)
    const int c = *cur_dev;
    if (c >= L) return;
    const int ngen = c - 1;
    const int lh = blockIdx.x, layer = lh / Hl, h = lh - layer * Hl, grp = blockIdx.z;
    const int j = threadIdx.x / LPR, piece = threadIdx.x - j * LPR;
    const bool live = j < nb;
    const int dst_row = grp * nb + j;
    const int src_row = live ? (int)beam_flat[dst_row] : 0;
    const size_t plane = (size_t)rows * Hl * s_max * D;
    for (int sl = blockIdx.y * chunk; sl < ngen && sl < (blockIdx.y + 1) * chunk; ++sl) {
        const size_t so = (((size_t)layer * rows + src_row) * Hl + h) * s_max + slot0 + sl;
        const size_t dofs = (((size_t)layer * rows + dst_row) * Hl + h) * s_max + slot0 + sl;
        u32x4 kv, vv;
        if (live) { kv = ld16(kc + so * D + piece * 8); vv = ld16(vc + so * D + piece * 8); }
        __syncthreads();                               // every source of this slot is in registers
        if (live) { st16(kc + dofs * D + piece * 8, kv); st16(vc + dofs * D + piece * 8, vv); }
        __syncthreads();
    }
    (void)plane;
}

}  // namespace

int launch_beam_advance(int32_t* cur_dev, int32_t* pos, int32_t* slot, const int32_t* pos0, int slot0, int rows, int L, int phase,
                        hipStream_t s) {
    if (!cur_dev || rows < 1 || (phase == 0 && (!pos || !slot || !pos0))) return -22;
    hipLaunchKernelGGL(beam_advance_kernel, dim3(1), dim3(64), 0, s, cur_dev, pos, slot, pos0, slot0, rows, L, phase);
    EMU_CHECK_LAUNCH();
    return 0;
}

int launch_beam_reorder(bf16_t* kc, bf16_t* vc, const long* beam_flat, const int32_t* cur_dev, int layers, int rows, int Hl, int s_max,
                        int D, int nb, int slot0, int L, hipStream_t s) {
    if (!kc || !vc || !beam_flat || !cur_dev || nb < 1 || nb > BEAM_MAXN || rows % nb || (D != 64 && D != 128) ||
        slot0 < 0 || slot0 + L > s_max + 1)
        return -22;
    if (L < 2) return 0;                               // never a generated slot to move
    constexpr int CHUNK_SLOTS = 8;
    const dim3 grid(layers * Hl, (L - 1 + CHUNK_SLOTS - 1) / CHUNK_SLOTS, rows / nb);
    if (D == 128) hipLaunchKernelGGL(beam_reorder_kernel<128>, grid, dim3(nb * 16 <= 128 ? 128 : 256), 0, s, kc, vc, beam_flat, cur_dev, rows, Hl, s_max, nb, slot0, L, CHUNK_SLOTS);
    else hipLaunchKernelGGL(beam_reorder_kernel<64>, grid, dim3(64), 0, s, kc, vc, beam_flat, cur_dev, rows, Hl, s_max, nb, slot0, L, CHUNK_SLOTS);
    EMU_CHECK_LAUNCH();
    return 0;
}

size_t beam_step_ws_floats(int B, int nb, int V) { return (size_t)B * nb * ((V + CHUNK - 1) / CHUNK) * PART; }

int launch_beam_step(const BeamStepArgs& a, float* ws, size_t ws_floats, hipStream_t s) {
    if (a.B < 1 || a.nb < 1 || a.nb > BEAM_MAXN || a.L < 1 || a.L > BEAM_MAXL || (!a.cur_dev && (a.cur < 0 || a.cur >= a.L)) || a.V < 2 * a.nb ||
        a.min_len < 0 || !ws || ws_floats < beam_step_ws_floats(a.B, a.nb, a.V))
        return -22;
    const int nchunk = (a.V + CHUNK - 1) / CHUNK;
    hipLaunchKernelGGL(beam_chunk_kernel, dim3(nchunk, a.nb, a.B), dim3(NT), 0, s, a, ws, nchunk);
    hipLaunchKernelGGL(beam_step_kernel, dim3(a.B), dim3(NT), 0, s, a, ws, nchunk);
    EMU_CHECK_LAUNCH();
    return 0;
}
