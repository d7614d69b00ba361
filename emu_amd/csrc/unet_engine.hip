// UNet denoise engine: the C ABI ``emu_unet_*`` (include/emu_hip.h).
//
// One call = one full denoise step of EmuVisualGeneration's loop (Emu2/emu/diffusion.py:130-149):
//   cat([latents]*2) -> scale_model_input -> UNet2DConditionModel (SDXL topology, conf/diffusion_config/unet/
//   config.json:1-72) -> chunk (cond, uncond) -> classifier-free guidance -> EulerDiscreteScheduler.step,
// as ~900 launches on one stream with no allocation or synchronisation (hipGraph-capturable; per-step inputs --
// timestep embedding row, sigma -- are read from device tables through a device-side step counter).
// Activations are NHWC ([B*H*W, C] bf16), so every 1x1 conv / Linear is a plain GEMM, 3x3 convs are implicit GEMMs
// (gemm.hip CONV mode, nearest-x2 upsample and stride 2 folded into the gather), and Transformer2D needs no permute.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/emu_hip.h"
#include "kernels.h"

struct emu_ctx;
int emu_ctx_fail(emu_ctx* c, int code, const char* what);      // engine.hip

namespace {
inline hipStream_t S(emu_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
inline const bf16_t* B16(const void* p) { return reinterpret_cast<const bf16_t*>(p); }
inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }
const ConvGeom NOCONV{0, 0, 0, 0, 0, 0};

struct Resnet {
    std::string name;
    int cin, cout, temb_off;
    const bf16_t *n1g, *n1b, *c1w, *c1b, *n2g, *n2b, *c2w, *c2b, *scw, *scb;
};
struct LnW {                        // a LayerNorm folded into its consumer GEMM: W * gamma, fp32 row sums, W @ beta + bias
    const bf16_t* w = nullptr;
    const float *c = nullptr, *d = nullptr;
    bool ok() const { return w && c && d; }
};
struct W8 {                         // optional e4m3 copy of a packed matrix: bytes [N, K] + one fp32 scale per output row
    const uint8_t* q = nullptr;
    const float* s = nullptr;
    bool ok() const { return q && s; }
};
struct TBlock {
    const bf16_t *ln1g, *ln1b, *qkv, *o1w, *o1b, *ln2g, *ln2b, *q2, *kv2, *o2w, *o2b, *ln3g, *ln3b, *ggw, *ggb, *ffw, *ffb;
    LnW qkv_ln, q2_ln, gg_ln;       // optional packed tensors of the fused-LayerNorm path
    W8 qkv8, o1_8, q2_8, o2_8, gg8, ff8;   // emu_unet_use_fp8: the six matrices of the block as fp8 operands
    size_t ctx_off;                 // element offset of this block's {K|V rows, Vt} in the context cache
};
struct Transformer {
    std::string name;
    int c, heads, depth;
    const bf16_t *gng, *gnb, *piw, *pib, *pow_, *pob;
    std::vector<TBlock> blocks;
};
struct Level { int c, hw_shift; };
}  // namespace

struct emu_unet {
    emu_ctx* ctx;
    emu_unet_cfg cfg;
    float* splitk = nullptr;            // bound to the caller's workspace at every entry (plan_ws)
    size_t splitk_floats = 0;
    std::map<std::string, const bf16_t*> w;
    bool finalized = false;
    int fusion = 0;                     // bit 0: LayerNorm folded into the consumer GEMMs, bit 1: V^T from the qkv epilogue,
                                        // bit 2: cross-attention inside the attn2 to_q epilogue
    int fusion_avail = 0;               // what the registered tensors allow (set by emu_unet_finalize)
    bool fp8 = false;                   // emu_unet_use_fp8: the transformer blocks' GEMMs W8A8 (fusion bit 0 has no fp8 form)
    int cfg_half = -1;                  // emu_unet_set_cfg_half: -1 = the CFG pair (batch 2); 0 / 1 = only the cond / uncond row (a rank pair splits the pair)
    // resolved structure
    const bf16_t *conv_in_w, *conv_in_b, *te1w, *te1b, *te2w, *te2b, *ae1w, *ae1b, *ae2w, *ae2b, *tpw, *tpb;
    const bf16_t *cno_g, *cno_b, *cout_w, *cout_b;
    std::vector<Resnet> down_res[3], up_res[3];
    std::vector<Transformer> down_tr[3], up_tr[3];
    const bf16_t *down_ds_w[3] = {}, *down_ds_b[3] = {}, *up_us_w[3] = {}, *up_us_b[3] = {};
    Resnet mid_res[2];
    Transformer mid_tr;
    int temb_total = 0;
    size_t ctx_elems = 0;
    int n_ctx = 0;
    const bf16_t* ctx_cache = nullptr;
    const bf16_t* aug_emb = nullptr;
    std::string err;
};

namespace {

int ufail(emu_unet* u, int code, const std::string& what) {
    u->err = what;
    return emu_ctx_fail(u->ctx, code, u->err.c_str());
}
#define UTRY(expr) do { int st__ = (expr); if (st__ != 0) return ufail(u, st__, #expr); } while (0)

const bf16_t* find(emu_unet* u, const std::string& k, bool required, bool* ok) {
    auto it = u->w.find(k);
    if (it == u->w.end()) {
        if (required) { *ok = false; u->err = "emu_unet_finalize: missing weight '" + k + "'"; }
        return nullptr;
    }
    return it->second;
}

void resolve_resnet(emu_unet* u, Resnet& r, bool* ok) {
    const std::string& p = r.name;
    r.n1g = find(u, p + "norm1.g", true, ok); r.n1b = find(u, p + "norm1.b", true, ok);
    r.c1w = find(u, p + "conv1.w", true, ok); r.c1b = find(u, p + "conv1.b", true, ok);
    r.n2g = find(u, p + "norm2.g", true, ok); r.n2b = find(u, p + "norm2.b", true, ok);
    r.c2w = find(u, p + "conv2.w", true, ok); r.c2b = find(u, p + "conv2.b", true, ok);
    const bool sc = r.cin != r.cout;
    r.scw = find(u, p + "shortcut.w", sc, ok); r.scb = find(u, p + "shortcut.b", sc, ok);
}

void resolve_transformer(emu_unet* u, Transformer& t, bool* ok) {
    const std::string& p = t.name;
    t.gng = find(u, p + "norm.g", true, ok); t.gnb = find(u, p + "norm.b", true, ok);
    t.piw = find(u, p + "proj_in.w", true, ok); t.pib = find(u, p + "proj_in.b", true, ok);
    t.pow_ = find(u, p + "proj_out.w", true, ok); t.pob = find(u, p + "proj_out.b", true, ok);
    t.blocks.resize(t.depth);
    for (int k = 0; k < t.depth; ++k) {
        const std::string b = p + "transformer_blocks." + std::to_string(k) + ".";
        TBlock& tb = t.blocks[k];
        tb.ln1g = find(u, b + "norm1.g", true, ok); tb.ln1b = find(u, b + "norm1.b", true, ok);
        tb.ln2g = find(u, b + "norm2.g", true, ok); tb.ln2b = find(u, b + "norm2.b", true, ok);
        tb.ln3g = find(u, b + "norm3.g", true, ok); tb.ln3b = find(u, b + "norm3.b", true, ok);
        tb.qkv = find(u, b + "attn1.qkv.w", true, ok);
        tb.o1w = find(u, b + "attn1.out.w", true, ok); tb.o1b = find(u, b + "attn1.out.b", true, ok);
        tb.q2 = find(u, b + "attn2.q.w", true, ok); tb.kv2 = find(u, b + "attn2.kv.w", true, ok);
        tb.o2w = find(u, b + "attn2.out.w", true, ok); tb.o2b = find(u, b + "attn2.out.b", true, ok);
        tb.ggw = find(u, b + "ff.geglu.w", true, ok); tb.ggb = find(u, b + "ff.geglu.b", true, ok);
        tb.ffw = find(u, b + "ff.out.w", true, ok); tb.ffb = find(u, b + "ff.out.b", true, ok);
        auto lnw = [&](const std::string& base) {
            LnW l;
            l.w = find(u, base + ".wln", false, ok);
            l.c = reinterpret_cast<const float*>(find(u, base + ".c", false, ok));
            l.d = reinterpret_cast<const float*>(find(u, base + ".d", false, ok));
            return l;
        };
        tb.qkv_ln = lnw(b + "attn1.qkv"); tb.q2_ln = lnw(b + "attn2.q"); tb.gg_ln = lnw(b + "ff.geglu");
        if (!(tb.qkv_ln.ok() && tb.q2_ln.ok() && tb.gg_ln.ok())) u->fusion_avail &= ~1;
    }
}

// ---------------------------------------------------------------- workspace plan
struct Ws {
    bf16_t *colin, *hA, *hB, *cat, *gn, *t1, *sc, *tokA, *tokB, *ln, *qkv, *vt, *att, *q2, *ff;
    bf16_t* skip[12];
    bf16_t *temb_in, *e1, *emb, *semb, *temb_all, *add1;
    uint8_t* x8;            // emu_unet_use_fp8: the current GEMM's activation rows as e4m3 bytes ...
    float* xs;              // ... and their per-row scales
    float* lnstats;         // per-row partial (sum, sum of squares) of the transformer stream, one pair per 128-column slot
    float* gnws;
    float* splitk;          // fp32 K-slices of the split-K GEMMs / convs of the lowest-resolution level
    size_t splitk_floats;
    size_t total;
};

Ws plan_ws(const emu_unet* u, int H, int W, void* base) {
    const emu_unet_cfg& c = u->cfg;
    const int Bn = 2;
    char* p = reinterpret_cast<char*>(base);
    size_t off = 0;
    auto take = [&](size_t elems, size_t esz = 2) { char* r = p ? p + off : nullptr; off += align_up(elems * esz); return r; };
    const size_t hw[3] = {(size_t)H * W, (size_t)((H + 1) / 2) * ((W + 1) / 2), (size_t)((H + 3) / 4) * ((W + 3) / 4)};
    size_t max_h = 0, max_cat = 0, max_tok = 0, max_qkv = 0, max_ff = 0, max_vt = 0, max_st = 0;
    for (int i = 0; i < 3; ++i) {
        max_h = std::max(max_h, Bn * hw[i] * c.ch[i]);
        // widest concat at level i: up block (2-i) first resnet input = ch[i] (or ch[i+1]) + skip
        const int wide = (i == 2) ? 2 * c.ch[2] : c.ch[i + 1] + c.ch[i];
        max_cat = std::max(max_cat, Bn * hw[i] * (size_t)wide);
        if (c.attn[i]) {
            max_tok = std::max(max_tok, Bn * hw[i] * c.ch[i]);
            max_qkv = std::max(max_qkv, Bn * hw[i] * 3 * c.ch[i]);
            max_ff = std::max(max_ff, Bn * hw[i] * 4 * c.ch[i]);
            max_vt = std::max(max_vt, (size_t)Bn * c.ch[i] * ((hw[i] + 63) / 64 * 64));
            max_st = std::max(max_st, (size_t)Bn * hw[i] * ((c.ch[i] + 127) / 128) * 2);
        }
    }
    // a downsampled/upsampled tensor is written at the next level's size with the previous level's channels
    max_h = std::max(max_h, Bn * hw[0] * (size_t)c.ch[1]);        // upsampler output of up block 1: [B, hw0, ch1]
    max_h = std::max(max_h, Bn * hw[1] * (size_t)c.ch[2]);        // upsampler output of up block 0: [B, hw1, ch2]
    Ws w;
    w.colin = (bf16_t*)take(Bn * hw[0] * c.kpad_in);
    w.hA = (bf16_t*)take(max_h); w.hB = (bf16_t*)take(max_h);
    w.cat = (bf16_t*)take(max_cat); w.gn = (bf16_t*)take(max_cat);
    w.t1 = (bf16_t*)take(max_h); w.sc = (bf16_t*)take(max_h);
    w.tokA = (bf16_t*)take(max_tok); w.tokB = (bf16_t*)take(max_tok); w.ln = (bf16_t*)take(max_tok);
    w.qkv = (bf16_t*)take(max_qkv); w.vt = (bf16_t*)take(max_vt); w.att = (bf16_t*)take(max_tok);
    w.q2 = (bf16_t*)take(max_tok); w.ff = (bf16_t*)take(max_ff);
    // skip connections: conv_in, then per down block its resnet/attn outputs and its downsampler output
    int k = 0;
    w.skip[k++] = (bf16_t*)take(Bn * hw[0] * c.ch[0]);
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < c.layers_per_block; ++j) w.skip[k++] = (bf16_t*)take(Bn * hw[i] * c.ch[i]);
        if (i < 2) w.skip[k++] = (bf16_t*)take(Bn * hw[i + 1] * c.ch[i]);
    }
    w.temb_in = (bf16_t*)take(Bn * c.ch[0]); w.e1 = (bf16_t*)take(Bn * c.temb_dim); w.emb = (bf16_t*)take(Bn * c.temb_dim);
    w.semb = (bf16_t*)take(Bn * c.temb_dim); w.temb_all = (bf16_t*)take((size_t)Bn * u->temb_total);
    w.add1 = (bf16_t*)take(Bn * c.temb_dim);
    w.x8 = (uint8_t*)take(u->fp8 ? max_ff : 0, 1);
    w.xs = (float*)take(u->fp8 ? max_tok : 0, 4);       // (>= rows of any transformer level)
    w.lnstats = (float*)take(max_st, 4);
    w.gnws = (float*)take(gn_ws_floats(Bn, 2 * c.ch[2] > c.ch[1] + c.ch[0] ? 2 * c.ch[2] : c.ch[1] + c.ch[0], (int)hw[0]), 4);
    w.splitk_floats = EMU_SPLITK_SCRATCH_FLOATS;
    w.splitk = (float*)take(w.splitk_floats, 4);
    w.total = off;
    return w;
}

// optional fused epilogues of one GEMM of a transformer block (GemmArgs: row_stats_out / ln_* / vt_*)
struct Fx {
    float* stats_out = nullptr;         // producer: per-row partial sums of the output for the next LayerNorm
    const LnW* ln = nullptr;            // consumer: LayerNorm of A folded in (W = ln->w), statistics from `stats_in`
    const float* stats_in = nullptr;
    bf16_t* vt = nullptr;               // qkv projection: V heads (columns >= vt_col0) stored key-contiguous
    int vt_col0 = 0, vt_s = 0, vt_spad = 0;
    const bf16_t* cross_k = nullptr;    // attn2 to_q projection: cross-attention over the cached prompt K / V^T in the epilogue
    const bf16_t* cross_vt = nullptr;
    int cross_ldk = 0, cross_n = 0, cross_npad = 0, cross_rows = 0;
    float cross_scale = 0.f;
    const void* pf = nullptr;           // GemmArgs::pf_*: the weight matrix of the GEMM that follows this one on the stream
    size_t pf_bytes = 0;
};

int gemm(emu_unet* u, const bf16_t* A, const bf16_t* Wt, const bf16_t* bias, const bf16_t* res, bf16_t* C, int M, int N, int K,
         int lda, int ldres, int ldc, int epi, hipStream_t s, const Fx* fx = nullptr) {
    if (M <= 8) {
        if (fx) return -22;
        GemvArgs g{A, Wt, nullptr, bias, res, C, M, N, K, lda, K, ldres, ldc, 0.f, epi, 0, nullptr};
        return launch_gemv(g, s);
    }
    GemmArgs g{A, Wt, bias, res, C, M, N, K, lda, K, ldres, ldc, epi, NOCONV, nullptr, 0, 0};
    if (u) { g.partial = u->splitk; g.partial_floats = u->splitk_floats; }    // primitives (u == null): process scratch
    if (fx) {
        g.row_stats_out = fx->stats_out;
        if (fx->ln) {
            g.W = fx->ln->w; g.bias = nullptr;
            g.ln_c = fx->ln->c; g.ln_d = fx->ln->d; g.ln_stats = fx->stats_in; g.ln_slots = K / 128; g.ln_eps = 1e-5f;
        }
        g.vt_out = fx->vt; g.vt_col0 = fx->vt_col0; g.vt_s = fx->vt_s; g.vt_spad = fx->vt_spad;
        g.cross_k = fx->cross_k; g.cross_vt = fx->cross_vt; g.cross_ldk = fx->cross_ldk; g.cross_n = fx->cross_n;
        g.cross_npad = fx->cross_npad; g.cross_rows = fx->cross_rows; g.cross_scale = fx->cross_scale;
        g.pf_ptr = fx->pf; g.pf_bytes = fx->pf_bytes;
    }
    return launch_gemm(g, s);
}

// epi(fp8 rows in w.x8 / w.xs  x  fp8 weights): launch_gemm_fp8 (emu_linear_fp8_bf16's kernels)
// (of the fused epilogues the V^T stores and the cross-attention are available with fp8 operands: Fx::vt / Fx::cross_*)
int gemm8(emu_unet* u, const Ws& w, const W8& W, const bf16_t* bias, const bf16_t* res, bf16_t* C, int M, int N, int K, int ldres,
          int ldc, int epi, hipStream_t s, const Fx* fx = nullptr) {
    GemmArgs g{reinterpret_cast<const bf16_t*>(w.x8), reinterpret_cast<const bf16_t*>(W.q), bias, res, C, M, N, K, K, K, ldres, ldc, epi,
               NOCONV, nullptr, 0, 0};
    g.a_scale = w.xs; g.w_scale = W.s;
    g.partial = u->splitk; g.partial_floats = u->splitk_floats;
    if (fx) {
        g.vt_out = fx->vt; g.vt_col0 = fx->vt_col0; g.vt_s = fx->vt_s; g.vt_spad = fx->vt_spad;
        g.cross_k = fx->cross_k; g.cross_vt = fx->cross_vt; g.cross_ldk = fx->cross_ldk; g.cross_n = fx->cross_n;
        g.cross_npad = fx->cross_npad; g.cross_rows = fx->cross_rows; g.cross_scale = fx->cross_scale;
    }
    return launch_gemm_fp8(g, s);
}

int conv3(emu_unet* u, const bf16_t* x, const bf16_t* Wt, const bf16_t* bias, const bf16_t* bias2, int ldb2, const bf16_t* res,
          bf16_t* y, int Bn, int Hin, int Win, int Cin, int Cout, int mode, hipStream_t s) {
    int Ho = Hin, Wo = Win;
    if (mode == CONV_3X3_S2) { Ho = (Hin + 1) / 2; Wo = (Win + 1) / 2; }
    if (mode == CONV_3X3_UP2) { Ho = 2 * Hin; Wo = 2 * Win; }
    GemmArgs g{x, Wt, bias, res, y, Bn * Ho * Wo, Cout, 9 * Cin, 0, 9 * Cin, Cout, Cout, res ? EPI_RESID : EPI_NONE,
               ConvGeom{mode, Hin, Win, Ho, Wo, Cin}, bias2, Ho * Wo, ldb2};
    if (u) { g.partial = u->splitk; g.partial_floats = u->splitk_floats; }    // primitives (u == null): process scratch
    return launch_gemm(g, s);
}

int run_resnet(emu_unet* u, const Resnet& r, const bf16_t* x, bf16_t* out, const Ws& w, int Bn, int H, int W, hipStream_t s) {
    const emu_unet_cfg& c = u->cfg;
    const int HW = H * W, M = Bn * HW;
    UTRY(launch_groupnorm(x, r.n1g, r.n1b, w.gn, w.gnws, Bn, HW, r.cin, c.groups, c.gn_eps, 1, s));
    UTRY(conv3(u, w.gn, r.c1w, r.c1b, w.temb_all + r.temb_off, u->temb_total, nullptr, w.t1, Bn, H, W, r.cin, r.cout, CONV_3X3, s));
    UTRY(launch_groupnorm(w.t1, r.n2g, r.n2b, w.gn, w.gnws, Bn, HW, r.cout, c.groups, c.gn_eps, 1, s));
    const bf16_t* res = x;
    if (r.cin != r.cout) {
        UTRY(gemm(u, x, r.scw, r.scb, nullptr, w.sc, M, r.cout, r.cin, r.cin, 0, r.cout, EPI_NONE, s));
        res = w.sc;
    }
    UTRY(conv3(u, w.gn, r.c2w, r.c2b, nullptr, 0, res, out, Bn, H, W, r.cout, r.cout, CONV_3X3, s));
    return 0;
}

int run_transformer(emu_unet* u, const Transformer& t, const bf16_t* x, bf16_t* out, const Ws& w, int Bn, int H, int W, hipStream_t s) {
    const int HW = H * W, M = Bn * HW, C = t.c, D = 64;
    const int hwpad = (HW + 63) / 64 * 64, npad = (u->n_ctx + 63) / 64 * 64, n = u->n_ctx;
    const float scale = 0.125f;
    const int hb = u->cfg_half < 0 ? 0 : u->cfg_half;     // first row of the CFG pair this call computes
    // Fused path (emu_unet_set_fusion): the three LayerNorms of a block live inside their consumer GEMMs -- the producer of the
    // stream (proj_in, attn1 / attn2 out-projection, ff-out) emits per-row partial sums from its epilogue, the consumer (qkv,
    // attn2 q, GEGLU) multiplies the un-normalised rows by W * gamma and corrects with mean / rstd in its epilogue -- and the qkv
    // projection writes V^T itself.  M <= 8 (toy latents) stays on the unfused GEMV path.
    // W8A8 mode (emu_unet_use_fp8; not a reference feature): the six GEMMs of a block take fp8 operands.  The three LayerNorms
    // run as launches again -- they hold whole rows, so their output leaves as the consumer's fp8 operand with its per-row scale
    // for free (launch_layernorm_q8) -- the attention outputs and the GEGLU product are quantised by a launch of their own
    // (their rows are spread over the workgroups of the producing kernel); the V^T stores of the qkv projection and the
    // cross-attention inside the to_q epilogue (fusion bits 1, 2) work on finished sums and stay available.  proj_in / proj_out,
    // the convs and everything outside the transformer blocks stay bf16.
    const bool f8 = u->fp8 && M > 8 && (C & 127) == 0 && C <= 2048;
    const bool fvt8 = f8 && (u->fusion & 2) && HW == hwpad;
    const bool fca8 = f8 && (u->fusion & 4) && HW == hwpad && n <= 64;
    const bool fln = !f8 && (u->fusion & 1) && M > 8 && (C & 127) == 0;     // statistics slots are 128 columns wide
    const bool fvt = !f8 && (u->fusion & 2) && M > 8 && HW == hwpad;
    const bool fca = !f8 && (u->fusion & 4) && M > 8 && HW == hwpad && n <= 64;     // rows of one tile within one batch element
    float* st = w.lnstats;
    UTRY(launch_groupnorm(x, t.gng, t.gnb, w.gn, w.gnws, Bn, HW, C, u->cfg.groups, 1e-6f, 0, s));
    // successor prefetch (Fx::pf, GemmArgs::pf_*): every GEMM of the chain names the weight matrix of the GEMM behind it (the
    // LayerNorm-folded copy where that is the operand); the attention launches in between only widen the lead.  Same-run A/B on the
    // true config: 25.07 -> 24.19 ms per denoise step (profiles/r05_prefetch_ab_unet_linear_transformer_chain.log); naming the
    // resnets' conv weights as well added nothing (..._plus_conv_modules.log) and is not done
    const size_t CC = (size_t)C * C * 2;
    const bool pfon = !f8 && M > 8 && !t.blocks.empty();
    auto qkv_w = [&](const TBlock& b_) { return fln ? (const void*)b_.qkv_ln.w : (const void*)b_.qkv; };
    { Fx fx; fx.stats_out = fln ? st : nullptr;
      if (pfon) { fx.pf = qkv_w(t.blocks[0]); fx.pf_bytes = 3 * CC; }
      UTRY(gemm(u, w.gn, t.piw, t.pib, nullptr, w.tokA, M, C, C, C, 0, C, EPI_NONE, s, (fln || pfon) ? &fx : nullptr)); }
    bf16_t *a = w.tokA, *b = w.tokB;
    for (size_t bi = 0; bi < t.blocks.size(); ++bi) {
        const TBlock& tb = t.blocks[bi];
        if (f8) {
            UTRY(launch_layernorm_q8(a, tb.ln1g, tb.ln1b, nullptr, nullptr, w.x8, w.xs, M, C, 1e-5f, s));
            { Fx fx;
              if (fvt8) { fx.vt = w.vt; fx.vt_col0 = 2 * C; fx.vt_s = HW; fx.vt_spad = hwpad; }
              UTRY(gemm8(u, w, tb.qkv8, nullptr, nullptr, w.qkv, M, 3 * C, C, 0, 3 * C, EPI_NONE, s, fvt8 ? &fx : nullptr)); }
            if (!fvt8) {
                TransposeVArgs tv{w.qkv + 2 * C, (long)HW * 3 * C, (long)D, (long)3 * C, w.vt, Bn, t.heads, HW, D, hwpad};
                UTRY(launch_transpose_v(tv, s));
            }
            { FlashArgs f{w.qkv, (long)HW * 3 * C, (long)D, (long)3 * C, w.qkv + C, (long)HW * 3 * C, (long)D, (long)3 * C, w.vt,
                          w.att, (long)HW * C, (long)D, (long)C, nullptr, Bn, t.heads, HW, HW, hwpad, D, 0, scale};
              UTRY(launch_flash_attn(f, s)); }
            UTRY(launch_quant_fp8_rows(w.att, C, w.x8, C, w.xs, M, C, s));
            UTRY(gemm8(u, w, tb.o1_8, tb.o1b, a, b, M, C, C, C, C, EPI_RESID, s));
            UTRY(launch_layernorm_q8(b, tb.ln2g, tb.ln2b, nullptr, nullptr, w.x8, w.xs, M, C, 1e-5f, s));
            { const bf16_t* kv = u->ctx_cache + tb.ctx_off + (size_t)hb * n * 2 * C;     // the cache holds both rows of the CFG pair
              const bf16_t* vt = u->ctx_cache + tb.ctx_off + (size_t)2 * n * 2 * C + (size_t)hb * C * npad;
              if (fca8) {                                // to_q + the 64-key attention in one launch: writes w.att directly
                  Fx fx;
                  fx.cross_k = kv; fx.cross_vt = vt; fx.cross_ldk = 2 * C; fx.cross_n = n; fx.cross_npad = npad; fx.cross_rows = HW;
                  fx.cross_scale = scale;
                  UTRY(gemm8(u, w, tb.q2_8, nullptr, nullptr, w.att, M, C, C, 0, C, EPI_NONE, s, &fx));
              } else {
                  UTRY(gemm8(u, w, tb.q2_8, nullptr, nullptr, w.q2, M, C, C, 0, C, EPI_NONE, s));
                  FlashArgs f{w.q2, (long)HW * C, (long)D, (long)C, kv, (long)n * 2 * C, (long)D, (long)2 * C, vt,
                              w.att, (long)HW * C, (long)D, (long)C, nullptr, Bn, t.heads, HW, n, npad, D, 0, scale};
                  UTRY(launch_flash_attn(f, s));
              } }
            UTRY(launch_quant_fp8_rows(w.att, C, w.x8, C, w.xs, M, C, s));
            UTRY(gemm8(u, w, tb.o2_8, tb.o2b, b, a, M, C, C, C, C, EPI_RESID, s));
            UTRY(launch_layernorm_q8(a, tb.ln3g, tb.ln3b, nullptr, nullptr, w.x8, w.xs, M, C, 1e-5f, s));
            UTRY(gemm8(u, w, tb.gg8, tb.ggb, nullptr, w.ff, M, 8 * C, C, 0, 4 * C, EPI_GEGLU, s));
            UTRY(launch_quant_fp8_rows(w.ff, 4 * C, w.x8, 4 * C, w.xs, M, 4 * C, s));
            UTRY(gemm8(u, w, tb.ff8, tb.ffb, a, b, M, C, 4 * C, C, C, EPI_RESID, s));
            std::swap(a, b);
            continue;
        }
        // self attention
        const bf16_t* lnx = a;
        if (!fln) { UTRY(launch_layernorm(a, tb.ln1g, tb.ln1b, nullptr, w.ln, M, C, 1e-5f, s)); lnx = w.ln; }
        { Fx fx;
          if (fln) { fx.ln = &tb.qkv_ln; fx.stats_in = st; }
          if (fvt) { fx.vt = w.vt; fx.vt_col0 = 2 * C; fx.vt_s = HW; fx.vt_spad = hwpad; }
          if (pfon) { fx.pf = tb.o1w; fx.pf_bytes = CC; }
          UTRY(gemm(u, lnx, tb.qkv, nullptr, nullptr, w.qkv, M, 3 * C, C, C, 0, 3 * C, EPI_NONE, s, (fln || fvt || pfon) ? &fx : nullptr)); }
        if (!fvt) {
            TransposeVArgs tv{w.qkv + 2 * C, (long)HW * 3 * C, (long)D, (long)3 * C, w.vt, Bn, t.heads, HW, D, hwpad};
            UTRY(launch_transpose_v(tv, s));
        }
        { FlashArgs f{w.qkv, (long)HW * 3 * C, (long)D, (long)3 * C, w.qkv + C, (long)HW * 3 * C, (long)D, (long)3 * C, w.vt,
                      w.att, (long)HW * C, (long)D, (long)C, nullptr, Bn, t.heads, HW, HW, hwpad, D, 0, scale};
          UTRY(launch_flash_attn(f, s)); }
        { Fx fx; fx.stats_out = fln ? st : nullptr;
          if (pfon) { fx.pf = fln ? (const void*)tb.q2_ln.w : (const void*)tb.q2; fx.pf_bytes = CC; }
          UTRY(gemm(u, w.att, tb.o1w, tb.o1b, a, b, M, C, C, C, C, C, EPI_RESID, s, (fln || pfon) ? &fx : nullptr)); }
        // cross attention on the cached context K / Vt
        lnx = b;
        if (!fln) { UTRY(launch_layernorm(b, tb.ln2g, tb.ln2b, nullptr, w.ln, M, C, 1e-5f, s)); lnx = w.ln; }
        { const bf16_t* kv = u->ctx_cache + tb.ctx_off + (size_t)hb * n * 2 * C;
          const bf16_t* vt = u->ctx_cache + tb.ctx_off + (size_t)2 * n * 2 * C + (size_t)hb * C * npad;
          Fx fx;
          if (fln) { fx.ln = &tb.q2_ln; fx.stats_in = st; }
          if (pfon) { fx.pf = tb.o2w; fx.pf_bytes = CC; }
          if (fca) {                                     // to_q + the whole 64-key attention in one launch: writes w.att directly
              fx.cross_k = kv; fx.cross_vt = vt; fx.cross_ldk = 2 * C; fx.cross_n = n; fx.cross_npad = npad; fx.cross_rows = HW;
              fx.cross_scale = scale;
              UTRY(gemm(u, lnx, tb.q2, nullptr, nullptr, w.att, M, C, C, C, 0, C, EPI_NONE, s, &fx));
          } else {
              UTRY(gemm(u, lnx, tb.q2, nullptr, nullptr, w.q2, M, C, C, C, 0, C, EPI_NONE, s, (fln || pfon) ? &fx : nullptr));
              FlashArgs f{w.q2, (long)HW * C, (long)D, (long)C, kv, (long)n * 2 * C, (long)D, (long)2 * C, vt,
                          w.att, (long)HW * C, (long)D, (long)C, nullptr, Bn, t.heads, HW, n, npad, D, 0, scale};
              UTRY(launch_flash_attn(f, s));
          } }
        { Fx fx; fx.stats_out = fln ? st : nullptr;
          if (pfon) { fx.pf = fln ? (const void*)tb.gg_ln.w : (const void*)tb.ggw; fx.pf_bytes = 8 * CC; }
          UTRY(gemm(u, w.att, tb.o2w, tb.o2b, b, a, M, C, C, C, C, C, EPI_RESID, s, (fln || pfon) ? &fx : nullptr)); }
        // GEGLU feed-forward
        lnx = a;
        if (!fln) { UTRY(launch_layernorm(a, tb.ln3g, tb.ln3b, nullptr, w.ln, M, C, 1e-5f, s)); lnx = w.ln; }
        { Fx fx;
          if (fln) { fx.ln = &tb.gg_ln; fx.stats_in = st; }
          if (pfon) { fx.pf = tb.ffw; fx.pf_bytes = 4 * CC; }
          UTRY(gemm(u, lnx, tb.ggw, tb.ggb, nullptr, w.ff, M, 8 * C, C, C, 0, 4 * C, EPI_GEGLU, s, (fln || pfon) ? &fx : nullptr)); }
        { Fx fx; fx.stats_out = st;                          // feeds the next block's first LayerNorm (none after the last)
          const bool last = bi + 1 == t.blocks.size();
          if (!(fln && !last)) fx.stats_out = nullptr;
          if (pfon) { fx.pf = last ? (const void*)t.pow_ : qkv_w(t.blocks[bi + 1]); fx.pf_bytes = last ? CC : 3 * CC; }
          UTRY(gemm(u, w.ff, tb.ffw, tb.ffb, a, b, M, C, 4 * C, 4 * C, C, C, EPI_RESID, s, ((fln && !last) || pfon) ? &fx : nullptr)); }
        std::swap(a, b);
    }
    UTRY(gemm(u, a, t.pow_, t.pob, x, out, M, C, C, C, C, C, EPI_RESID, s));
    return 0;
}

}  // namespace

extern "C" {

size_t emu_groupnorm_ws_bytes(int Bn, int HW, int C) { return gn_ws_floats(Bn, C, HW) * sizeof(float); }

int emu_groupnorm_nhwc_bf16(const void* x, const void* gamma, const void* beta, void* y, void* ws, int Bn, int HW, int C,
                            int groups, float eps, int silu, emu_stream_t s) {
    return launch_groupnorm(B16(x), B16(gamma), B16(beta), reinterpret_cast<bf16_t*>(y), reinterpret_cast<float*>(ws), Bn, HW, C,
                            groups, eps, silu, S(s));
}

int emu_conv3x3_nhwc_bf16(const void* x, const void* w, const void* bias, const void* bias2, int ld_bias2, const void* res,
                          void* y, int Bn, int Hin, int Win, int Cin, int Cout, int mode, emu_stream_t s) {
    return conv3(nullptr, B16(x), B16(w), B16(bias), B16(bias2), ld_bias2, B16(res), reinterpret_cast<bf16_t*>(y), Bn, Hin, Win,
                 Cin, Cout, mode, S(s));
}

int emu_unet_create(emu_ctx* ctx, const emu_unet_cfg* cfg, emu_unet** out) {
    if (!ctx || !cfg || !out) return -22;
    for (int i = 0; i < 3; ++i)
        if ((cfg->ch[i] & 63) || (cfg->attn[i] && cfg->ch[i] != cfg->heads[i] * 64)) return emu_ctx_fail(ctx, -22, "emu_unet_create: channels must be multiples of 64 and heads*64");
    if (cfg->kpad_in < 9 * cfg->in_ch || (cfg->kpad_in & 7) || cfg->out_ch != 4 || cfg->in_ch != 4) return emu_ctx_fail(ctx, -22, "emu_unet_create: bad conv_in/out config");
    emu_unet* u = new emu_unet();
    u->ctx = ctx; u->cfg = *cfg;
    *out = u;
    return 0;
}
void emu_unet_destroy(emu_unet* u) { delete u; }

int emu_unet_set_weight(emu_unet* u, const char* name, const void* ptr) {
    if (!u || !name || !ptr) return -22;
    u->w[name] = B16(ptr);
    const std::string nm(name);
    const bool f8name = (nm.size() > 4 && nm.compare(nm.size() - 4, 4, ".fp8") == 0) || (nm.size() > 5 && nm.compare(nm.size() - 5, 5, ".fp8s") == 0);
    if (!f8name) u->finalized = false;                  // (fp8 copies are resolved by emu_unet_use_fp8, not by finalize)
    return 0;
}

// W8A8 mode of the transformer blocks: every block's six packed matrices must have been registered a second time as
// "<name>.fp8" (e4m3 bytes of emu_quantize_fp8_rows) and "<name>.fp8s" (fp32 row scales) with emu_unet_set_weight.
// Changes emu_unet_workspace_bytes; invalidates captured graphs of the step (the caller's business).
int emu_unet_use_fp8(emu_unet* u, int enable) {
    if (!u || !u->finalized) return -22;
    if (!enable) { u->fp8 = false; return 0; }
    bool ok = true;
    auto w8 = [&](const std::string& base) {
        W8 r;
        r.q = reinterpret_cast<const uint8_t*>(find(u, base + ".fp8", true, &ok));
        r.s = reinterpret_cast<const float*>(find(u, base + ".fp8s", true, &ok));
        return r;
    };
    auto resolve = [&](Transformer& t) {
        for (int k = 0; k < t.depth; ++k) {
            const std::string b = t.name + "transformer_blocks." + std::to_string(k) + ".";
            TBlock& tb = t.blocks[k];
            tb.qkv8 = w8(b + "attn1.qkv.w"); tb.o1_8 = w8(b + "attn1.out.w"); tb.q2_8 = w8(b + "attn2.q.w");
            tb.o2_8 = w8(b + "attn2.out.w"); tb.gg8 = w8(b + "ff.geglu.w"); tb.ff8 = w8(b + "ff.out.w");
        }
    };
    for (int i = 0; i < 3; ++i) {
        for (auto& t : u->down_tr[i]) resolve(t);
        for (auto& t : u->up_tr[i]) resolve(t);
    }
    resolve(u->mid_tr);
    if (!ok) return emu_ctx_fail(u->ctx, -2, u->err.c_str());
    u->fp8 = true;
    return 0;
}

// Traversal order shared with emu_amd/unet.py (temb_proj_all concatenation): down i,j ; mid 0,1 ; up i,j.
int emu_unet_finalize(emu_unet* u) {
    if (!u) return -22;
    const emu_unet_cfg& c = u->cfg;
    bool ok = true;
    u->fusion_avail = 7;                               // resolve_transformer clears bit 0 when a packed tensor is missing
    int toff = 0;
    auto mk_res = [&](const std::string& name, int cin, int cout) {
        Resnet r{}; r.name = name; r.cin = cin; r.cout = cout; r.temb_off = toff; toff += cout;
        resolve_resnet(u, r, &ok);
        return r;
    };
    size_t coff = 0;
    auto mk_tr = [&](const std::string& name, int ch, int depth, int heads) {
        Transformer t{}; t.name = name; t.c = ch; t.depth = depth; t.heads = heads;
        resolve_transformer(u, t, &ok);
        return t;
    };
    u->conv_in_w = find(u, "conv_in.w", true, &ok); u->conv_in_b = find(u, "conv_in.b", true, &ok);
    u->te1w = find(u, "time_embedding.linear_1.w", true, &ok); u->te1b = find(u, "time_embedding.linear_1.b", true, &ok);
    u->te2w = find(u, "time_embedding.linear_2.w", true, &ok); u->te2b = find(u, "time_embedding.linear_2.b", true, &ok);
    u->ae1w = find(u, "add_embedding.linear_1.w", true, &ok); u->ae1b = find(u, "add_embedding.linear_1.b", true, &ok);
    u->ae2w = find(u, "add_embedding.linear_2.w", true, &ok); u->ae2b = find(u, "add_embedding.linear_2.b", true, &ok);
    u->tpw = find(u, "temb_proj_all.w", true, &ok); u->tpb = find(u, "temb_proj_all.b", true, &ok);
    u->cno_g = find(u, "conv_norm_out.g", true, &ok); u->cno_b = find(u, "conv_norm_out.b", true, &ok);
    u->cout_w = find(u, "conv_out.w", true, &ok); u->cout_b = find(u, "conv_out.b", true, &ok);
    int cin = c.ch[0];
    for (int i = 0; i < 3; ++i) {
        u->down_res[i].clear(); u->down_tr[i].clear();
        for (int j = 0; j < c.layers_per_block; ++j) {
            const std::string p = "down_blocks." + std::to_string(i);
            u->down_res[i].push_back(mk_res(p + ".resnets." + std::to_string(j) + ".", j == 0 ? cin : c.ch[i], c.ch[i]));
            if (c.attn[i]) u->down_tr[i].push_back(mk_tr(p + ".attentions." + std::to_string(j) + ".", c.ch[i], c.depth[i], c.heads[i]));
        }
        if (i < 2) {
            const std::string p = "down_blocks." + std::to_string(i) + ".downsamplers.0.conv.";
            u->down_ds_w[i] = find(u, p + "w", true, &ok); u->down_ds_b[i] = find(u, p + "b", true, &ok);
        }
        cin = c.ch[i];
    }
    u->mid_res[0] = mk_res("mid_block.resnets.0.", c.ch[2], c.ch[2]);
    u->mid_tr = mk_tr("mid_block.attentions.0.", c.ch[2], c.depth[2], c.heads[2]);
    u->mid_res[1] = mk_res("mid_block.resnets.1.", c.ch[2], c.ch[2]);
    int out = c.ch[2];
    for (int i = 0; i < 3; ++i) {                      // up block i works at level 2-i
        const int lvl = 2 - i;
        const int prev = out; out = c.ch[lvl];
        const int inp = c.ch[lvl > 0 ? lvl - 1 : 0];
        u->up_res[i].clear(); u->up_tr[i].clear();
        for (int j = 0; j <= c.layers_per_block; ++j) {
            const int skip = (j == c.layers_per_block) ? inp : out;
            const int first = (j == 0) ? prev : out;
            const std::string p = "up_blocks." + std::to_string(i);
            u->up_res[i].push_back(mk_res(p + ".resnets." + std::to_string(j) + ".", first + skip, out));
            if (c.attn[lvl]) u->up_tr[i].push_back(mk_tr(p + ".attentions." + std::to_string(j) + ".", out, c.depth[lvl], c.heads[lvl]));
        }
        if (i < 2) {
            const std::string p = "up_blocks." + std::to_string(i) + ".upsamplers.0.conv.";
            u->up_us_w[i] = find(u, p + "w", true, &ok); u->up_us_b[i] = find(u, p + "b", true, &ok);
        }
    }
    (void)coff;
    u->temb_total = toff;
    if (!ok) return emu_ctx_fail(u->ctx, -2, u->err.c_str());
    u->fusion = u->fusion_avail;
    u->fp8 = false;                                    // the blocks were re-resolved: emu_unet_use_fp8 again
    u->finalized = true;
    return 0;
}

int emu_unet_set_fusion(emu_unet* u, int mask) {
    if (!u || !u->finalized) return -22;
    u->fusion = mask & u->fusion_avail;
    return u->fusion;
}

int emu_unet_temb_total(const emu_unet* u) { return u ? u->temb_total : 0; }

size_t emu_unet_workspace_bytes(const emu_unet* u, int H, int W) {
    if (!u || !u->finalized) return 0;
    return plan_ws(u, H, W, nullptr).total;
}

// context cache layout, per transformer block in traversal order: KV rows [B*n, 2C] then Vt [B, heads, 64, npad]
size_t emu_unet_context_bytes(const emu_unet* u, int n_ctx) {
    if (!u || !u->finalized) return 0;
    const size_t npad = (size_t)(n_ctx + 63) / 64 * 64;
    size_t el = 0;
    auto add = [&](const Transformer& t) { el += t.blocks.size() * ((size_t)2 * n_ctx * 2 * t.c + (size_t)2 * t.c * npad); };
    for (int i = 0; i < 3; ++i) { for (auto& t : u->down_tr[i]) add(t); for (auto& t : u->up_tr[i]) add(t); }
    add(u->mid_tr);
    return el * 2 + (size_t)2 * u->cfg.temb_dim * 2 + 256;
}

// Once per generation (prompt fixed over all steps): cross-attention K / V^T of every transformer block from the
// prompt embeddings ctx [2, n, cross] (cond first), and aug_emb = add_embedding(cat(text_embeds, time_ids emb))
// from add_in [2, proj_class_in].  Results live in `cache` (emu_unet_context_bytes), owned by the caller.
int emu_unet_set_context(emu_unet* u, const void* ctx_tokens, int n_ctx, const void* add_in, int add_dim, void* cache,
                         size_t cache_bytes, void* workspace, size_t ws_bytes, emu_stream_t s_) {
    if (!u || !u->finalized || !ctx_tokens || !add_in || !cache) return -22;
    if (cache_bytes < emu_unet_context_bytes(u, n_ctx)) return ufail(u, -12, "emu_unet_set_context: cache too small");
    if (ws_bytes < (size_t)2 * u->cfg.temb_dim * 2) return ufail(u, -12, "emu_unet_set_context: workspace too small");
    hipStream_t s = S(s_);
    const int Bn = 2, cross = u->cfg.cross_dim;
    const int npad = (n_ctx + 63) / 64 * 64;
    bf16_t* base = reinterpret_cast<bf16_t*>(cache);
    size_t off = 0;
    u->n_ctx = n_ctx;
    auto run = [&](Transformer& t) -> int {
        for (TBlock& tb : t.blocks) {
            tb.ctx_off = off;
            bf16_t* kv = base + off;
            bf16_t* vt = kv + (size_t)Bn * n_ctx * 2 * t.c;
            UTRY(gemm(u, B16(ctx_tokens), tb.kv2, nullptr, nullptr, kv, Bn * n_ctx, 2 * t.c, cross, cross, 0, 2 * t.c, EPI_NONE, s));
            TransposeVArgs tv{kv + t.c, (long)n_ctx * 2 * t.c, 64L, (long)2 * t.c, vt, Bn, t.heads, n_ctx, 64, npad};
            UTRY(launch_transpose_v(tv, s));
            off += (size_t)Bn * n_ctx * 2 * t.c + (size_t)Bn * t.c * npad;
        }
        return 0;
    };
    for (int i = 0; i < 3; ++i) for (auto& t : u->down_tr[i]) { int st = run(t); if (st) return st; }
    { int st = run(u->mid_tr); if (st) return st; }
    for (int i = 0; i < 3; ++i) for (auto& t : u->up_tr[i]) { int st = run(t); if (st) return st; }
    bf16_t* aug = base + off;
    bf16_t* tmp = reinterpret_cast<bf16_t*>(workspace);
    UTRY(gemm(u, B16(add_in), u->ae1w, u->ae1b, nullptr, tmp, Bn, u->cfg.temb_dim, add_dim, add_dim, 0, u->cfg.temb_dim, EPI_SILU, s));
    UTRY(gemm(u, tmp, u->ae2w, u->ae2b, nullptr, aug, Bn, u->cfg.temb_dim, u->cfg.temb_dim, u->cfg.temb_dim, 0, u->cfg.temb_dim, EPI_NONE, s));
    u->ctx_cache = base;
    u->aug_emb = aug;
    return 0;
}

// The UNet forward on the im2col'ed, already scaled input in ws.colin; result (noise prediction, NHWC [2*HW, 4]) in out.
static int unet_body(emu_unet* u, const Ws& w, int H, int W, const bf16_t* temb_table, const int32_t* step, bf16_t* out, hipStream_t s) {
    const emu_unet_cfg& c = u->cfg;
    const int Bn = u->cfg_half < 0 ? 2 : 1;
    const bf16_t* aug = u->aug_emb + (u->cfg_half < 0 ? 0 : (size_t)u->cfg_half * c.temb_dim);
    // ---- time embedding: Timesteps row of this step -> MLP; + aug_emb; SiLU; every resnet's time_emb_proj in one GEMV
    UTRY(launch_gather_step_row(temb_table, step, w.temb_in, Bn, c.ch[0], s));
    UTRY(gemm(u, w.temb_in, u->te1w, u->te1b, nullptr, w.e1, Bn, c.temb_dim, c.ch[0], c.ch[0], 0, c.temb_dim, EPI_SILU, s));
    UTRY(gemm(u, w.e1, u->te2w, u->te2b, nullptr, w.emb, Bn, c.temb_dim, c.temb_dim, c.temb_dim, 0, c.temb_dim, EPI_NONE, s));
    UTRY(launch_add_silu(w.emb, aug, nullptr, w.semb, Bn * c.temb_dim, s));
    UTRY(gemm(u, w.semb, u->tpw, u->tpb, nullptr, w.temb_all, Bn, u->temb_total, c.temb_dim, c.temb_dim, 0, u->temb_total, EPI_NONE, s));
    // ---- conv_in
    int hs[3] = {H, (H + 1) / 2, (H + 3) / 4}, wsz[3] = {W, (W + 1) / 2, (W + 3) / 4};
    int k = 0;
    UTRY(gemm(u, w.colin, u->conv_in_w, u->conv_in_b, nullptr, w.skip[k], Bn * H * W, c.ch[0], c.kpad_in, c.kpad_in, 0, c.ch[0], EPI_NONE, s));
    const bf16_t* h = w.skip[k++];
    // ---- down
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < c.layers_per_block; ++j) {
            bf16_t* dst = c.attn[i] ? w.hA : w.skip[k];
            { int st = run_resnet(u, u->down_res[i][j], h, dst, w, Bn, hs[i], wsz[i], s); if (st) return st; }
            if (c.attn[i]) { int st = run_transformer(u, u->down_tr[i][j], dst, w.skip[k], w, Bn, hs[i], wsz[i], s); if (st) return st; }
            h = w.skip[k++];
        }
        if (i < 2) {
            UTRY(conv3(u, h, u->down_ds_w[i], u->down_ds_b[i], nullptr, 0, nullptr, w.skip[k], Bn, hs[i], wsz[i], c.ch[i], c.ch[i], CONV_3X3_S2, s));
            h = w.skip[k++];
        }
    }
    // ---- mid
    { int st = run_resnet(u, u->mid_res[0], h, w.hA, w, Bn, hs[2], wsz[2], s); if (st) return st; }
    { int st = run_transformer(u, u->mid_tr, w.hA, w.hB, w, Bn, hs[2], wsz[2], s); if (st) return st; }
    { int st = run_resnet(u, u->mid_res[1], w.hB, w.hA, w, Bn, hs[2], wsz[2], s); if (st) return st; }
    bf16_t* cur = w.hA;
    bf16_t* oth = w.hB;
    int cur_c = c.ch[2];
    // ---- up
    for (int i = 0; i < 3; ++i) {
        const int lvl = 2 - i;
        for (size_t j = 0; j < u->up_res[i].size(); ++j) {
            const Resnet& r = u->up_res[i][j];
            const bf16_t* skip = w.skip[--k];
            const int skip_c = r.cin - cur_c;
            UTRY(launch_concat_channels(cur, skip, w.cat, Bn * hs[lvl] * wsz[lvl], cur_c, skip_c, s));
            if (c.attn[lvl]) {
                { int st = run_resnet(u, r, w.cat, oth, w, Bn, hs[lvl], wsz[lvl], s); if (st) return st; }
                { int st = run_transformer(u, u->up_tr[i][j], oth, cur, w, Bn, hs[lvl], wsz[lvl], s); if (st) return st; }
            } else {
                { int st = run_resnet(u, r, w.cat, oth, w, Bn, hs[lvl], wsz[lvl], s); if (st) return st; }
                std::swap(cur, oth);
            }
            cur_c = r.cout;
        }
        if (i < 2) {
            UTRY(conv3(u, cur, u->up_us_w[i], u->up_us_b[i], nullptr, 0, nullptr, oth, Bn, hs[lvl], wsz[lvl], cur_c, cur_c, CONV_3X3_UP2, s));
            std::swap(cur, oth);
        }
    }
    // ---- out
    UTRY(launch_groupnorm(cur, u->cno_g, u->cno_b, w.gn, w.gnws, Bn, H * W, c.ch[0], c.groups, c.gn_eps, 1, s));
    UTRY(conv3(u, w.gn, u->cout_w, u->cout_b, nullptr, 0, nullptr, out, Bn, H, W, c.ch[0], c.out_ch, CONV_3X3, s));
    return 0;
}

int emu_unet_step(emu_unet* u, void* latents, int H, int W, const void* temb_table, const void* sigmas, int32_t* step_dev,
                  float guidance, void* workspace, size_t ws_bytes, emu_stream_t s_) {
    if (!u || !u->finalized || !latents || !temb_table || !sigmas || !step_dev) return -22;
    if (!u->ctx_cache) return ufail(u, -22, "emu_unet_step: emu_unet_set_context has not been called");
    if ((H & 3) || (W & 3)) return ufail(u, -22, "emu_unet_step: latent H, W must be multiples of 4");
    if (u->cfg_half >= 0) return ufail(u, -22, "emu_unet_step: one row of the CFG pair is set (emu_unet_set_cfg_half): emu_unet_forward + emu_unet_cfg_euler_step");
    const Ws w = plan_ws(u, H, W, workspace);
    u->splitk = w.splitk; u->splitk_floats = w.splitk_floats;
    if (w.total > ws_bytes) return ufail(u, -12, "emu_unet_step: workspace too small");
    hipStream_t s = S(s_);
    UTRY(launch_unet_prep_input(reinterpret_cast<bf16_t*>(latents), reinterpret_cast<const float*>(sigmas), step_dev, w.colin,
                                u->cfg.in_ch, H, W, u->cfg.kpad_in, s));
    bf16_t* eps = w.t1;                                  // [2*H*W, 4], t1 is free once the last resnet is done
    { int st = unet_body(u, w, H, W, B16(temb_table), step_dev, eps, s); if (st) return st; }
    UTRY(launch_cfg_euler_step(eps, reinterpret_cast<bf16_t*>(latents), reinterpret_cast<const float*>(sigmas), step_dev, guidance,
                               u->cfg.in_ch, H * W, s));
    return 0;
}

// Bare UNet forward for parity tests: scaled NCHW input of ONE sample is taken from `latents` with sigma table entry
// sigmas[step] (pass sigma = 0 for no scaling), noise prediction returned as NHWC [2*H*W, 4] in eps_out.
int emu_unet_forward(emu_unet* u, const void* latents, int H, int W, const void* temb_table, const void* sigmas,
                     const int32_t* step_dev, void* eps_out, void* workspace, size_t ws_bytes, emu_stream_t s_) {
    if (!u || !u->finalized || !latents || !temb_table || !sigmas || !step_dev || !eps_out) return -22;
    if (!u->ctx_cache) return ufail(u, -22, "emu_unet_forward: emu_unet_set_context has not been called");
    const Ws w = plan_ws(u, H, W, workspace);
    u->splitk = w.splitk; u->splitk_floats = w.splitk_floats;
    if (w.total > ws_bytes) return ufail(u, -12, "emu_unet_forward: workspace too small");
    hipStream_t s = S(s_);
    UTRY(launch_unet_prep_input(B16(latents), reinterpret_cast<const float*>(sigmas), step_dev, w.colin, u->cfg.in_ch, H, W,
                                u->cfg.kpad_in, s));
    return unet_body(u, w, H, W, B16(temb_table), step_dev, reinterpret_cast<bf16_t*>(eps_out), s);
}

int emu_unet_set_cfg_half(emu_unet* u, int half) {
    if (!u || half < -1 || half > 1) return -22;
    u->cfg_half = half;
    return 0;
}

int emu_unet_cfg_euler_step(emu_unet* u, const void* eps_pair, void* latents, int H, int W, const void* sigmas, int32_t* step_dev,
                            float guidance, emu_stream_t s_) {
    if (!u || !eps_pair || !latents || !sigmas || !step_dev) return -22;
    UTRY(launch_cfg_euler_step(B16(eps_pair), reinterpret_cast<bf16_t*>(latents), reinterpret_cast<const float*>(sigmas), step_dev, guidance,
                               u->cfg.in_ch, H * W, S(s_)));
    return 0;
}

}  // extern "C"
