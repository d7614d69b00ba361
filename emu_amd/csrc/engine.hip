// C ABI of libemu_hip.so (declared in include/emu_hip.h): context + RCCL communicator, primitive operator
// entry points, and the LLaMA / EVA-ViT engines that chain the kernels on one HIP stream without any
// allocation or synchronisation (so whole forwards are hipGraph / stream-capture safe).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/emu_hip.h"
#include "kernels.h"

struct emu_ctx {
    int device = 0, tp_rank = 0, tp_size = 1;
    ncclComm_t comm = nullptr;
    EmuP2p* p2p = nullptr;                          // one-shot all-reduce blocks (p2p.hip); used only once enabled
    bool p2p_on = false;
    // tensor-parallel prefill in two concurrent lanes (emu_llama_set_tp_overlap): the second half of a prompt's rows runs on
    // lane_stream beside the caller's stream; the events chain the two
    hipStream_t lane_stream = nullptr;
    hipEvent_t ar_ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    std::string err;
};

namespace {
inline hipStream_t S(emu_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
inline const bf16_t* B(const void* p) { return reinterpret_cast<const bf16_t*>(p); }
inline bf16_t* B(void* p) { return reinterpret_cast<bf16_t*>(p); }
inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

int fail(emu_ctx* c, int code, const char* what) {
    if (c) { char buf[256]; snprintf(buf, sizeof buf, "%s (status %d)", what, code); c->err = buf; }
    return code;
}
#define TRY(c, expr) do { int st__ = (expr); if (st__ != 0) return fail((c), st__, #expr); } while (0)

// HIP-event timing of the weight-streaming GEMV launches (bench.py roofline leg; eager mode only)
struct GemvProfiler {
    bool on = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
    size_t used = 0;
    double bytes = 0.0;
} g_prof;

// launch profiler: event pairs + what was launched, aggregated on read
struct LaunchProfiler {
    bool on = false;
    struct Rec { hipEvent_t a, b; const char* klass; int M, N, K, tag; double flops; };
    std::vector<Rec> recs;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;
    size_t used = 0;
    hipEvent_t cur_a = nullptr, cur_b = nullptr;
} g_lprof;

int linear(const bf16_t* A, const bf16_t* W, const bf16_t* bias, const bf16_t* res, const bf16_t* norm_w,
           bf16_t* C, int M, int N, int K, int lda, int ldw, int ldres, int ldc, float eps, int epi, hipStream_t s,
           const float* wscale = nullptr, float* splitk = nullptr, size_t splitk_floats = 0) {
    if (wscale && M > 2) return -22;                 // fp8 weights are a decode-only stream
    // rows <= 8 stream the weights through the GEMV family (M >= 2 without a fused norm: skinny MFMA kernel); 9..16
    // rows too when the MFMA kernel covers the shape -- a 128-row GEMM tile would be > 87 % padding there
    const bool skinny = M > 8 && M <= 16 && !norm_w && (K & 31) == 0 && (ldw & 7) == 0 && (lda & 7) == 0 &&
                        (epi == EPI_NONE || epi == EPI_RESID || epi == EPI_SWIGLU || epi == EPI_SILU || epi == EPI_GELU);
    // A/B aid (emu_gemm_tune bit 2): 2..32 rows without a fused norm go through the thin MFMA tile instead
    if (M <= 8 || skinny) {
        GemvArgs g{A, W, norm_w, bias, res, C, M, N, K, lda, ldw, ldres, ldc, eps, epi, 0, wscale};
        if (!g_prof.on) return launch_gemv(g, s);
        if (g_prof.used == g_prof.ev.size()) {
            hipEvent_t a, b;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return -12;
            g_prof.ev.emplace_back(a, b);
        }
        auto& e = g_prof.ev[g_prof.used++];
        g_prof.bytes += (wscale ? 1.0 : 2.0) * (double)N * (double)K;
        (void)hipEventRecord(e.first, s);
        const int st = launch_gemv(g, s);
        (void)hipEventRecord(e.second, s);
        return st;
    }
    if (norm_w) return -22;
    GemmArgs g{A, W, bias, res, C, M, N, K, lda, ldw, ldres, ldc, epi, ConvGeom{0, 0, 0, 0, 0, 0}, nullptr, 0, 0};
    g.partial = splitk; g.partial_floats = splitk_floats;
    return launch_gemm(g, s);
}
// the merged o_proj of a short shard (gemv_merge.hip) under the same GEMV launch profiler as linear()
int gemv_merge_profiled(const GemvMergeArgs& g, hipStream_t s) {
    if (!g_prof.on) return launch_gemv_merge(g, s);
    if (g_prof.used == g_prof.ev.size()) {
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return -12;
        g_prof.ev.emplace_back(a, b);
    }
    auto& e = g_prof.ev[g_prof.used++];
    g_prof.bytes += 2.0 * (double)g.N * (double)g.K;
    (void)hipEventRecord(e.first, s);
    const int st = launch_gemv_merge(g, s);
    (void)hipEventRecord(e.second, s);
    return st;
}
}  // namespace

int emu_ctx_fail(emu_ctx* c, int code, const char* what) { return fail(c, code, what); }

bool emu_prof_on() { return g_lprof.on; }
void emu_prof_begin(hipStream_t s) {
    g_lprof.cur_a = nullptr;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;             // eager launches only: no events inside a stream capture
    if (hipStreamIsCapturing(s, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return; }
    if (g_lprof.used == g_lprof.pool.size()) {
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { g_lprof.cur_a = nullptr; return; }
        g_lprof.pool.emplace_back(a, b);
    }
    g_lprof.cur_a = g_lprof.pool[g_lprof.used].first;
    g_lprof.cur_b = g_lprof.pool[g_lprof.used].second;
    ++g_lprof.used;
    (void)hipEventRecord(g_lprof.cur_a, s);
}
void emu_prof_drop() {                                   // the launch between begin and end did not happen (-95 / -22): forget the pair
    if (g_lprof.cur_a && g_lprof.used) --g_lprof.used;
    g_lprof.cur_a = nullptr;
}
void emu_prof_end(hipStream_t s, const char* klass, int M, int N, int K, int tag, double flops) {
    if (!g_lprof.cur_a) return;
    (void)hipEventRecord(g_lprof.cur_b, s);
    g_lprof.recs.push_back({g_lprof.cur_a, g_lprof.cur_b, klass, M, N, K, tag, flops});
    g_lprof.cur_a = nullptr;
}

extern "C" {

int emu_version(void) { return 3; }      // ABI version: emu_amd/_lib.py::ABI_VERSION must match

void emu_set_splitk_scratch(void* ptr, size_t bytes) { emu_gemm_set_splitk_scratch(reinterpret_cast<float*>(ptr), bytes / sizeof(float)); }
void emu_gemm_force_config(int cfg) { emu_gemm_force_config_set(cfg); }
void emu_gemm_tune(int mask) { emu_gemm_tune_set(mask); }
void emu_gemm_trace(void* buf) { emu_gemm_trace_set(reinterpret_cast<unsigned long long*>(buf)); }
void emu_gemm_trace_nth(long n) { emu_gemm_trace_select(n); }
int emu_gemm_trace_built(void) {
#ifdef EMU_TRACE
    return 1;
#else
    return 0;
#endif
}

int emu_profile_launches(int enable) {
    g_lprof.on = enable != 0;
    g_lprof.recs.clear();
    g_lprof.used = 0;
    return 0;
}
int emu_profile_launches_read(emu_prof_row* rows, int cap) {
    int n = 0;
    for (const auto& r : g_lprof.recs) {
        if (hipEventSynchronize(r.b) != hipSuccess) return -5;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) return -5;
        int i = 0;
        for (; i < n; ++i)
            if (!strcmp(rows[i].klass, r.klass) && rows[i].M == r.M && rows[i].N == r.N && rows[i].K == r.K && rows[i].tag == r.tag) break;
        if (i == n) {
            if (n == cap) continue;
            memset(&rows[n], 0, sizeof rows[n]);
            strncpy(rows[n].klass, r.klass, sizeof rows[n].klass - 1);
            rows[n].M = r.M; rows[n].N = r.N; rows[n].K = r.K; rows[n].tag = r.tag;
            ++n;
        }
        rows[i].launches += 1; rows[i].ms += ms; rows[i].flops += r.flops;
    }
    return n;
}

int emu_profile_gemv(int enable) {
    g_prof.on = enable != 0;
    g_prof.used = 0;
    g_prof.bytes = 0.0;
    return 0;
}

int emu_profile_gemv_read(double* total_ms, double* weight_bytes, long* launches) {
    double t = 0.0;
    for (size_t i = 0; i < g_prof.used; ++i) {
        hipError_t e = hipEventSynchronize(g_prof.ev[i].second);
        if (e != hipSuccess) return (int)e;
        float ms = 0.f;
        e = hipEventElapsedTime(&ms, g_prof.ev[i].first, g_prof.ev[i].second);
        if (e != hipSuccess) return (int)e;
        t += ms;
    }
    if (total_ms) *total_ms = t;
    if (weight_bytes) *weight_bytes = g_prof.bytes;
    if (launches) *launches = (long)g_prof.used;
    return 0;
}

int emu_ctx_create(int device, int tp_rank, int tp_size, emu_ctx** out) {
    if (!out || tp_size < 1 || tp_rank < 0 || tp_rank >= tp_size) return -22;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || device < 0 || device >= n) return e != hipSuccess ? (int)e : -19;
    emu_ctx* c = new emu_ctx();
    c->device = device; c->tp_rank = tp_rank; c->tp_size = tp_size;
    *out = c;
    return 0;
}

void emu_ctx_destroy(emu_ctx* ctx) {
    if (!ctx) return;
    if (ctx->comm) ncclCommDestroy(ctx->comm);
    emu_p2p_destroy(ctx->p2p);
    for (hipEvent_t e : ctx->ar_ev) if (e) (void)hipEventDestroy(e);
    if (ctx->lane_stream) (void)hipStreamDestroy(ctx->lane_stream);
    delete ctx;
}

const char* emu_last_error(const emu_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int emu_tp_unique_id(void* out128) {
    static_assert(sizeof(ncclUniqueId) == 128, "RCCL unique id is 128 bytes");
    ncclUniqueId id;
    ncclResult_t r = ncclGetUniqueId(&id);
    if (r != ncclSuccess) return 1000 + (int)r;
    memcpy(out128, &id, sizeof id);
    return 0;
}

int emu_tp_init(emu_ctx* ctx, const void* id128) {
    if (!ctx) return -22;
    if (ctx->comm) return 0;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) return fail(ctx, (int)e, "hipSetDevice");
    ncclResult_t r = ncclCommInitRank(&ctx->comm, ctx->tp_size, id, ctx->tp_rank);
    if (r != ncclSuccess) { ctx->comm = nullptr; return fail(ctx, 1000 + (int)r, "ncclCommInitRank"); }
    return 0;
}

int emu_tp_p2p_create(emu_ctx* ctx, void* handle64_out) {
    if (!ctx || !handle64_out) return -22;
    if (ctx->p2p) return fail(ctx, -17, "emu_tp_p2p_create: already created");
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) return fail(ctx, (int)e, "hipSetDevice");
    ctx->p2p = emu_p2p_create(ctx->tp_rank, ctx->tp_size, handle64_out);
    return ctx->p2p ? 0 : fail(ctx, -12, "emu_tp_p2p_create: comm block allocation / IPC export failed");
}

int emu_tp_p2p_open(emu_ctx* ctx, const void* handles, int timeout_ms) {
    if (!ctx || !ctx->p2p) return -22;
    hipError_t e = hipSetDevice(ctx->device);              // the peer blocks are mapped for this context's device
    if (e != hipSuccess) return fail(ctx, (int)e, "hipSetDevice");
    emu_p2p_set_timeout_ms(ctx->p2p, timeout_ms);
    int st = emu_p2p_open(ctx->p2p, handles);
    return st == 0 ? 0 : fail(ctx, st, "hipIpcOpenMemHandle");
}

int emu_tp_p2p_allreduce_bf16(emu_ctx* ctx, void* buf, size_t n, emu_stream_t s) {
    if (!ctx || !ctx->p2p) return -22;
    int st = emu_p2p_allreduce(ctx->p2p, reinterpret_cast<bf16_t*>(buf), n, S(s));
    return st == 0 ? 0 : fail(ctx, st, "emu_p2p_allreduce");
}

int emu_tp_p2p_enable(emu_ctx* ctx, int on) {
    if (!ctx || (on && !ctx->p2p)) return -22;
    ctx->p2p_on = on != 0;
    return 0;
}

int emu_tp_p2p_set_fenced(emu_ctx* ctx, int fenced) {
    if (!ctx || !ctx->p2p) return -22;
    emu_p2p_set_fenced(ctx->p2p, fenced);
    return 0;
}
int emu_tp_p2p_fenced(const emu_ctx* ctx) { return ctx && ctx->p2p ? emu_p2p_fenced(ctx->p2p) : -1; }
unsigned int emu_tp_p2p_giveups(void) { return emu_p2p_giveups_read(); }

int emu_allreduce_bf16(emu_ctx* ctx, void* buf, size_t n, emu_stream_t s) {
    if (!ctx) return -22;
    // small messages (decode: 13 KB) take the one-shot P2P path once the host has enabled it; without a RCCL communicator
    // (two ranks on one GPU in the tests) everything does, in slot-sized chunks
    if (ctx->p2p_on && (n * sizeof(bf16_t) <= EMU_P2P_SLOT_BYTES || !ctx->comm)) {
        int st = emu_p2p_allreduce(ctx->p2p, reinterpret_cast<bf16_t*>(buf), n, S(s));
        return st == 0 ? 0 : fail(ctx, st, "emu_p2p_allreduce");
    }
    if (!ctx->comm) {
        if (ctx->tp_size == 1) return 0;
        return fail(ctx, -107, "emu_allreduce_bf16: communicator not initialised (emu_tp_init)");
    }
    ncclResult_t r = ncclAllReduce(buf, buf, n, ncclBfloat16, ncclSum, ctx->comm, S(s));
    return r == ncclSuccess ? 0 : fail(ctx, 1000 + (int)r, "ncclAllReduce");
}

// ---------------------------------------------------------------------------------------------- primitives
int emu_linear_bf16(const void* A, const void* W, const void* bias, const void* res, const void* norm_w, void* C,
                    int M, int N, int K, int lda, int ldw, int ldres, int ldc, float eps, int epi, emu_stream_t s) {
    return linear(B(A), B(W), B(bias), B(res), B(norm_w), B(C), M, N, K, lda, ldw, ldres, ldc, eps, epi, S(s));
}
int emu_linear_fp8w_bf16(const void* A, const void* W8, const float* wscale, const void* bias, const void* res,
                         const void* norm_w, void* C, int M, int N, int K, int lda, int ldw, int ldres, int ldc, float eps,
                         int epi, emu_stream_t s) {
    if (!wscale) return -22;
    return linear(B(A), B(W8), B(bias), B(res), B(norm_w), B(C), M, N, K, lda, ldw, ldres, ldc, eps, epi, S(s), wscale);
}
int emu_linear_fp8_bf16(const void* A8, const float* a_scale, const void* W8, const float* w_scale, const void* bias,
                        const void* res, void* C, int M, int N, int K, int lda, int ldw, int ldres, int ldc, int epi,
                        emu_stream_t s) {
    if (!A8 || !W8 || !a_scale || !w_scale || !C) return -22;
    GemmArgs g{B(A8), B(W8), B(bias), B(res), B(C), M, N, K, lda, ldw, ldres, ldc, epi, ConvGeom{0, 0, 0, 0, 0, 0}, nullptr, 0, 0};
    g.a_scale = a_scale; g.w_scale = w_scale;
    return launch_gemm_fp8(g, S(s));
}
int emu_linear_fused_bf16(const void* A, const void* W, const void* bias, const void* res, void* C, int M, int N, int K,
                          int lda, int ldw, int ldres, int ldc, int epi, const emu_linear_fx* fx, emu_stream_t s) {
    if (!A || !W || !C || M <= 8) return -22;
    GemmArgs g{B(A), B(W), B(bias), B(res), B(C), M, N, K, lda, ldw, ldres, ldc, epi, ConvGeom{0, 0, 0, 0, 0, 0}, nullptr, 0, 0};
    if (fx) {
        g.row_stats_out = fx->row_stats_out;
        g.ln_c = fx->ln_c; g.ln_d = fx->ln_d; g.ln_stats = fx->ln_stats; g.ln_slots = fx->ln_slots; g.ln_eps = fx->ln_eps;
        g.vt_out = B(fx->vt_out); g.vt_col0 = fx->vt_col0; g.vt_s = fx->vt_s; g.vt_spad = fx->vt_spad;
        g.cross_k = B(fx->cross_k); g.cross_vt = B(fx->cross_vt); g.cross_ldk = fx->cross_ldk; g.cross_n = fx->cross_n;
        g.cross_npad = fx->cross_npad; g.cross_rows = fx->cross_rows; g.cross_scale = fx->cross_scale;
    }
    return launch_gemm(g, S(s));
}
int emu_quantize_fp8_rows(const void* w, int ldw, void* q, int ldq, float* scale, int N, int K, emu_stream_t s) {
    if (!w || !q || !scale) return -22;
    return launch_quant_fp8_rows(B(w), ldw, reinterpret_cast<uint8_t*>(q), ldq, scale, N, K, S(s));
}
int emu_rmsnorm_bf16(const void* x, const void* w, void* y, int rows, int cols, int ldx, int ldy, float eps, emu_stream_t s) {
    return launch_rmsnorm(B(x), B(w), B(y), rows, cols, ldx, ldy, eps, S(s));
}
int emu_layernorm_bf16(const void* x, const void* w, const void* b, const void* res, void* y, int rows, int cols,
                       float eps, emu_stream_t s) {
    return launch_layernorm(B(x), B(w), B(b), B(res), B(y), rows, cols, eps, S(s));
}
int emu_layernorm_q8_bf16(const void* x, const void* w, const void* b, const void* res, void* y, void* q, float* scale, int rows,
                          int cols, float eps, emu_stream_t s) {
    if (!x || !w || !b || !q || !scale) return -22;
    return launch_layernorm_q8(B(x), B(w), B(b), B(res), reinterpret_cast<bf16_t*>(y), reinterpret_cast<uint8_t*>(q), scale, rows, cols,
                               eps, S(s));
}
static int chain_out_len(const emu_chain_op& o) { return o.epi == EPI_SWIGLU ? o.N / 2 : o.N; }
size_t emu_gemv_chain_granule_bytes(const emu_chain_op* ops, int nops) {
    size_t b = 0;
    for (int i = 0; ops && i + 1 < nops; ++i)
        if (ops[i + 1].x_from_prev) b += align_up((size_t)chain_out_len(ops[i]) * 4);
    return b;
}
int emu_gemv_chain_bf16(emu_ctx* ctx, const emu_chain_op* ops, int nops, void* granules, size_t granule_bytes, unsigned int* err,
                        emu_stream_t s_) {
    if (!ctx || !ops || nops < 1 || nops > ENG_MAX_OPS || !err) return -22;
    const size_t need = emu_gemv_chain_granule_bytes(ops, nops);
    if (need > granule_bytes || (need && !granules)) return fail(ctx, -12, "emu_gemv_chain_bf16: granule buffer too small");
    hipStream_t s = S(s_);
    EngArgs a{};
    a.nops = nops; a.err = err; a.limit_ticks = 200000000LL;           // 2 s
    if (const char* e = getenv("EMU_ENGINE_TIMEOUT_MS")) { const long ms = atol(e); if (ms > 0) a.limit_ticks = ms * 100000LL; }   // tools
    if (const char* e = getenv("EMU_ENGINE_LOADERS")) a.nload = atoi(e);
    if (const char* e = getenv("EMU_ENGINE_DBG")) a.dbg = atoi(e);
    int dev = 0, ncu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -5;
    a.ncu = ncu;
    char* gp = reinterpret_cast<char*>(granules);
    size_t off = 0;
    for (int i = 0; i < nops; ++i) {
        const emu_chain_op& c = ops[i];
        EngOp& o = a.op[i];
        o.W = B(c.W); o.N = c.N; o.K = c.K; o.gain = B(c.gain); o.eps = c.eps; o.epi = c.epi; o.res = B(c.res);
        o.vw = emu_gemv_partition(c.N, c.K, c.gain != nullptr, c.epi);
        o.x_src = c.x_from_prev ? 1 : 0;
        o.xg = B(c.x);
        if (c.x_from_prev) {
            if (i == 0 || chain_out_len(ops[i - 1]) != c.K) return fail(ctx, -22, "emu_gemv_chain_bf16: op input is not the previous op's output");
            o.xgran = a.op[i - 1].ogran;
        }
        const bool to_next = i + 1 < nops && ops[i + 1].x_from_prev;
        o.out_dst = to_next ? 1 : 0;
        o.out = B(c.out);
        if (to_next) { o.ogran = reinterpret_cast<uint32_t*>(gp + off); off += align_up((size_t)chain_out_len(c) * 4); }
    }
    if (need && hipMemsetAsync(granules, 0, need, s) != hipSuccess) return fail(ctx, -5, "emu_gemv_chain_bf16: hipMemsetAsync");
    const int st = launch_decode_engine(a, s);
    return st == 0 || st == -95 ? st : fail(ctx, st, "emu_gemv_chain_bf16");
}
int emu_prefetch(const void* ptr, size_t bytes, int workgroups, emu_stream_t s) { return launch_prefetch(ptr, bytes, workgroups, S(s)); }
int emu_softmax_rows_bf16(void* x, const void* bias, int rows, int cols, int ld, int ld_bias, float scale, emu_stream_t s) {
    return launch_softmax_rows(B(x), B(bias), rows, cols, ld, ld_bias, scale, S(s));
}
int emu_embed_gather_bf16(const int32_t* ids, const void* table, void* out, int n_tok, int hidden, int vocab, emu_stream_t s) {
    return launch_embed_gather(ids, B(table), B(out), n_tok, hidden, vocab, S(s));
}
int emu_scatter_rows_bf16(const void* src, const int32_t* dst_rows, void* out, int n_rows, int hidden, emu_stream_t s) {
    return launch_scatter_rows(B(src), dst_rows, B(out), n_rows, hidden, S(s));
}
int emu_argmax_bf16(const void* logits, int ld, int rows, int vocab, int suppress_id, int32_t* out, emu_stream_t s) {
    return launch_argmax(B(logits), ld, rows, vocab, suppress_id, out, S(s));
}
int emu_patchify(const void* image, int image_is_f32, void* out, int Bn, int C, int HW, int patch, int Kpad, emu_stream_t s) {
    return launch_patchify(image, image_is_f32, B(out), Bn, C, HW, patch, Kpad, S(s));
}
int emu_vit_assemble_bf16(const void* patches, const void* cls, const void* pos, void* x, int Bn, int T, int C, emu_stream_t s) {
    return launch_vit_assemble(B(patches), B(cls), B(pos), B(x), Bn, T, C, S(s));
}
int emu_avgpool_tokens_bf16(const void* x, void* out, int Bn, int g, int C, int stride, emu_stream_t s) {
    return launch_avgpool_tokens(B(x), B(out), Bn, g, C, stride, S(s));
}
int emu_rope_kv_append_bf16(void* qkv, const void* cos, const void* sin, const int32_t* pos, const int32_t* slot,
                            void* kcache, void* vcache, int Bn, int T, int H, int D, int S_max, emu_stream_t s) {
    RopeKvArgs a{B(qkv), B(cos), B(sin), pos, slot, B(kcache), B(vcache), Bn, T, H, D, S_max};
    return launch_rope_kv(a, S(s));
}
int emu_transpose_v_bf16(const void* v, long v_sb, long v_sh, long v_ss, void* vt, int Bn, int H, int Sn, int D,
                         int S_pad, emu_stream_t s) {
    TransposeVArgs a{B(v), v_sb, v_sh, v_ss, B(vt), Bn, H, Sn, D, S_pad};
    return launch_transpose_v(a, S(s));
}
int emu_flash_attn_bf16(const void* q, long q_sb, long q_sh, long q_ss, const void* k, long k_sb, long k_sh, long k_ss,
                        const void* vt, void* o, long o_sb, long o_sh, long o_ss, const int32_t* kstart, int Bn, int H,
                        int Sq, int Sk, int Sk_pad, int D, int causal, float scale, emu_stream_t s) {
    FlashArgs a{B(q), q_sb, q_sh, q_ss, B(k), k_sb, k_sh, k_ss, B(vt), B(o), o_sb, o_sh, o_ss, kstart,
                Bn, H, Sq, Sk, Sk_pad, D, causal, scale};
    return launch_flash_attn(a, S(s));
}
size_t emu_decode_attn_ws_bytes(int Bn, int H, int D, int ctx_max) {
    return (size_t)Bn * H * decode_attn_nsplit(ctx_max) * (D + 2) * sizeof(float);
}
int emu_decode_attn_bf16(const void* q, long q_sb, long q_sh, const void* kcache, const void* vcache, void* o,
                         long o_sb, long o_sh, const int32_t* kstart, const int32_t* ctx_dev, int ctx, int ctx_max,
                         void* ws, int Bn, int H, int D, int S_max, float scale, emu_stream_t s) {
    DecodeAttnArgs a{B(q), q_sb, q_sh, B(kcache), B(vcache), B(o), o_sb, o_sh, kstart, ctx_dev,
                     reinterpret_cast<float*>(ws), Bn, H, D, S_max, ctx, ctx_max, scale};
    return launch_decode_attn(a, S(s));
}

}  // extern "C"

// =============================================================================================== LLaMA engine
struct emu_llama {
    emu_ctx* ctx;
    emu_llama_cfg cfg;
    struct Layer { const bf16_t *wqkv, *wo, *wgu, *wdown, *ln1, *ln2; };
    std::vector<Layer> layers;
    // optional fp8 (e4m3) copies of the packed weights for the decode stream: bytes + one fp32 scale per output row
    struct Layer8 { const uint8_t *wqkv = nullptr, *wo = nullptr, *wgu = nullptr, *wdown = nullptr;
                    const float *sqkv = nullptr, *so = nullptr, *sgu = nullptr, *sdown = nullptr; };
    std::vector<Layer8> layers8;
    const uint8_t* lm_head8 = nullptr;
    const float* lm_scale8 = nullptr;
    bool fp8_decode = false;
    bool fp8_prefill = false;      // emu_llama_use_fp8(m, 2): W8A8 GEMMs for prefill rows as well
    const bf16_t *final_norm = nullptr, *lm_head = nullptr, *embed = nullptr, *cos = nullptr, *sin = nullptr;
    // tensor parallelism: lm_head holds rows [head_row0, head_row0 + head_rows) of the vocabulary only (emu_llama_set_head_shard);
    // head_rows < 0: the whole matrix
    int head_row0 = 0, head_rows = -1;
    bf16_t *kcache = nullptr, *vcache = nullptr;
    int kv_batch = 0, s_max = 0;
    int kv_share_nb = 0, kv_share_len = 0;       // emu_llama_set_kv_share: beams of a prompt share its cache slots
    int l0 = 0, l1 = -1;                         // emu_llama_set_layer_range: layers [l0, l1) run (l1 < 0: all)
    bool fuse_norm_on = false;                   // sticky: o_proj / down_proj K-slice sums apply the RMSNorm behind them (M > 16, no TP)
    bool prefill_fusion = false;                 // emu_llama_set_prefill_fusion: RoPE + KV append + V^T in the qkv GEMM's epilogue (one-shot: the next T > 1 forward consumes it)
    // decode attention without the combine launch (emu_llama_set_decode_tail, off by default): per (row, head) arrival counters of
    // the split workgroups (zero between launches; owned here: EMU_ARRIVE_INTS ints of device memory)
    int* arrive = nullptr;
    bool decode_tail = false;                    // measured 0.4 % SLOWER than the combine launch (profiles/r04_decode_tail_merge_ab.log): opt-in
    // whole decoder layers of a one-row step in one launch (decode_layer.hip; emu_llama_set_decode_fused)
    int decode_fused = 0;                        // 0: the launches above; 1: fused where the shape / mode allows; 2: + in-kernel all-reduce
    int dl_per_launch = 0;                       // layers per launch (0: all)
    DecodeLayerPtrs* dl_table = nullptr;         // device copy of the layers' weight pointers
    bool dl_dirty = true;
    int* dl_cnt = nullptr;                       // arrival counters of all layers (zeroed at the head of every fused forward)
    size_t dl_cnt_bytes = 0;
    unsigned* dl_err = nullptr;                  // give-up counter
    long dl_forwards = 0;                        // fused forwards issued (tests: the path under test is the one that ran)
    unsigned long long* dl_trace = nullptr;      // emu_llama_set_decode_trace (tools; -DEMU_TRACE twin library only)
    // mode 4: the persistent weight-streaming engine (decode_engine.hip), one launch per layer between the attention launches
    uint32_t* eng_gran = nullptr;                // granule arrays of all layers (zeroed at the head of every forward)
    size_t eng_gran_bytes = 0;
    long eng_forwards = 0;
    // tensor-parallel prefill in two row halves whose all-reduces run on the context's second stream (emu_llama_set_tp_overlap)
    int tp_overlap_rows = 0;                     // 0: off; else the smallest prompt (rows) that takes the two-half schedule
    long ov_forwards = 0;                        // forwards that took it (tests, tools)
};
constexpr int EMU_ARRIVE_INTS = 65536;

namespace {
struct LlamaWs {
    bf16_t *hB, *xn, *qkv, *attn, *act, *vt;
    uint8_t* x8;            // prefill with fp8 weights: the current GEMM's activation rows as e4m3 bytes ...
    float* xs;              // ... and their per-row scales
    float* dec;
    float* splitk;          // prefill only: K-slices of the GEMMs' tail round
    size_t splitk_floats;
    bf16_t* vt2;            // two-lane tensor-parallel prefill (emu_llama_set_tp_overlap): the second lane's own V^T ...
    float* splitk2;         // ... and K-slice scratch (the lanes' GEMMs run concurrently)
    size_t total;
};
LlamaWs llama_ws(const emu_llama* m, int Bn, int T, void* base) {
    const emu_llama_cfg& c = m->cfg;
    const size_t M = (size_t)Bn * T, HD = (size_t)c.heads_local * c.head_dim;
    char* p = reinterpret_cast<char*>(base);
    size_t off = 0;
    auto take = [&](size_t bytes) { char* r = p ? p + off : nullptr; off += align_up(bytes); return r; };
    LlamaWs w;
    w.hB = (bf16_t*)take(M * c.hidden * 2);
    w.xn = (bf16_t*)take(M * c.hidden * 2);
    w.qkv = (bf16_t*)take(M * 3 * HD * 2);
    w.attn = (bf16_t*)take(M * HD * 2);
    w.act = (bf16_t*)take(M * (size_t)c.ffn_local * 2);
    const size_t spad = (size_t)((m->s_max + 63) / 64) * 64;
    w.vt = (bf16_t*)take(T > 1 ? (size_t)Bn * HD * spad * 2 : 0);
    w.dec = (float*)take(decode_fused_ws_floats(Bn, c.heads_local, c.head_dim, m->s_max > 0 ? m->s_max : 1) * sizeof(float));
    w.splitk_floats = M > 16 ? EMU_SPLITK_SCRATCH_FLOATS : 0;
    w.splitk = (float*)take(w.splitk_floats * sizeof(float));
    const size_t kmax = std::max<size_t>(std::max<size_t>(c.hidden, HD), c.ffn_local);
    w.x8 = (uint8_t*)take(M > 16 ? M * kmax : 0);
    w.xs = (float*)take(M > 16 ? M * sizeof(float) : 0);
    const bool lanes = m->tp_overlap_rows > 0 && Bn == 1 && T >= m->tp_overlap_rows;
    w.vt2 = (bf16_t*)take(lanes ? HD * spad * 2 : 0);
    w.splitk2 = (float*)take(lanes ? w.splitk_floats * sizeof(float) : 0);
    w.total = off;
    return w;
}

// out = epi(fp8(A rows, quantised here) x fp8 weights): the prefill form of the fp8 weight set (emu_linear_fp8_bf16)
int linear_q8(const LlamaWs& w, const bf16_t* A, int lda, const uint8_t* W8, const float* wscale, const bf16_t* res, bf16_t* C,
              int M, int N, int K, int ldres, int ldc, int epi, hipStream_t s) {
    int st = launch_quant_fp8_rows(A, lda, w.x8, K, w.xs, M, K, s);
    if (st) return st;
    GemmArgs g{reinterpret_cast<const bf16_t*>(w.x8), reinterpret_cast<const bf16_t*>(W8), nullptr, res, C, M, N, K, K, K, ldres, ldc,
               epi, ConvGeom{0, 0, 0, 0, 0, 0}, nullptr, 0, 0};
    g.a_scale = w.xs; g.w_scale = wscale;
    g.partial = w.splitk; g.partial_floats = w.splitk_floats;
    return launch_gemm_fp8(g, s);
}
// C = epi(A W^T [+ res]) followed by xn = RMSNorm(C) * gain: one row-wise slice-sum launch where the 256x256 tile K-slices every
// tile of the GEMM (GemmArgs::norm_*; launch_gemm answers -95 otherwise), else the GEMM and the rmsnorm launch apart.  Same bits.
int linear_then_rmsnorm(const LlamaWs& w, const bf16_t* A, const bf16_t* W, const bf16_t* res, bf16_t* C, int M, int N, int K, int epi,
                        const bf16_t* gain, bf16_t* xn, float eps, bool fuse, hipStream_t s) {
    if (fuse && M > 16) {
        GemmArgs g{A, W, nullptr, res, C, M, N, K, K, K, N, N, epi, ConvGeom{0, 0, 0, 0, 0, 0}, nullptr, 0, 0};
        g.partial = w.splitk; g.partial_floats = w.splitk_floats;
        g.norm_w = gain; g.norm_out = xn; g.norm_ld = N; g.norm_eps = eps;
        const int st = launch_gemm(g, s);
        if (st != -95) return st;
    }
    int st = linear(A, W, nullptr, res, nullptr, C, M, N, K, K, K, N, N, 0.f, epi, s, nullptr, w.splitk, w.splitk_floats);
    if (st) return st;
    return launch_rmsnorm(C, gain, xn, M, N, N, N, eps, s);
}

// ---- Tensor-parallel prefill of a long prompt as TWO CONCURRENT LANES (SURVEY 8e / north_star: "all-reduce overlapped with the
// next GEMM"; replaces the serial schedule of the loop in emu_llama_forward).  The prompt's rows are cut once, at a multiple of 256
// (whole 256-row GEMM tiles and whole 64-key attention tiles below the cut), into A = [0, Ma) and B = [Ma, M).  Causal attention
// makes A independent of B, and B needs nothing of A but its keys / values of the SAME layer, so each half walks all layers as its
// own chain
//     [RMSNorm, qkv (+ RoPE / KV append / V^T), attention, o_proj] -> all-reduce -> [RMSNorm, gate/up + SwiGLU, down] -> all-reduce
// on its own stream: A on the caller's, B on cx->lane_stream, one event per layer (A's K / V rows are in the cache) the only edge
// between them.  While one lane's partial sums are on the wire the other lane's GEMMs have the CUs, and where both lanes compute,
// their launches (a TP = 8 shard's GEMM over 772 rows is 30-odd tiles on 256 CUs) share the chip instead of queueing behind each
// other.  A first version kept ONE compute stream and interleaved the halves stage by stage with every all-reduce on a side
// stream: +43 % per rank on one GPU (profiles/r05_tp_prefill_two_half_v1_single_compute_stream.log) -- half-size launches back to
// back leave most of the chip idle, and every all-reduce cost two cross-stream edges.  All-reduces run in their lane's stream
// where RCCL takes them (one communicator, calls in one host order on every rank); on a context without a communicator (rank
// processes sharing a device: validation) every peer-to-peer all-reduce additionally waits for the one issued before it in
// either lane, because the comm blocks serve one all-reduce at a time.  Lane B keeps its own V^T buffer (A's keys transposed from the cache + its own columns
// from the epilogue) and K-slice scratch, so no buffer is written by one lane while the other reads it.  Same kernels and
// rounding points as the serial schedule; a half may take another GEMM tile configuration than the whole prompt (K-slice sums in
// another order), so the schedules agree to bf16 rounding, not bit for bit.  Capturable: lane_stream forks from and joins the
// caller's stream through events.
int llama_prefill_lanes(emu_llama* m, const LlamaWs& w, bf16_t* hA, int M, const int32_t* pos, const int32_t* slot,
                        const int32_t* kstart, bool fuse_rope, hipStream_t s) {
    emu_ctx* cx = m->ctx;
    const emu_llama_cfg& c = m->cfg;
    const int H = c.hidden, Hl = c.heads_local, D = c.head_dim, HD = Hl * D, Fl = c.ffn_local;
    const int epi_res = cx->tp_rank == 0 ? EPI_RESID : EPI_NONE;        // the residual enters the all-reduce once
    const float scale = 1.0f / sqrtf((float)D);
    const size_t kv_layer = (size_t)Hl * m->s_max * D;                  // one batch element
    const int spad = (M + 63) / 64 * 64;
    const int l_end = m->l1 < 0 ? c.layers : m->l1;
    if (l_end <= m->l0) return 0;
    int Ma = (M / 2 + 128) / 256 * 256;
    if (Ma < 256) Ma = 256;
    if (Ma > M - 256) Ma = (M - 256) / 256 * 256;
    const int r0[2] = {0, Ma}, rows[2] = {Ma, M - Ma};
    hipStream_t lane[2] = {s, cx->lane_stream};
    bf16_t* vt[2] = {w.vt, w.vt2};
    float* sk[2] = {w.splitk, w.splitk2};
    hipEvent_t eStart = cx->ar_ev[0], eKV = cx->ar_ev[1];              // (ar_ev[2]: the join in llama_prefill_overlapped)
    hipEvent_t* ePrev = cx->ar_ev + 3;   // [lane]  the lane's latest peer-to-peer all-reduce is through (an event is only ever recorded on
    int prev_lane = -1;                  //         ONE stream: one event recorded on both streams of a capture crashed hipGraphInstantiate)
#define HIPTRY(expr) do { if ((expr) != hipSuccess) { (void)hipGetLastError(); return fail(cx, -5, #expr); } } while (0)
    auto lane_allreduce = [&](int h, bf16_t* buf, size_t n) -> int {
        const bool by_rccl = cx->comm && !(cx->p2p_on && n * sizeof(bf16_t) <= EMU_P2P_SLOT_BYTES);   // emu_allreduce_bf16's own choice
        if (by_rccl) return emu_allreduce_bf16(cx, buf, n, reinterpret_cast<emu_stream_t>(lane[h]));
        // the comm blocks serve ONE all-reduce at a time: every peer-to-peer all-reduce waits for the one issued before it (host order,
        // the same on every rank), whichever lane that was in
        if (prev_lane >= 0 && prev_lane != h) HIPTRY(hipStreamWaitEvent(lane[h], ePrev[prev_lane], 0));
        TRY(cx, emu_allreduce_bf16(cx, buf, n, reinterpret_cast<emu_stream_t>(lane[h])));
        HIPTRY(hipEventRecord(ePrev[h], lane[h]));
        prev_lane = h;
        return 0;
    };
    HIPTRY(hipEventRecord(eStart, s));                                   // inputs (and whatever the caller queued before) are ready
    HIPTRY(hipStreamWaitEvent(lane[1], eStart, 0));
    for (int l = m->l0; l < l_end; ++l) {
        const emu_llama::Layer& L = m->layers[l];
        if (!L.wqkv) return fail(cx, -22, "emu_llama_forward: layer weights not set");
        bf16_t* kc = m->kcache + l * kv_layer;
        bf16_t* vc = m->vcache + l * kv_layer;
        // Host order per layer: A.attention, B.attention, A.mlp, B.mlp -- so B's wait for eKV sees A's record of this layer, and the
        // all-reduces are issued in the order o_A, o_B, down_A, down_B.  The order matters beyond bookkeeping: all-reduces of one
        // communicator run in issue order (RCCL makes a call on another stream wait for the stream of the call before it; the
        // peer-to-peer chain below does the same), so an all-reduce can only wait for work issued ahead of it.  Issued lane by lane
        // (A's whole layer, then B's), B's first all-reduce would wait for A's second and the lanes would take turns instead of overlapping.
        for (int stage = 0; stage < 2; ++stage)
        for (int h = 0; h < 2; ++h) {
            hipStream_t ls = lane[h];
            const int Mh = rows[h], kend = r0[h] + Mh;
            bf16_t* x = hA + (size_t)r0[h] * H;
            bf16_t* xn = w.xn + (size_t)r0[h] * H;
            bf16_t* qkv = w.qkv + (size_t)r0[h] * 3 * HD;
            bf16_t* att = w.attn + (size_t)r0[h] * HD;
            bf16_t* act = w.act + (size_t)r0[h] * Fl;
            bf16_t* hB = w.hB + (size_t)r0[h] * H;
            if (stage == 1) {                                            // ---- SwiGLU MLP
                TRY(cx, launch_rmsnorm(hB, L.ln2, xn, Mh, H, H, H, c.rms_eps, ls));
                TRY(cx, linear(xn, L.wgu, nullptr, nullptr, nullptr, act, Mh, 2 * Fl, H, H, H, 0, Fl, 0.f, EPI_SWIGLU, ls, nullptr, sk[h], w.splitk_floats));
                TRY(cx, linear(act, L.wdown, nullptr, hB, nullptr, x, Mh, H, Fl, Fl, Fl, H, H, 0.f, epi_res, ls, nullptr, sk[h], w.splitk_floats));
                TRY(cx, lane_allreduce(h, x, (size_t)Mh * H));
                continue;
            }
            // ---- attention
            TRY(cx, launch_rmsnorm(x, L.ln1, xn, Mh, H, H, H, c.rms_eps, ls));
            int st = -95;
            if (fuse_rope) {
                if (h == 1) {
                    // lane B's V^T: A's keys of this layer out of the cache (zero behind them), its own columns from the epilogue below
                    HIPTRY(hipStreamWaitEvent(ls, eKV, 0));
                    TransposeVArgs tv{vc, (long)Hl * m->s_max * D, (long)m->s_max * D, (long)D, vt[1], 1, Hl, Ma, D, spad};
                    TRY(cx, launch_transpose_v(tv, ls));
                }
                GemmArgs g{xn, L.wqkv, nullptr, nullptr, qkv, Mh, 3 * HD, H, H, H, 0, 3 * HD, EPI_NONE, ConvGeom{0, 0, 0, 0, 0, 0}, nullptr, 0, 0};
                g.partial = sk[h]; g.partial_floats = w.splitk_floats;
                g.rope_cos = m->cos; g.rope_sin = m->sin; g.rope_pos = pos + r0[h]; g.rope_slot = slot + r0[h]; g.rope_kc = kc; g.rope_vc = vc;
                g.rope_hl = Hl; g.rope_smax = m->s_max;
                g.vt_out = vt[h] + r0[h]; g.vt_col0 = 2 * HD; g.vt_s = Mh; g.vt_spad = spad;   // key index = row index: column r0 + m
                st = launch_gemm(g, ls);
                if (st != 0 && st != -95) return fail(cx, st, "emu_llama_forward: qkv projection with the RoPE epilogue");
                if (st == -95) fuse_rope = false;                        // (before anything of this call was launched fused: lane A, first layer)
            }
            if (st == -95) {
                TRY(cx, linear(xn, L.wqkv, nullptr, nullptr, nullptr, qkv, Mh, 3 * HD, H, H, H, 0, 3 * HD, 0.f, EPI_NONE, ls, nullptr, sk[h], w.splitk_floats));
                RopeKvArgs r{qkv, m->cos, m->sin, pos + r0[h], slot + r0[h], kc, vc, 1, Mh, Hl, D, m->s_max};
                TRY(cx, launch_rope_kv(r, ls));
                if (h == 1) HIPTRY(hipStreamWaitEvent(ls, eKV, 0));
                TransposeVArgs tv{vc, (long)Hl * m->s_max * D, (long)m->s_max * D, (long)D, vt[h], 1, Hl, kend, D, spad};
                TRY(cx, launch_transpose_v(tv, ls));                     // keys [0, kend) key-contiguous, zero up to spad
            }
            if (h == 0) HIPTRY(hipEventRecord(eKV, ls));                 // K / V rows [0, Ma) of layer l are in the cache
            FlashArgs f{qkv, (long)Mh * 3 * HD, (long)D, (long)3 * HD,
                        kc, (long)Hl * m->s_max * D, (long)m->s_max * D, (long)D,
                        vt[h], att, (long)Mh * HD, (long)D, (long)HD, kstart,
                        1, Hl, Mh, kend, spad, D, 1, scale};                 // query i of the half sees keys <= r0 + i
            TRY(cx, launch_flash_attn(f, ls));
            TRY(cx, linear(att, L.wo, nullptr, x, nullptr, hB, Mh, H, HD, HD, HD, H, H, 0.f, epi_res, ls, nullptr, sk[h], w.splitk_floats));
            TRY(cx, lane_allreduce(h, hB, (size_t)Mh * H));
        }
    }
#undef HIPTRY
    return 0;
}
int llama_prefill_overlapped(emu_llama* m, const LlamaWs& w, bf16_t* hA, int M, const int32_t* pos, const int32_t* slot,
                             const int32_t* kstart, bool fuse_rope, hipStream_t s) {
    emu_ctx* cx = m->ctx;
    const int st = llama_prefill_lanes(m, w, hA, M, pos, slot, kstart, fuse_rope, s);
    // join, on the error path as well: whatever reached the second lane is ordered ahead of the caller's next launch (and a stream
    // capture in progress ends with the lane joined); the caller's stream owns the residual stream again
    const bool joined = hipEventRecord(cx->ar_ev[2], cx->lane_stream) == hipSuccess && hipStreamWaitEvent(s, cx->ar_ev[2], 0) == hipSuccess;
    if (st != 0) return st;
    if (!joined) { (void)hipGetLastError(); return fail(cx, -5, "emu_llama_forward: joining the second lane"); }
    ++m->ov_forwards;
    return 0;
}
}  // namespace

extern "C" {

int emu_llama_create(emu_ctx* ctx, const emu_llama_cfg* cfg, emu_llama** out) {
    if (!ctx || !cfg || !out) return -22;
    if ((cfg->head_dim != 128 && cfg->head_dim != 64) || (cfg->hidden & 7) || (cfg->ffn_local & 7) || cfg->layers < 1)
        return fail(ctx, -22, "emu_llama_create: head_dim must be 64/128, hidden and ffn_local multiples of 8");
    emu_llama* m = new emu_llama();
    m->ctx = ctx; m->cfg = *cfg;
    if (hipMalloc(reinterpret_cast<void**>(&m->arrive), EMU_ARRIVE_INTS * sizeof(int)) != hipSuccess ||
        hipMemset(m->arrive, 0, EMU_ARRIVE_INTS * sizeof(int)) != hipSuccess) {
        (void)hipGetLastError();
        if (m->arrive) (void)hipFree(m->arrive);
        m->arrive = nullptr;                           // the two-launch form needs none
    }
    m->layers.resize(cfg->layers);
    memset(m->layers.data(), 0, sizeof(emu_llama::Layer) * cfg->layers);
    *out = m;
    return 0;
}
void emu_llama_destroy(emu_llama* m) {
    if (m && m->arrive) (void)hipFree(m->arrive);
    if (m && m->dl_table) (void)hipFree(m->dl_table);
    if (m && m->dl_cnt) (void)hipFree(m->dl_cnt);
    if (m && m->dl_err) (void)hipFree(m->dl_err);
    if (m && m->eng_gran) (void)hipFree(m->eng_gran);
    delete m;
}

int emu_llama_set_decode_fused(emu_llama* m, int enable, int layers_per_launch) {
    if (!m || layers_per_launch < 0) return -22;
    if (enable && !m->dl_cnt) {
        const emu_llama_cfg& c = m->cfg;
        m->dl_cnt_bytes = decode_layers_cnt_ints(c.layers, c.heads_local) * sizeof(int);
        if (hipMalloc(reinterpret_cast<void**>(&m->dl_cnt), m->dl_cnt_bytes) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&m->dl_table), sizeof(DecodeLayerPtrs) * c.layers) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&m->dl_err), sizeof(unsigned)) != hipSuccess ||
            hipMemset(m->dl_err, 0, sizeof(unsigned)) != hipSuccess)
            return fail(m->ctx, -12, "emu_llama_set_decode_fused: device allocation");
        m->dl_dirty = true;
    }
    if (enable == 4 && !m->eng_gran) {
        const emu_llama_cfg& c = m->cfg;
        // per layer: the summed post-attention stream (hidden), the SwiGLU product (ffn_local), the summed layer output (hidden)
        m->eng_gran_bytes = (size_t)c.layers * (align_up((size_t)c.hidden * 4) * 2 + align_up((size_t)c.ffn_local * 4));
        if (hipMalloc(reinterpret_cast<void**>(&m->eng_gran), m->eng_gran_bytes) != hipSuccess)
            return fail(m->ctx, -12, "emu_llama_set_decode_fused: device allocation");
    }
    m->decode_fused = enable < 0 ? 0 : (enable > 4 ? 4 : enable);
    m->dl_per_launch = layers_per_launch;
    return 0;
}
int emu_llama_set_decode_trace(emu_llama* m, void* buf) {
    if (!m) return -22;
    m->dl_trace = reinterpret_cast<unsigned long long*>(buf);
    return 0;
}
int emu_llama_decode_fused_stats(emu_llama* m, unsigned int* giveups, long* forwards) {
    if (!m) return -22;
    if (giveups) {
        *giveups = 0;
        if (m->dl_err && hipMemcpy(giveups, m->dl_err, sizeof(unsigned), hipMemcpyDeviceToHost) != hipSuccess) return -5;
    }
    if (forwards) *forwards = m->dl_forwards;
    return 0;
}

int emu_llama_set_layer(emu_llama* m, int layer, const void* wqkv, const void* wo, const void* wgu, const void* wdown,
                        const void* ln1, const void* ln2) {
    if (!m || layer < 0 || layer >= m->cfg.layers) return -22;
    m->layers[layer] = {B(wqkv), B(wo), B(wgu), B(wdown), B(ln1), B(ln2)};
    m->dl_dirty = true;
    return 0;
}
int emu_llama_set_layer_fp8(emu_llama* m, int layer, const void* wqkv8, const float* sqkv, const void* wo8, const float* so,
                            const void* wgu8, const float* sgu, const void* wdown8, const float* sdown) {
    if (!m || layer < 0 || layer >= m->cfg.layers) return -22;
    if (!wqkv8 || !sqkv || !wo8 || !so || !wgu8 || !sgu || !wdown8 || !sdown) return -22;
    const emu_llama_cfg& c = m->cfg;
    if ((c.hidden & 15) || ((c.heads_local * c.head_dim) & 15) || (c.ffn_local & 15))
        return fail(m->ctx, -22, "emu_llama_set_layer_fp8: reduction widths must be multiples of 16");
    if (m->layers8.size() != (size_t)c.layers) m->layers8.assign(c.layers, emu_llama::Layer8{});
    auto U = [](const void* p) { return reinterpret_cast<const uint8_t*>(p); };
    m->layers8[layer] = {U(wqkv8), U(wo8), U(wgu8), U(wdown8), sqkv, so, sgu, sdown};
    return 0;
}
int emu_llama_set_head_fp8(emu_llama* m, const void* lm_head8, const float* lm_scale) {
    if (!m || !lm_head8 || !lm_scale) return -22;
    m->lm_head8 = reinterpret_cast<const uint8_t*>(lm_head8); m->lm_scale8 = lm_scale;
    return 0;
}
int emu_llama_set_decode_tail(emu_llama* m, int enable) {
    if (!m) return -22;
    m->decode_tail = enable != 0;
    return 0;
}
int emu_llama_set_prefill_fusion(emu_llama* m, int enable) {
    if (!m) return -22;
    m->prefill_fusion = enable != 0;
    m->fuse_norm_on = enable != 0;               // sticky part: the K-slice sum + RMSNorm fusion (independent of the slot order)
    return 0;
}
int emu_llama_set_tp_overlap(emu_llama* m, int min_rows) {
    if (!m || min_rows < 0) return -22;
    emu_ctx* cx = m->ctx;
    if (min_rows > 0 && !cx->lane_stream) {                              // created here, never inside a forward (stream capture)
        if (hipSetDevice(cx->device) != hipSuccess) return fail(cx, -19, "emu_llama_set_tp_overlap: hipSetDevice");
        if (hipStreamCreateWithFlags(&cx->lane_stream, hipStreamNonBlocking) != hipSuccess) {
            cx->lane_stream = nullptr;
            return fail(cx, -12, "emu_llama_set_tp_overlap: hipStreamCreateWithFlags");
        }
        for (hipEvent_t& e : cx->ar_ev)
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
                e = nullptr;
                return fail(cx, -12, "emu_llama_set_tp_overlap: hipEventCreateWithFlags");
            }
    }
    m->tp_overlap_rows = min_rows > 0 && min_rows < 512 ? 512 : min_rows;   // two halves of at least one 256-row tile each
    return 0;
}
long emu_llama_tp_overlap_count(const emu_llama* m) { return m ? m->ov_forwards : -1; }
int emu_llama_use_fp8(emu_llama* m, int enable) {
    if (!m) return -22;
    if (enable && m->layers8.size() != (size_t)m->cfg.layers)
        return fail(m->ctx, -22, "emu_llama_use_fp8: fp8 layer weights not set");
    m->fp8_decode = enable != 0;
    m->fp8_prefill = enable == 2;
    return 0;
}
int emu_llama_set_head(emu_llama* m, const void* final_norm, const void* lm_head, const void* embed, const void* rope_cos,
                       const void* rope_sin) {
    if (!m) return -22;
    m->final_norm = B(final_norm); m->lm_head = B(lm_head); m->embed = B(embed); m->cos = B(rope_cos); m->sin = B(rope_sin);
    return 0;
}
int emu_llama_set_head_shard(emu_llama* m, int row0, int rows) {
    if (!m) return -22;
    if (rows < 0) { m->head_row0 = 0; m->head_rows = -1; return 0; }
    if (row0 < 0 || rows < 1 || row0 + rows > m->cfg.vocab) return fail(m->ctx, -22, "emu_llama_set_head_shard: rows outside the vocabulary");
    m->head_row0 = row0; m->head_rows = rows;
    return 0;
}
int emu_llama_set_kv(emu_llama* m, void* kcache, void* vcache, int batch, int s_max) {
    if (!m) return -22;
    if (!kcache && !vcache) {                    // detach: the caller freed the caches; every later forward fails with -22 until new ones are set
        m->kcache = m->vcache = nullptr; m->kv_batch = m->s_max = 0;
        m->kv_share_nb = m->kv_share_len = 0;
        return 0;
    }
    if (!kcache || !vcache || batch < 1 || s_max < 1) return -22;
    m->kcache = B(kcache); m->vcache = B(vcache); m->kv_batch = batch; m->s_max = s_max;
    m->kv_share_nb = m->kv_share_len = 0;
    return 0;
}
int emu_llama_set_kv_share(emu_llama* m, int beams, int shared_slots) {
    if (!m) return -22;
    if (beams <= 1) { m->kv_share_nb = m->kv_share_len = 0; return 0; }
    if (beams > DECODE_SHARE_MAX || !m->kcache || m->kv_batch % beams || shared_slots < 0 || shared_slots > m->s_max)
        return fail(m->ctx, -22, "emu_llama_set_kv_share: 2..8 beams dividing the cache batch, shared slots within the capacity");
    m->kv_share_nb = beams; m->kv_share_len = shared_slots;
    return 0;
}
int emu_llama_set_layer_range(emu_llama* m, int l0, int l1) {
    if (!m || l0 < 0 || l1 > m->cfg.layers || (l1 >= 0 && l1 < l0)) return -22;
    m->l0 = l0; m->l1 = l1;
    return 0;
}
size_t emu_llama_workspace_bytes(const emu_llama* m, int Bn, int T) {
    if (!m) return 0;
    return llama_ws(m, Bn, T, nullptr).total;
}

int emu_llama_forward(emu_llama* m, void* hidden, int Bn, int T, const int32_t* pos, const int32_t* slot,
                      const int32_t* kstart, const int32_t* ctx_dev, int ctx, void* workspace, size_t ws_bytes,
                      emu_stream_t s_) {
    if (!m) return -22;
    // the slot-order promise (emu_llama_set_prefill_fusion) is per call: every T > 1 entry consumes it, whatever happens below -- an
    // early return must not leave it armed for an unrelated caller
    const bool promise = m->prefill_fusion;
    if (T > 1) m->prefill_fusion = false;
    if (!hidden || !pos || !slot) return -22;
    emu_ctx* cx = m->ctx;
    const emu_llama_cfg& c = m->cfg;
    if (!m->kcache || Bn != m->kv_batch) return fail(cx, -22, "emu_llama_forward: KV cache not set for this batch size");
    if (ctx < 1 || ctx > m->s_max) return fail(cx, -22, "emu_llama_forward: ctx out of range");
    if (!m->cos) return fail(cx, -22, "emu_llama_forward: rope tables not set");
    const LlamaWs w = llama_ws(m, Bn, T, workspace);
    if (w.total > ws_bytes) return fail(cx, -12, "emu_llama_forward: workspace too small");
    hipStream_t s = S(s_);
    const int M = Bn * T, H = c.hidden, Hl = c.heads_local, D = c.head_dim, HD = Hl * D, Fl = c.ffn_local;
    const bool tp = cx->tp_size > 1 || cx->comm != nullptr || cx->p2p_on;   // a 1-rank communicator / comm block still runs the all-reduces (tests, tools/tp_emulate.py)
    const int epi_res = (!tp || cx->tp_rank == 0) ? EPI_RESID : EPI_NONE;   // residual enters the all-reduce once
    const float scale = 1.0f / sqrtf((float)D);
    const size_t kv_layer = (size_t)Bn * Hl * m->s_max * D;
    const int spad = (ctx + 63) / 64 * 64;
    bf16_t* hA = B(hidden);
    const int l_end = m->l1 < 0 ? c.layers : m->l1;
    // Prefill of ONE batch element whose rows are the whole context in slot order (emu_llama_set_prefill_fusion: the caller
    // promises slot[i] = i): the qkv projection rotates q / k, appends k / v to the cache and writes V^T itself
    // (GemmArgs::rope_*), instead of the rope_kv and transpose_v launches.  The V^T buffer's pad columns [ctx, spad) are
    // never written on that path: zeroed once per call (the attention kernel multiplies them by masked probabilities).
    // The promise is per call: it is consumed here, so a later emu_llama_forward with rows in another slot order (any caller that did
    // not renew it) runs the unfused sequence, which honours slot[] everywhere.
    bool fuse_rope = promise && Bn == 1 && T > 16 && T == ctx && D == 128 && !(HD & 255) && !m->fp8_prefill && m->kv_share_nb <= 1;
    if (fuse_rope && hipMemsetAsync(w.vt, 0, (size_t)HD * spad * 2, s) != hipSuccess) return fail(cx, -5, "emu_llama_forward: hipMemsetAsync");
    // (prefill fusion, no tensor parallelism: the K-slice sums of o_proj / down_proj apply the RMSNorm that follows them; for
    // down_proj that is the NEXT layer's input norm, so a layer may find its normalised rows in w.xn already)
    // The K-slice sum + RMSNorm fusion does not depend on the slot order: it follows the sticky capability (fuse_norm_on: set with the
    // first promise, cleared by emu_llama_set_prefill_fusion(0)), so M > 16 rows of single-token steps (beam / contrastive search
    // with B * beams > 16) keep the fused path they had before the promise became one-shot.
    const bool fuse_norm = m->fuse_norm_on && !tp && M > 16 && !m->fp8_prefill;
    bool xn_ready = false;
    // ---- one-row step with bf16 weights on the persistent weight-streaming engine (decode_engine.hip; mode 4): per layer the attention
    // launches, then ONE launch for  o_proj -> all-reduce -> RMSNorm + gate/up (SwiGLU) -> down -> all-reduce -> RMSNorm + the NEXT
    // layer's qkv projection, the weight stream running ahead of the four hand-offs; same bits as the launches below.  Tensor-parallel
    // shards with rows of at most 13 KiB (TP >= 4) whose ranks have the device to themselves; anything else takes the launches.
    if (m->decode_fused == 4 && T == 1 && Bn == 1 && D == 128 && !m->fp8_decode && m->kv_share_nb <= 1 && l_end > m->l0 && m->eng_gran && m->dl_err &&
        tp && cx->p2p && cx->p2p_on && emu_p2p_fenced(cx->p2p) == 0 && H <= 6656 && HD <= 6656 && Fl <= 6656 && !(H & 7) && !(Fl & 7)) {
        EngArgs e{};
        unsigned int* eseq = nullptr;
        int en = 0, er = 0;
        int ncu = 0;
        if (emu_p2p_engine_view(cx->p2p, e.comm, &eseq, &en, &er) && hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, cx->device) == hipSuccess &&
            3 * HD >= 2 * ncu && H >= 2 * ncu && Fl >= ncu && ncu <= 1024) {
            e.tp_n = en; e.tp_rank = er; e.seq = eseq; e.err = m->dl_err; e.limit_ticks = 200000000LL; e.ncu = ncu;
            if (const char* ev = getenv("EMU_ENGINE_TIMEOUT_MS")) { const long ms = atol(ev); if (ms > 0) e.limit_ticks = ms * 100000LL; }
            if (const char* ev = getenv("EMU_ENGINE_LOADERS")) e.nload = atoi(ev);
            for (int l = m->l0; l < l_end; ++l)
                if (!m->layers[l].wqkv) return fail(cx, -22, "emu_llama_forward: layer weights not set");
            if (hipMemsetAsync(m->eng_gran, 0, m->eng_gran_bytes, s) != hipSuccess) return fail(cx, -5, "emu_llama_forward: hipMemsetAsync");
            const size_t gH = align_up((size_t)H * 4), gF = align_up((size_t)Fl * 4);
            const bool res_here = epi_res == EPI_RESID;           // rank 0 adds the residuals (they enter the all-reduce once)
            // the first layer's qkv projection is a launch of its own
            TRY(cx, linear(hA, m->layers[m->l0].wqkv, nullptr, nullptr, m->layers[m->l0].ln1, w.qkv, 1, 3 * HD, H, H, H, 0, 3 * HD, c.rms_eps, EPI_NONE, s));
            for (int l = m->l0; l < l_end; ++l) {
                const emu_llama::Layer& L = m->layers[l];
                bf16_t* kc = m->kcache + l * kv_layer;
                bf16_t* vc = m->vcache + l * kv_layer;
                DecodeFusedArgs da{w.qkv, m->cos, m->sin, pos, slot, kc, vc, w.attn, (long)HD, (long)D, kstart, w.dec, Bn, Hl, D, m->s_max, ctx, scale, 0, 0};
                TRY(cx, launch_decode_fused(da, s));
                const bool last = l + 1 == l_end;
                char* gl = reinterpret_cast<char*>(m->eng_gran) + (size_t)(l - 0) * (2 * gH + gF);
                uint32_t* g_hb = reinterpret_cast<uint32_t*>(gl);
                uint32_t* g_act = reinterpret_cast<uint32_t*>(gl + gH);
                uint32_t* g_ha = reinterpret_cast<uint32_t*>(gl + gH + gF);
                int k = 0;
                EngOp& o0 = e.op[k++];                          // o_proj: partial sums of the post-attention stream
                o0 = EngOp{};
                o0.W = L.wo; o0.N = H; o0.K = HD; o0.epi = res_here ? EPI_RESID : EPI_NONE; o0.res = hA; o0.res_src = 0;
                o0.vw = emu_gemv_partition(H, HD, false, epi_res); o0.x_src = 0; o0.xg = w.attn; o0.out_dst = 2; o0.ar_k = 0;
                EngOp& o1 = e.op[k++];                          // all-reduce -> RMSNorm -> gate / up -> SwiGLU
                o1 = EngOp{};
                o1.W = L.wgu; o1.N = 2 * Fl; o1.K = H; o1.gain = L.ln2; o1.eps = c.rms_eps; o1.epi = EPI_SWIGLU;
                o1.vw = emu_gemv_partition(2 * Fl, H, true, EPI_SWIGLU); o1.x_src = 2; o1.xgran = g_hb; o1.ar_k = 0; o1.keep_raw = 1;
                o1.out_dst = 1; o1.ogran = g_act;
                EngOp& o2 = e.op[k++];                          // down: partial sums of the layer output (+ the summed stream on rank 0)
                o2 = EngOp{};
                o2.W = L.wdown; o2.N = H; o2.K = Fl; o2.epi = res_here ? EPI_RESID : EPI_NONE; o2.res_src = 1;
                o2.vw = emu_gemv_partition(H, Fl, false, epi_res); o2.x_src = 1; o2.xgran = g_act;
                if (!last) {
                    o2.out_dst = 2; o2.ar_k = 1;
                    EngOp& o3 = e.op[k++];                      // all-reduce -> RMSNorm -> the NEXT layer's qkv projection
                    o3 = EngOp{};
                    const emu_llama::Layer& Ln = m->layers[l + 1];
                    o3.W = Ln.wqkv; o3.N = 3 * HD; o3.K = H; o3.gain = Ln.ln1; o3.eps = c.rms_eps; o3.epi = EPI_NONE;
                    o3.vw = emu_gemv_partition(3 * HD, H, true, EPI_NONE); o3.x_src = 2; o3.xgran = g_ha; o3.ar_k = 1; o3.sum_out = hA;
                    o3.out_dst = 0; o3.out = w.qkv;
                    e.n_ar = 2;
                } else {
                    o2.out_dst = 0; o2.out = hA;                // the last layer's output leaves as this rank's partial sums
                    e.n_ar = 1;
                }
                e.nops = k;
                const int st = launch_decode_engine(e, s);
                if (st) return fail(cx, st, "emu_llama_forward: decode engine launch");
                if (last) TRY(cx, emu_allreduce_bf16(cx, hA, (size_t)H, s_));
            }
            ++m->eng_forwards;
            ++m->dl_forwards;
            return 0;
        }
    }
    // ---- one-row step with bf16 weights: whole layers per launch (decode_layer.hip), same bits as the launches below
    if (m->decode_fused && m->decode_fused != 4 && T == 1 && Bn == 1 && D == 128 && !m->fp8_decode && m->kv_share_nb <= 1 && l_end > m->l0 && m->dl_cnt) {
        DecodeLayersArgs d{};
        d.table = m->dl_table; d.hA = hA; d.hB = w.hB; d.qkv = w.qkv; d.attn = w.attn; d.act = w.act; d.ws = w.dec;
        d.cos = m->cos; d.sin = m->sin; d.pos = pos; d.slot = slot; d.kstart = kstart;
        d.kcache = m->kcache; d.vcache = m->vcache; d.kv_layer = kv_layer;
        d.H = H; d.Hl = Hl; d.Fl = Fl; d.S_max = m->s_max; d.ctx_max = ctx;
        d.eps = c.rms_eps; d.scale = scale; d.epi_res = (!tp || cx->tp_rank == 0) ? 1 : 0;
        // wait bound: 2 s of wall clock (100 MHz ticks), or the peer-to-peer time-out where that is longer -- rank processes that SHARE
        // a GPU (validation runs) are time-sliced against each other, and a wave that is switched out keeps its start time: the 0.2 s
        // of the first version ran out under eight ranks on one device (garbage from step 4 on, give-ups counted)
        d.cnt = m->dl_cnt; d.err = m->dl_err; d.limit_ticks = 200000000LL;
        d.trace = m->dl_trace;
        bool ok = decode_layers_ok(d);
        // tensor parallelism: mode 2 runs the all-reduces inside the launch over the P2P comm blocks (every rank on its own GPU); mode 1
        // cuts every layer at its two all-reduces -- [q, attention, o_proj] | all-reduce | [gate/up, down] | all-reduce -- which also
        // serves RCCL and ranks that share a GPU (a launch that waits for a peer must not hold the CUs the peer needs)
        bool in_kernel_ar = false;
        if (ok && tp && cx->p2p) {
            long long lim = 0;
            const bool view = emu_p2p_view(cx->p2p, d.tp_block, &d.tp_seq, &d.tp_n, &d.tp_rank, &lim);
            in_kernel_ar = view && cx->p2p_on && m->decode_fused == 2 && emu_p2p_fenced(cx->p2p) == 0;   // the in-launch all-reduce is the fence-free form only
            if (!in_kernel_ar) d.tp_n = 0;
            if (view && lim > d.limit_ticks) d.limit_ticks = lim;  // a lagging peer holds every downstream wait: the peer bound applies
            ok = decode_layers_ok(d);
        }
        if (ok) {
            if (m->dl_dirty) {
                hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
                (void)hipStreamIsCapturing(s, &cs);
                if (cs != hipStreamCaptureStatusNone)
                    return fail(cx, -16, "emu_llama_forward: the fused decode path needs one eager step before a capture (weight table upload)");
                std::vector<DecodeLayerPtrs> t(c.layers);
                for (int l = 0; l < c.layers; ++l) {
                    const emu_llama::Layer& L = m->layers[l];
                    t[l] = {L.wqkv, L.wo, L.wgu, L.wdown, L.ln1, L.ln2};
                }
                if (hipStreamSynchronize(s) != hipSuccess ||        // fused launches still in flight on s read the table
                    hipMemcpy(m->dl_table, t.data(), sizeof(DecodeLayerPtrs) * c.layers, hipMemcpyHostToDevice) != hipSuccess)
                    return fail(cx, -5, "emu_llama_forward: weight table upload");
                m->dl_dirty = false;
            }
            for (int l = m->l0; l < l_end; ++l)
                if (!m->layers[l].wqkv) return fail(cx, -22, "emu_llama_forward: layer weights not set");
            if (hipMemsetAsync(m->dl_cnt, 0, m->dl_cnt_bytes, s) != hipSuccess) return fail(cx, -5, "emu_llama_forward: hipMemsetAsync");
            if (tp && m->decode_fused == 3 && cx->p2p_on && emu_p2p_fenced(cx->p2p) == 0 && emu_p2p_view(cx->p2p, d.tp_block, &d.tp_seq, &d.tp_n, &d.tp_rank, &d.limit_ticks)) {
                // mode 3: the weight streams with an RMSNorm in front stay stand-alone launches; the attention (split merge by the
                // head's last split) and the two row-sharded projections run as single-role launches whose LAST workgroup to arrive
                // runs the all-reduce -- nobody waits inside a launch except that one workgroup for its peers, so rank processes that
                // share a device cannot starve each other: 5 launches per layer instead of 8
                if (d.limit_ticks < 200000000LL) d.limit_ticks = 200000000LL;
                for (int l = m->l0; l < l_end; ++l) {
                    const emu_llama::Layer& L = m->layers[l];
                    d.layer0 = l; d.nlayers = 1;
                    TRY(cx, linear(hA, L.wqkv, nullptr, nullptr, L.ln1, w.qkv, 1, 3 * HD, H, H, H, 0, 3 * HD, c.rms_eps, EPI_NONE, s));
                    DecodeLayersArgs a1 = d; a1.tp_n = 0; a1.role0 = 1; a1.role1 = 2;
                    TRY(cx, launch_decode_layers(a1, s));
                    d.role0 = 2; d.role1 = 3;
                    TRY(cx, launch_decode_layers(d, s));
                    TRY(cx, linear(w.hB, L.wgu, nullptr, nullptr, L.ln2, w.act, 1, 2 * Fl, H, H, H, 0, Fl, c.rms_eps, EPI_SWIGLU, s));
                    d.role0 = 4; d.role1 = 5;
                    TRY(cx, launch_decode_layers(d, s));
                }
                ++m->dl_forwards;
                return 0;
            }
            if (tp && !in_kernel_ar) {
                d.tp_n = 0;
                for (int l = m->l0; l < l_end; ++l) {
                    d.layer0 = l; d.nlayers = 1;
                    d.role0 = 0; d.role1 = 3;
                    TRY(cx, launch_decode_layers(d, s));
                    TRY(cx, emu_allreduce_bf16(cx, w.hB, (size_t)H, s_));
                    d.role0 = 3; d.role1 = 5;
                    TRY(cx, launch_decode_layers(d, s));
                    TRY(cx, emu_allreduce_bf16(cx, hA, (size_t)H, s_));
                }
            } else {
                const int per = m->dl_per_launch > 0 ? m->dl_per_launch : l_end - m->l0;
                for (int l = m->l0; l < l_end; l += per) {
                    d.layer0 = l; d.nlayers = std::min(per, l_end - l);
                    TRY(cx, launch_decode_layers(d, s));
                }
            }
            ++m->dl_forwards;
            return 0;
        }
    }
    // ---- long prompt under tensor parallelism: two row halves, every all-reduce behind the other half's GEMMs (needs the slot-order
    // promise: the rows of the one batch element are the whole context in order, so the first half never reads the second's keys)
    if (tp && m->tp_overlap_rows > 0 && cx->lane_stream && cx->ar_ev[4] && w.vt2 && promise && Bn == 1 && T == ctx && M >= m->tp_overlap_rows && l_end > m->l0 &&
        !m->fp8_prefill && m->kv_share_nb <= 1)
        return llama_prefill_overlapped(m, w, hA, M, pos, slot, kstart, fuse_rope, s);
    for (int l = m->l0; l < l_end; ++l) {
        const emu_llama::Layer& L = m->layers[l];
        if (!L.wqkv) return fail(cx, -22, "emu_llama_forward: layer weights not set");
        bf16_t* kc = m->kcache + l * kv_layer;
        bf16_t* vc = m->vcache + l * kv_layer;
        // ---- attention
        bool merge_o = false;
        const bool f8 = m->fp8_decode && M <= 2;
        // prefill with the fp8 weight set: activations are quantised per row ahead of every GEMM, the block-scaled MFMA
        // runs at twice the bf16 rate (BASELINE configs[4]); needs whole 128-element k tiles
        const bool f8p = m->fp8_prefill && M > 16 && !(H & 127) && !(HD & 127) && !(Fl & 127) && m->layers8[l].wqkv;
        const emu_llama::Layer8 L8 = f8 ? m->layers8[l] : emu_llama::Layer8{};
        if (f8 && !L8.wqkv) return fail(cx, -22, "emu_llama_forward: fp8 decode enabled but fp8 layer weights not set");
        if (f8) {
            TRY(cx, linear(hA, B(L8.wqkv), nullptr, nullptr, L.ln1, w.qkv, M, 3 * HD, H, H, H, 0, 3 * HD, c.rms_eps, EPI_NONE, s, L8.sqkv));
        } else if (M == 1) {
            TRY(cx, linear(hA, L.wqkv, nullptr, nullptr, L.ln1, w.qkv, M, 3 * HD, H, H, H, 0, 3 * HD, c.rms_eps, EPI_NONE, s));
        } else {                                     // 2..16 rows: norm once, skinny MFMA stream; more: GEMM
            if (!xn_ready) TRY(cx, launch_rmsnorm(hA, L.ln1, w.xn, M, H, H, H, c.rms_eps, s));
            xn_ready = false;
            if (f8p) TRY(cx, linear_q8(w, w.xn, H, m->layers8[l].wqkv, m->layers8[l].sqkv, nullptr, w.qkv, M, 3 * HD, H, 0, 3 * HD, EPI_NONE, s));
            else {
                int st = -95;
                if (fuse_rope) {
                    GemmArgs g{w.xn, L.wqkv, nullptr, nullptr, w.qkv, M, 3 * HD, H, H, H, 0, 3 * HD, EPI_NONE, ConvGeom{0, 0, 0, 0, 0, 0}, nullptr, 0, 0};
                    g.partial = w.splitk; g.partial_floats = w.splitk_floats;
                    g.rope_cos = m->cos; g.rope_sin = m->sin; g.rope_pos = pos; g.rope_slot = slot; g.rope_kc = kc; g.rope_vc = vc;
                    g.rope_hl = Hl; g.rope_smax = m->s_max;
                    g.vt_out = w.vt; g.vt_col0 = 2 * HD; g.vt_s = M; g.vt_spad = spad;
                    st = launch_gemm(g, s);
                    if (st != 0 && st != -95) return fail(cx, st, "emu_llama_forward: qkv projection with the RoPE epilogue");
                    if (st == -95) fuse_rope = false;  // the 256x256 tile does not take this shape unsliced: unfused sequence
                }
                if (st == -95)
                    TRY(cx, linear(w.xn, L.wqkv, nullptr, nullptr, nullptr, w.qkv, M, 3 * HD, H, H, H, 0, 3 * HD, 0.f, EPI_NONE, s, nullptr, w.splitk, w.splitk_floats));
            }
        }
        if (T == 1) {
            // RoPE + KV append + attention in one launch; context = slot + 1 is read on the device (graph replay)
            DecodeFusedArgs a{w.qkv, m->cos, m->sin, pos, slot, kc, vc, w.attn, (long)HD, (long)D, kstart, w.dec,
                              Bn, Hl, D, m->s_max, ctx, scale, m->kv_share_nb, m->kv_share_len};
            if (m->decode_tail && m->arrive && m->kv_share_nb <= 1 && (long)Bn * Hl <= EMU_ARRIVE_INTS) a.arrive = m->arrive;
            // short shards (a TP = 8 rank's 7 heads), opt-in (emu_gemm_tune bit 19): the o_proj launch merges the attention's splits
            // itself, no combine launch (gemv_merge.hip).  Bit-identical and measured LEVEL with the two launches (3.15 vs 3.15-3.18 ms
            // per token of a TP = 8 shard, profiles/r05_tp_emulate_merged_o_proj.log): the merge is a dependent L2 trip inside the
            // projection, which is what the combine launch cost -- one launch less buys nothing here, like the in-kernel split merge
            // and the tail all-reduce before it
            merge_o = M == 1 && !f8 && !a.arrive && m->kv_share_nb <= 1 && gemv_merge_ok(Hl, D, H, (ctx + 127) / 128) &&
                      (emu_gemm_tune_get() & (1 << 19)) != 0;
            a.skip_combine = merge_o;
            TRY(cx, launch_decode_fused(a, s));
        } else {
            if (m->kv_share_nb > 1) return fail(cx, -22, "emu_llama_forward: shared-prefix KV rows serve single-token steps only");
            if (!fuse_rope) {
                { RopeKvArgs r{w.qkv, m->cos, m->sin, pos, slot, kc, vc, Bn, T, Hl, D, m->s_max};
                  TRY(cx, launch_rope_kv(r, s)); }
                TransposeVArgs tv{vc, (long)Hl * m->s_max * D, (long)m->s_max * D, (long)D, w.vt, Bn, Hl, ctx, D, spad};
                TRY(cx, launch_transpose_v(tv, s));
            }
            FlashArgs f{w.qkv, (long)T * 3 * HD, (long)D, (long)3 * HD,
                        kc, (long)Hl * m->s_max * D, (long)m->s_max * D, (long)D,
                        w.vt, w.attn, (long)T * HD, (long)D, (long)HD, kstart,
                        Bn, Hl, T, ctx, spad, D, 1, scale};
            TRY(cx, launch_flash_attn(f, s));
        }
        if (merge_o) {
            GemvMergeArgs g{w.dec, slot, (ctx + 127) / 128, Hl, L.wo, hA, w.hB, H, HD, HD, epi_res, nullptr};
            TRY(cx, gemv_merge_profiled(g, s));
        } else
        if (f8) TRY(cx, linear(w.attn, B(L8.wo), nullptr, hA, nullptr, w.hB, M, H, HD, HD, HD, H, H, 0.f, epi_res, s, L8.so));
        else if (f8p) TRY(cx, linear_q8(w, w.attn, HD, m->layers8[l].wo, m->layers8[l].so, hA, w.hB, M, H, HD, H, H, epi_res, s));
        else if (fuse_norm) TRY(cx, linear_then_rmsnorm(w, w.attn, L.wo, hA, w.hB, M, H, HD, epi_res, L.ln2, w.xn, c.rms_eps, true, s));
        else TRY(cx, linear(w.attn, L.wo, nullptr, hA, nullptr, w.hB, M, H, HD, HD, HD, H, H, 0.f, epi_res, s, nullptr, w.splitk, w.splitk_floats));
        if (tp) TRY(cx, emu_allreduce_bf16(cx, w.hB, (size_t)M * H, s_));
        // ---- SwiGLU MLP
        if (f8) {
            TRY(cx, linear(w.hB, B(L8.wgu), nullptr, nullptr, L.ln2, w.act, M, 2 * Fl, H, H, H, 0, Fl, c.rms_eps, EPI_SWIGLU, s, L8.sgu));
        } else if (M == 1) {
            TRY(cx, linear(w.hB, L.wgu, nullptr, nullptr, L.ln2, w.act, M, 2 * Fl, H, H, H, 0, Fl, c.rms_eps, EPI_SWIGLU, s));
        } else {
            if (!fuse_norm) TRY(cx, launch_rmsnorm(w.hB, L.ln2, w.xn, M, H, H, H, c.rms_eps, s));
            if (f8p) TRY(cx, linear_q8(w, w.xn, H, m->layers8[l].wgu, m->layers8[l].sgu, nullptr, w.act, M, 2 * Fl, H, 0, Fl, EPI_SWIGLU, s));
            else TRY(cx, linear(w.xn, L.wgu, nullptr, nullptr, nullptr, w.act, M, 2 * Fl, H, H, H, 0, Fl, 0.f, EPI_SWIGLU, s, nullptr, w.splitk, w.splitk_floats));
        }
        if (f8) TRY(cx, linear(w.act, B(L8.wdown), nullptr, w.hB, nullptr, hA, M, H, Fl, Fl, Fl, H, H, 0.f, epi_res, s, L8.sdown));
        else if (f8p) TRY(cx, linear_q8(w, w.act, Fl, m->layers8[l].wdown, m->layers8[l].sdown, w.hB, hA, M, H, Fl, H, H, epi_res, s));
        else if (fuse_norm && l + 1 < l_end && m->layers[l + 1].ln1) {
            TRY(cx, linear_then_rmsnorm(w, w.act, L.wdown, w.hB, hA, M, H, Fl, epi_res, m->layers[l + 1].ln1, w.xn, c.rms_eps, true, s));
            xn_ready = true;
        }
        else TRY(cx, linear(w.act, L.wdown, nullptr, w.hB, nullptr, hA, M, H, Fl, Fl, Fl, H, H, 0.f, epi_res, s, nullptr, w.splitk, w.splitk_floats));
        if (tp) TRY(cx, emu_allreduce_bf16(cx, hA, (size_t)M * H, s_));
    }
    return 0;
}

size_t emu_beam_step_workspace_bytes(int Bn, int nb, int V) { return beam_step_ws_floats(Bn, nb, V) * sizeof(float); }
int emu_beam_step_bf16(const void* logits, long ld_prompt, long ld_beam, int V, int Bn, int nb, int L, int cur, const int32_t* cur_dev,
                       int min_len, int eos_id, float length_penalty, int hf431, int32_t* running_seq, int32_t* sequences, float* running_scores, float* beam_scores,
                       unsigned char* finished, int32_t* seq_len, unsigned char* heuristic_open, int32_t* next_tok, long* beam_flat,
                       void* workspace, size_t ws_bytes, emu_stream_t s) {
    if (!logits || !running_seq || !sequences || !running_scores || !beam_scores || !finished || !seq_len || !heuristic_open ||
        !next_tok || !beam_flat)
        return -22;
    BeamStepArgs a{B(logits), ld_prompt, ld_beam, V, Bn, nb, L, cur, cur_dev, min_len, eos_id, length_penalty, hf431, running_seq, sequences,
                   running_scores, beam_scores, finished, seq_len, heuristic_open, next_tok, beam_flat};
    return launch_beam_step(a, reinterpret_cast<float*>(workspace), ws_bytes / sizeof(float), S(s));
}

int emu_regress_advance_bf16(const void* src, void* out_all, void* prev, int32_t* pos, int32_t* slot, int32_t* step_dev, int Bn,
                             int cols, emu_stream_t s) {
    if (!src || !out_all || !prev || !pos || !slot || !step_dev) return -22;
    return launch_regress_advance(B(src), B(out_all), B(prev), pos, slot, step_dev, Bn, cols, S(s));
}
int emu_beam_advance(int32_t* cur_dev, int32_t* pos, int32_t* slot, const int32_t* pos0, int slot0, int rows, int L, int phase,
                     emu_stream_t s) {
    return launch_beam_advance(cur_dev, pos, slot, pos0, slot0, rows, L, phase, S(s));
}
int emu_llama_beam_reorder_kv(emu_llama* m, const long* beam_flat, const int32_t* cur_dev, int beams, int slot0, int L,
                              emu_stream_t s) {
    if (!m || !m->kcache) return -22;
    return launch_beam_reorder(m->kcache, m->vcache, beam_flat, cur_dev, m->cfg.layers, m->kv_batch, m->cfg.heads_local, m->s_max,
                               m->cfg.head_dim, beams, slot0, L, S(s));
}

int emu_llama_final_norm(emu_llama* m, const void* hidden, void* out, int rows, emu_stream_t s) {
    if (!m || !m->final_norm) return -22;
    return launch_rmsnorm(B(hidden), m->final_norm, B(out), rows, m->cfg.hidden, m->cfg.hidden, m->cfg.hidden,
                          m->cfg.rms_eps, S(s));
}

int emu_llama_logits(emu_llama* m, const void* hidden, int ldh, int M, void* logits, int ld, void* workspace,
                     size_t ws_bytes, emu_stream_t s) {
    if (!m || !m->lm_head || !m->final_norm) return -22;
    const emu_llama_cfg& c = m->cfg;
    if (m->head_rows >= 0) {
        // vocabulary-sharded head (tensor parallelism): this rank streams its rows of lm_head only (4035 of 32 274 at TP = 8:
        // 54 MB instead of 430 MB per token) into its own columns of the caller's [M, vocab] rows, every other column zero, and one
        // all-reduce makes the rows whole on every rank -- each logit is computed by exactly one rank and summed with zeros, so the
        // result is bit-identical to the replicated head and every consumer (arg-max, beam scorer, samplers) stays as it is.
        emu_ctx* cx = m->ctx;
        bf16_t* lg = B(logits);
        const int N = m->head_rows, n0 = m->head_row0;
        if (hipMemsetAsync(lg, 0, ((size_t)(M - 1) * ld + c.vocab) * sizeof(bf16_t), S(s)) != hipSuccess)
            return fail(cx, -5, "emu_llama_logits: hipMemsetAsync");
        if (m->fp8_decode && M <= 2 && m->lm_head8) {             // the e4m3 copy covers the same rows (quantised from the shard)
            TRY(cx, linear(B(hidden), B(m->lm_head8), nullptr, nullptr, m->final_norm, lg + n0, M, N, c.hidden, ldh, c.hidden, 0, ld,
                           c.rms_eps, EPI_NONE, S(s), m->lm_scale8));
        } else if (M == 1 || (M <= 8 && ws_bytes < (size_t)M * c.hidden * 2)) {
            TRY(cx, linear(B(hidden), m->lm_head, nullptr, nullptr, m->final_norm, lg + n0, M, N, c.hidden, ldh, c.hidden, 0, ld,
                           c.rms_eps, EPI_NONE, S(s)));
        } else {
            if (ws_bytes < (size_t)M * c.hidden * 2) return fail(cx, -12, "emu_llama_logits: workspace too small");
            TRY(cx, launch_rmsnorm(B(hidden), m->final_norm, B(workspace), M, c.hidden, ldh, c.hidden, c.rms_eps, S(s)));
            TRY(cx, linear(B(workspace), m->lm_head, nullptr, nullptr, nullptr, lg + n0, M, N, c.hidden, c.hidden, c.hidden, 0, ld, 0.f,
                           EPI_NONE, S(s)));
        }
        if (ld == c.vocab) return emu_allreduce_bf16(cx, lg, (size_t)M * c.vocab, s);
        for (int r = 0; r < M; ++r) TRY(cx, emu_allreduce_bf16(cx, lg + (size_t)r * ld, (size_t)c.vocab, s));
        return 0;
    }
    if (m->fp8_decode && M <= 2 && m->lm_head8)
        return linear(B(hidden), B(m->lm_head8), nullptr, nullptr, m->final_norm, B(logits), M, c.vocab, c.hidden, ldh,
                      c.hidden, 0, ld, c.rms_eps, EPI_NONE, S(s), m->lm_scale8);
    if (M == 1 || (M <= 8 && ws_bytes < (size_t)M * c.hidden * 2))
        return linear(B(hidden), m->lm_head, nullptr, nullptr, m->final_norm, B(logits), M, c.vocab, c.hidden, ldh,
                      c.hidden, 0, ld, c.rms_eps, EPI_NONE, S(s));
    if (ws_bytes < (size_t)M * c.hidden * 2) return fail(m->ctx, -12, "emu_llama_logits: workspace too small");
    TRY(m->ctx, launch_rmsnorm(B(hidden), m->final_norm, B(workspace), M, c.hidden, ldh, c.hidden, c.rms_eps, S(s)));
    return linear(B(workspace), m->lm_head, nullptr, nullptr, nullptr, B(logits), M, c.vocab, c.hidden, c.hidden, c.hidden,
                  0, ld, 0.f, EPI_NONE, S(s));
}

int emu_llama_greedy_step(emu_llama* m, int Bn, int32_t* cur_ids, int32_t* pos, int32_t* slot, const int32_t* kstart,
                          int32_t* ctx_dev, int32_t* step_dev, int32_t* out_ids, int ctx_upper, void* hidden,
                          void* logits, int ld_logits, void* workspace, size_t ws_bytes, emu_stream_t s) {
    if (!m || !m->embed) return -22;
    const emu_llama_cfg& c = m->cfg;
    TRY(m->ctx, launch_embed_gather(cur_ids, m->embed, B(hidden), Bn, c.hidden, c.vocab, S(s)));
    TRY(m->ctx, emu_llama_forward(m, hidden, Bn, 1, pos, slot, kstart, ctx_dev, ctx_upper, workspace, ws_bytes, s));
    TRY(m->ctx, emu_llama_logits(m, hidden, c.hidden, Bn, logits, ld_logits, workspace, ws_bytes, s));
    TRY(m->ctx, launch_argmax(B(logits), ld_logits, Bn, c.vocab, -1, cur_ids, S(s)));
    TRY(m->ctx, launch_greedy_advance(cur_ids, pos, slot, ctx_dev, step_dev, out_ids, Bn, S(s)));
    return 0;
}

}  // extern "C"

// =============================================================================================== ViT engine
struct emu_vit {
    emu_ctx* ctx;
    emu_vit_cfg cfg;
    const bf16_t *wpatch = nullptr, *bpatch = nullptr, *cls = nullptr, *pos = nullptr;
    struct Block { const bf16_t *wqkv, *bqkv, *wproj, *bproj, *ln1w, *ln1b, *fc1w, *fc1b, *fc2w, *fc2b, *ln2w, *ln2b; };
    std::vector<Block> blocks;
    // optional fp8 (e4m3, one fp32 scale per output row) copies of the four matrices of every block: emu_vit_use_fp8 runs the
    // blocks' GEMMs W8A8 on the block-scaled MFMA, the activation rows quantised per row ahead of every GEMM
    struct Block8 { const uint8_t *wqkv = nullptr, *wproj = nullptr, *fc1w = nullptr, *fc2w = nullptr;
                    const float *sqkv = nullptr, *sproj = nullptr, *sfc1 = nullptr, *sfc2 = nullptr; };
    std::vector<Block8> blocks8;
    bool fp8 = false;
    bool fuse_vt = true;                             // emu_vit_set_fusion bit 0: V^T out of the qkv projection's epilogue (one image)
    bool fuse_norm = true;                           // bit 1: fc2's K-slice sum applies bias + LayerNorm + residual (post-norm blocks)
};

namespace {
constexpr int VIT_DP = 128;      // padded head dim
struct VitWs { bf16_t *patches, *pemb, *qkv, *vt, *attn, *tmp, *h1; float* splitk; size_t splitk_floats;
               uint8_t* x8; float* xs;           // fp8 mode: the current GEMM's activation rows as e4m3 bytes + per-row scales
               size_t total; };
VitWs vit_ws(const emu_vit* m, int Bn, void* base) {
    const emu_vit_cfg& c = m->cfg;
    const int g = c.image_size / c.patch_size, T = g * g, N = T + 1;
    const size_t M = (size_t)Bn * N;
    const size_t npad = (size_t)(N + 63) / 64 * 64;
    char* p = reinterpret_cast<char*>(base);
    size_t off = 0;
    auto take = [&](size_t bytes) { char* r = p ? p + off : nullptr; off += align_up(bytes); return r; };
    VitWs w;
    w.patches = (bf16_t*)take((size_t)Bn * T * c.kpad * 2);
    w.pemb = (bf16_t*)take((size_t)Bn * T * c.width * 2);
    w.qkv = (bf16_t*)take(M * 3 * c.heads * VIT_DP * 2);
    w.vt = (bf16_t*)take((size_t)Bn * c.heads * VIT_DP * npad * 2);
    w.attn = (bf16_t*)take(M * c.heads * VIT_DP * 2);
    w.tmp = (bf16_t*)take(M * c.width * 2);
    w.h1 = (bf16_t*)take(M * (size_t)c.mlp_hidden * 2);
    w.splitk_floats = EMU_SPLITK_SCRATCH_FLOATS;          // K-slices of fc2 (63 tiles of 256x128) and of fc1's tail round
    w.splitk = (float*)take(w.splitk_floats * sizeof(float));
    const size_t kmax = std::max<size_t>(std::max<size_t>(c.width, (size_t)c.heads * VIT_DP), c.mlp_hidden);
    w.x8 = (uint8_t*)take(m->blocks8.empty() ? 0 : M * kmax);
    w.xs = (float*)take(m->blocks8.empty() ? 0 : M * sizeof(float));
    w.total = off;
    return w;
}
}  // namespace

extern "C" {

int emu_vit_create(emu_ctx* ctx, const emu_vit_cfg* cfg, emu_vit** out) {
    if (!ctx || !cfg || !out) return -22;
    if (cfg->head_width > VIT_DP || (cfg->width & 7) || (cfg->mlp_hidden & 7) || (cfg->kpad & 7) ||
        cfg->kpad < 3 * cfg->patch_size * cfg->patch_size || cfg->image_size % cfg->patch_size)
        return fail(ctx, -22, "emu_vit_create: unsupported shape");
    emu_vit* m = new emu_vit();
    m->ctx = ctx; m->cfg = *cfg;
    m->blocks.resize(cfg->layers);
    memset(m->blocks.data(), 0, sizeof(emu_vit::Block) * cfg->layers);
    *out = m;
    return 0;
}
void emu_vit_destroy(emu_vit* m) { delete m; }
int emu_vit_set_stem(emu_vit* m, const void* wpatch, const void* bpatch, const void* cls, const void* pos) {
    if (!m) return -22;
    m->wpatch = B(wpatch); m->bpatch = B(bpatch); m->cls = B(cls); m->pos = B(pos);
    return 0;
}
int emu_vit_set_block(emu_vit* m, int layer, const void* wqkv, const void* bqkv, const void* wproj, const void* bproj,
                      const void* ln1w, const void* ln1b, const void* fc1w, const void* fc1b, const void* fc2w,
                      const void* fc2b, const void* ln2w, const void* ln2b) {
    if (!m || layer < 0 || layer >= m->cfg.layers) return -22;
    m->blocks[layer] = {B(wqkv), B(bqkv), B(wproj), B(bproj), B(ln1w), B(ln1b), B(fc1w), B(fc1b), B(fc2w), B(fc2b), B(ln2w), B(ln2b)};
    return 0;
}
int emu_vit_set_block_fp8(emu_vit* m, int layer, const void* wqkv8, const float* sqkv, const void* wproj8, const float* sproj,
                          const void* fc1w8, const float* sfc1, const void* fc2w8, const float* sfc2) {
    if (!m || layer < 0 || layer >= m->cfg.layers) return -22;
    if (!wqkv8 || !sqkv || !wproj8 || !sproj || !fc1w8 || !sfc1 || !fc2w8 || !sfc2) return -22;
    const emu_vit_cfg& c = m->cfg;
    if ((c.width & 127) || (c.mlp_hidden & 127))
        return fail(m->ctx, -22, "emu_vit_set_block_fp8: width and mlp_hidden must be multiples of 128 (k tiles of the fp8 MFMA)");
    if (m->blocks8.size() != (size_t)c.layers) m->blocks8.assign(c.layers, emu_vit::Block8{});
    auto U = [](const void* p) { return reinterpret_cast<const uint8_t*>(p); };
    m->blocks8[layer] = {U(wqkv8), U(wproj8), U(fc1w8), U(fc2w8), sqkv, sproj, sfc1, sfc2};
    return 0;
}
int emu_vit_set_fusion(emu_vit* m, int mask) {
    if (!m) return -22;
    m->fuse_vt = (mask & 1) != 0;
    m->fuse_norm = (mask & 2) != 0;
    return 0;
}
int emu_vit_use_fp8(emu_vit* m, int enable) {
    if (!m) return -22;
    if (enable) {
        if (m->blocks8.size() != (size_t)m->cfg.layers) return fail(m->ctx, -22, "emu_vit_use_fp8: fp8 block weights not set");
        for (const auto& b : m->blocks8)
            if (!b.wqkv) return fail(m->ctx, -22, "emu_vit_use_fp8: fp8 block weights not set");
    }
    m->fp8 = enable != 0;
    return 0;
}
size_t emu_vit_workspace_bytes(const emu_vit* m, int Bn) { return m ? vit_ws(m, Bn, nullptr).total : 0; }

// blocks [l0, l1) in place on tokens x [B * N, C]
static int vit_blocks(emu_vit* m, bf16_t* x, int Bn, int l0, int l1, const VitWs& w, hipStream_t s) {
    emu_ctx* cx = m->ctx;
    const emu_vit_cfg& c = m->cfg;
    const int g = c.image_size / c.patch_size, T = g * g, N = T + 1, M = Bn * N, C = c.width, Hh = c.heads;
    const int QK = Hh * VIT_DP, F = c.mlp_hidden;
    const int npad = (N + 63) / 64 * 64;
    const float scale = 1.0f / sqrtf((float)c.head_width);
    // one GEMM of a block: bf16, or (emu_vit_use_fp8) the activation rows quantised per row + the fp8 x fp8 GEMM
    // (quantised = true: the LayerNorm in front of this GEMM has left the rows in w.x8 / w.xs already: launch_layernorm_q8)
    auto lin = [&](const bf16_t* A, const bf16_t* W, const uint8_t* W8, const float* ws8, const bf16_t* bias, const bf16_t* res,
                   bf16_t* Cc, int N_, int K_, int ldres, int epi, bool quantised = false) -> int {
        if (!W8)
            return linear(A, W, bias, res, nullptr, Cc, M, N_, K_, K_, K_, ldres, N_, 0.f, epi, s, nullptr, w.splitk, w.splitk_floats);
        if (!quantised) {
            const int st = launch_quant_fp8_rows(A, K_, w.x8, K_, w.xs, M, K_, s);
            if (st) return st;
        }
        GemmArgs ga{reinterpret_cast<const bf16_t*>(w.x8), reinterpret_cast<const bf16_t*>(W8), bias, res, Cc, M, N_, K_, K_, K_, ldres, N_,
                    epi, ConvGeom{0, 0, 0, 0, 0, 0}, nullptr, 0, 0};
        ga.a_scale = w.xs; ga.w_scale = ws8;
        ga.partial = w.splitk; ga.partial_floats = w.splitk_floats;
        return launch_gemm_fp8(ga, s);
    };
    // V^T out of the qkv epilogue (one image, bf16) writes keys [0, N) and the few pad keys of the last 8-key group; the rest of the
    // pad columns [N, npad) of the V^T buffer is nobody's: zeroed once per call (the attention kernel multiplies them by masked
    // probabilities, and a stale NaN pattern times zero is NaN)
    if (m->fuse_vt && Bn == 1 && !m->fp8 && npad != N && l1 > l0 &&
        hipMemsetAsync(w.vt, 0, (size_t)Hh * VIT_DP * npad * 2, s) != hipSuccess)
        return fail(cx, -5, "emu_vit_forward: hipMemsetAsync");
    const bool q8 = m->fp8 && C <= 2048;                 // LayerNorm rows leave as fp8 operands (launch_layernorm_q8)
    bool x8_valid = false;                               // post-norm: w.x8 / w.xs hold the quantised rows of x
    for (int l = l0; l < l1; ++l) {
        const emu_vit::Block& Bk = m->blocks[l];
        if (!Bk.wqkv) return fail(cx, -22, "emu_vit_forward: block weights not set");
        const emu_vit::Block8 B8 = m->fp8 ? m->blocks8[l] : emu_vit::Block8{};
        const bf16_t* ain = x;                           // attention input
        if (c.prenorm) {
            if (q8) TRY(cx, launch_layernorm_q8(x, Bk.ln1w, Bk.ln1b, nullptr, nullptr, w.x8, w.xs, M, C, c.ln_eps, s));
            else TRY(cx, launch_layernorm(x, Bk.ln1w, Bk.ln1b, nullptr, w.tmp, M, C, c.ln_eps, s));
            ain = w.tmp;
        }
        // one image, bf16: the V heads leave the qkv projection key-contiguous (GemmArgs::vt_*, round 3's UNet epilogue; the
        // transposed staging takes a single batch element's ragged last tile since round 4) -- no transpose launch
        const bool fvt = m->fuse_vt && Bn == 1 && !B8.wqkv;
        if (fvt) {
            GemmArgs gq{ain, Bk.wqkv, Bk.bqkv, nullptr, w.qkv, M, 3 * QK, C, C, C, 0, 3 * QK, EPI_NONE, ConvGeom{0, 0, 0, 0, 0, 0}, nullptr, 0, 0};
            gq.partial = w.splitk; gq.partial_floats = w.splitk_floats;
            gq.vt_out = w.vt; gq.vt_col0 = 2 * QK; gq.vt_s = N; gq.vt_spad = npad;
            TRY(cx, launch_gemm(gq, s));
        } else {
            TRY(cx, lin(ain, Bk.wqkv, B8.wqkv, B8.sqkv, Bk.bqkv, nullptr, w.qkv, 3 * QK, C, 0, EPI_NONE, q8 && (c.prenorm || x8_valid)));
            TransposeVArgs tv{w.qkv + 2 * QK, (long)N * 3 * QK, (long)VIT_DP, (long)3 * QK, w.vt, Bn, Hh, N, VIT_DP, npad};
            TRY(cx, launch_transpose_v(tv, s));
        }
        FlashArgs f{w.qkv, (long)N * 3 * QK, (long)VIT_DP, (long)3 * QK,
                    w.qkv + QK, (long)N * 3 * QK, (long)VIT_DP, (long)3 * QK,
                    w.vt, w.attn, (long)N * QK, (long)VIT_DP, (long)QK, nullptr,
                    Bn, Hh, N, N, npad, VIT_DP, 0, scale};
        TRY(cx, launch_flash_attn(f, s));
        if (c.prenorm) {
            // x = x + proj(attn);  x = x + fc2(gelu(fc1(LN2(x))))
            TRY(cx, lin(w.attn, Bk.wproj, B8.wproj, B8.sproj, Bk.bproj, x, x, C, QK, C, EPI_RESID));
            if (q8) TRY(cx, launch_layernorm_q8(x, Bk.ln2w, Bk.ln2b, nullptr, nullptr, w.x8, w.xs, M, C, c.ln_eps, s));
            else TRY(cx, launch_layernorm(x, Bk.ln2w, Bk.ln2b, nullptr, w.tmp, M, C, c.ln_eps, s));
            TRY(cx, lin(w.tmp, Bk.fc1w, B8.fc1w, B8.sfc1, Bk.fc1b, nullptr, w.h1, F, C, 0, EPI_GELU, q8));
            TRY(cx, lin(w.h1, Bk.fc2w, B8.fc2w, B8.sfc2, Bk.fc2b, x, x, C, F, C, EPI_RESID));
        } else {
            TRY(cx, lin(w.attn, Bk.wproj, B8.wproj, B8.sproj, Bk.bproj, nullptr, w.tmp, C, QK, 0, EPI_NONE));
            if (q8) TRY(cx, launch_layernorm_q8(w.tmp, Bk.ln1w, Bk.ln1b, x, x, w.x8, w.xs, M, C, c.ln_eps, s));
            else TRY(cx, launch_layernorm(w.tmp, Bk.ln1w, Bk.ln1b, x, x, M, C, c.ln_eps, s));
            TRY(cx, lin(x, Bk.fc1w, B8.fc1w, B8.sfc1, Bk.fc1b, nullptr, w.h1, F, C, 0, EPI_GELU, q8));
            // bf16: fc2 is a K-sliced GEMM (28 big tiles, K = 15360): its slice sum applies bias, LayerNorm and the residual add
            // row-wise in the same launch (GemmArgs::norm_*; -95 = this shape is not sliced that way: the launches apart)
            int st2 = -95;
            if (m->fuse_norm && !B8.fc2w) {
                GemmArgs gf{w.h1, Bk.fc2w, Bk.fc2b, nullptr, nullptr, M, C, F, F, F, 0, C, EPI_NONE, ConvGeom{0, 0, 0, 0, 0, 0}, nullptr, 0, 0};
                gf.partial = w.splitk; gf.partial_floats = w.splitk_floats;
                gf.norm_w = Bk.ln2w; gf.norm_b = Bk.ln2b; gf.norm_res = x; gf.norm_ldres = C; gf.norm_out = x; gf.norm_ld = C; gf.norm_eps = c.ln_eps;
                st2 = launch_gemm(gf, s);
                if (st2 != 0 && st2 != -95) return fail(cx, st2, "emu_vit_forward: fc2 with the LayerNorm slice sum");
            }
            if (st2 == -95) {
                TRY(cx, lin(w.h1, Bk.fc2w, B8.fc2w, B8.sfc2, Bk.fc2b, nullptr, w.tmp, C, F, 0, EPI_NONE));
                if (q8) TRY(cx, launch_layernorm_q8(w.tmp, Bk.ln2w, Bk.ln2b, x, x, w.x8, w.xs, M, C, c.ln_eps, s));   // the next block's qkv rows
                else TRY(cx, launch_layernorm(w.tmp, Bk.ln2w, Bk.ln2b, x, x, M, C, c.ln_eps, s));
            }
            x8_valid = q8;
        }
    }
    return 0;
}

int emu_vit_forward(emu_vit* m, const void* image, int image_is_f32, int Bn, void* out_tokens, void* workspace,
                    size_t ws_bytes, emu_stream_t s_) {
    if (!m || !image || !out_tokens || !m->wpatch) return -22;
    emu_ctx* cx = m->ctx;
    const emu_vit_cfg& c = m->cfg;
    const VitWs w = vit_ws(m, Bn, workspace);
    if (w.total > ws_bytes) return fail(cx, -12, "emu_vit_forward: workspace too small");
    hipStream_t s = S(s_);
    const int g = c.image_size / c.patch_size, T = g * g, C = c.width;
    bf16_t* x = B(out_tokens);
    TRY(cx, launch_patchify(image, image_is_f32, w.patches, Bn, 3, c.image_size, c.patch_size, c.kpad, s));
    TRY(cx, linear(w.patches, m->wpatch, m->bpatch, nullptr, nullptr, w.pemb, Bn * T, C, c.kpad, c.kpad, c.kpad, 0, C, 0.f, EPI_NONE, s));
    TRY(cx, launch_vit_assemble(w.pemb, m->cls, m->pos, x, Bn, T, C, s));
    return vit_blocks(m, x, Bn, 0, c.layers, w, s);
}

int emu_vit_blocks(emu_vit* m, void* tokens, int Bn, int l0, int l1, void* workspace, size_t ws_bytes, emu_stream_t s_) {
    if (!m || !tokens || l0 < 0 || l1 > m->cfg.layers || l1 < l0) return -22;
    const VitWs w = vit_ws(m, Bn, workspace);
    if (w.total > ws_bytes) return fail(m->ctx, -12, "emu_vit_blocks: workspace too small");
    return vit_blocks(m, B(tokens), Bn, l0, l1, w, S(s_));
}

}  // extern "C"
