// Internal launch interface between the C-ABI layer (capi.cpp / engine.cpp) and the HIP kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;

enum EmuEpilogue { EPI_NONE = 0, EPI_RESID = 1, EPI_SWIGLU = 2, EPI_SILU = 3, EPI_GELU = 4, EPI_GEGLU = 5 };

struct GemvArgs {
    const bf16_t* x;        // [M, ldx]
    const bf16_t* W;        // [N, ldw]   (K contiguous)
    const bf16_t* norm_w;   // [K] or null: fused RMSNorm prologue on x
    const bf16_t* bias;     // [N] or null
    const bf16_t* res;      // [M, ldres] (EPI_RESID)
    bf16_t* out;            // [M, ldo]   (EPI_SWIGLU: N/2 columns)
    int M, N, K;
    int ldx, ldw, ldres, ldo;
    float eps;
    int epi;
    int rows_per_block;     // 0 = heuristic
    const float* wscale;    // non-null: W holds OCP fp8 e4m3 bytes [N, ldw] with one fp32 scale per row (K % 16 == 0)
};
int launch_gemv(const GemvArgs& a, hipStream_t s);
// o_proj of a tensor-parallel shard's one-row step with the decode attention's split merge in its prologue (gemv_merge.hip):
// x = merge of the live splits in decode_fused_kernel's workspace (batch row 0), out = epi(x W^T [+ res]); bit-identical to
// decode_fused_combine_kernel followed by the wave-form GEMV.  K = H * 128 <= 1024, nsplit <= 8, N >= 1024 (gemv_merge_ok).
struct GemvMergeArgs {
    const float* ws;        // [H][nsplit][130] split states of the one batch row (decode_fused_ws_floats layout)
    const int32_t* slot;    // [1] cache slot of the new token: live splits = slot / 128 + 1
    int nsplit, H;          // splits the attention launch was sized for, local heads
    const bf16_t* W;        // [N, ldw]
    const bf16_t* res;      // [N] (EPI_RESID)
    bf16_t* out;            // [N]
    int N, K, ldw, epi;
    bf16_t* x_out;          // null, or [K]: the merged attention output (what the combine launch would have written)
};
bool gemv_merge_ok(int heads, int D, int N, int nsplit);
int launch_gemv_merge(const GemvMergeArgs& a, hipStream_t s);
// 2..16 rows through LDS-DMA stages and v_mfma_f32_16x16x32_bf16 (gemv_thin.hip); needs K % 256 == 0, no fused norm, bf16 weights
bool gemv_thin_ok(const GemvArgs& a);
int launch_gemv_thin(const GemvArgs& a, hipStream_t s);

// Implicit-GEMM 3x3 convolution over an NHWC activation: A is [B, Hin, Win, Cin], the GEMM row m is the output
// pixel (b, yo, xo), K = 9*Cin ordered (ky, kx, ci) -- weights repacked to [Cout, 3, 3, Cin].  Cin % 64 == 0.
enum EmuConvMode { CONV_NONE = 0, CONV_3X3 = 1, CONV_3X3_S2 = 2, CONV_3X3_UP2 = 3 };
struct ConvGeom {
    int mode, Hin, Win, Hout, Wout, Cin;
    // set by launch_gemm (0 = off): the gather's fast form for the stride-1 / stride-2 modes.  A k tile (64 channels) lies inside one
    // filter tap; tap = k_tile / cpt is taken as (k_tile * cpt_magic) >> 16 (exact for the < 9 * cpt tiles of a conv), and a tap's
    // source is the output pixel's centre source + (ky - 1) * Win + (kx - 1) pixels, valid where a 9-bit mask computed once per row says
    int cpt = 0, cpt_magic = 0;
};

struct GemmArgs {
    const bf16_t* A;        // [M, lda]  activations (K contiguous)   (conv: NHWC input)
    const bf16_t* W;        // [N, ldw]  weights     (K contiguous)
    const bf16_t* bias;     // [N] or null
    const bf16_t* res;      // [M, ldres] (EPI_RESID)
    bf16_t* C;              // [M, ldc]  (EPI_SWIGLU / EPI_GEGLU: N/2 columns)
    int M, N, K;
    int lda, ldw, ldres, ldc;
    int epi;
    ConvGeom conv;          // mode 0 = plain GEMM
    const bf16_t* bias2;    // [M / rows_per_batch, ld_bias2] or null: per-batch bias added after the first rounding
    int rows_per_batch;     //   (ResnetBlock2D: conv1(x) + time_emb_proj(silu(temb))[:, :, None, None])
    int ld_bias2;
    // optional split-K scratch (fp32, caller-owned): few-tile long-K problems (UNet 32x32-level convs, ff-out) are cut
    // into ksplit K-slices of the big 256x128 tile so every CU gets one workgroup; slices land here and a second
    // launch sums them and applies the epilogue.  null = never split.
    float* partial = nullptr;
    size_t partial_floats = 0;
    int ksplit = 1;         // set by launch_gemm: K-slices of the tail tiles
    // fp8 GEMM (launch_gemm_fp8): A and W hold OCP e4m3 bytes (lda / ldw in elements = bytes), one fp32 scale per
    // activation row / weight row; the fp32 sum is multiplied by a_scale[m] * w_scale[n] ahead of the epilogue
    const float* a_scale = nullptr;
    const float* w_scale = nullptr;
    int full_tiles = 0;     // set by launch_gemm: tiles [0, full_tiles) run whole-K (workgroups [0, full_tiles)); every
                            // later tile t is cut into ksplit slices (workgroup full_tiles + (t - full_tiles) * ksplit + ks)
                            // whose fp32 tiles land compactly at partial[((t - full_tiles) * ksplit + ks) * BM * BN]
    int slice_rr = 0;       // A/B only (emu_gemm_tune bit 1): deal the slice workgroups tile by tile round-robin over the XCDs
                            // (the order before round 3) instead of the XCD-aware slice-major order
    // ---- fused LayerNorm (UNet transformer blocks: removes the LayerNorm launch between two GEMMs)
    // Producer side: besides C, emit per-row partial (sum, sum of squares) of the bf16-rounded outputs, one pair per
    // 128-column slot: row_stats_out[(slot * M + m) * 2 + {0, 1}], slot = n / 128 (N % 128 == 0; EPI_NONE / EPI_RESID).
    float* row_stats_out = nullptr;
    // Consumer side: A holds the UN-normalised rows x [M, K], W holds W * gamma (bf16), and the epilogue computes
    //   out[m, n] = rstd_m * (acc[m, n] - mean_m * ln_c[n]) + ln_d[n]
    // with mean_m / rstd_m from the ln_slots partial pairs of row m (LayerNorm over the K columns of A, eps ln_eps),
    // ln_c[n] = sum_k float(W'[n, k]) and ln_d[n] = sum_k W[n, k] * beta[k] + bias[n] (fp32; `bias` must be null).
    const float* ln_c = nullptr;
    const float* ln_d = nullptr;
    const float* ln_stats = nullptr;
    int ln_slots = 0;
    float ln_eps = 0.f;
    // ---- V^T epilogue (self-attention qkv projection): output columns n >= vt_col0 are the V heads; instead of row-major C
    // they are stored key-contiguous for the P.V MFMA: vt_out[((b * (N - vt_col0) + (n - vt_col0)) * vt_spad) + s],
    // m = b * vt_s + s  (= [B, H, 64, S_pad] with H * 64 = N - vt_col0).  EPI_NONE only.
    bf16_t* vt_out = nullptr;
    int vt_col0 = 0, vt_s = 0, vt_spad = 0;
    // ---- cross-attention epilogue (UNet attn2 over <= 64 fixed context tokens): this GEMM is the to_q projection of head-dim-64
    // heads (N = heads * 64); instead of writing q, every wave runs the whole attention of its (32 queries, 1 head) tile
    // against the cached K rows / V^T of the prompt -- S = K q^T, softmax over the keys, O = V^T P -- and C receives the
    // attention output (what the flash kernel would have written), so neither q nor a second launch exists.  128 x 64 tile only.
    //   cross_k  [B * cross_n, cross_ldk] K rows of every batch element (head h at columns h * 64 ..)
    //   cross_vt [B, N / 64, 64, cross_npad] V^T, key-contiguous (zero beyond cross_n)
    //   rows m of batch element b = m / cross_rows; scores scaled by cross_scale
    const bf16_t* cross_k = nullptr;
    const bf16_t* cross_vt = nullptr;
    int cross_ldk = 0, cross_n = 0, cross_npad = 0, cross_rows = 0;
    float cross_scale = 0.f;
    int sup_m = 0, sup_n = 0;   // set by launch_gemm (lock-step tiles, unsplit launches): every XCD owns one sup_m x sup_n block of
                            // tiles instead of a run of the column-major order (fewer distinct A + W rows per L2); 0 = off
    int stage_vt = 0;       // set by launch_gemm: whole V^T tiles leave through the transposed staging (gemm_tile.h::EpiStageT)
    int stage = 0;          // set by launch_gemm: the staged (LDS-transposed, 16-byte coalesced) epilogue may be taken (gemm_tile.h)
    // ---- per-workgroup timeline (tools/gemm_trace.py; written only by a library built with -DEMU_TRACE): 8 x u64 per workgroup
    unsigned long long* trace = nullptr;
    // ---- LLaMA prefill qkv projection with RoPE, KV append and V^T in the epilogue (256x256 tile only; N = 3 * rope_hl * 128,
    // columns [q heads | k heads | v heads], ONE batch element whose rows sit at cache slots rope_slot[m] = m): q is rotated and
    // stored to C, k is rotated and stored to rope_kc[head][slot][128], v goes to rope_vc[head][slot][128] and, key-contiguous, to
    // vt_out (GemmArgs::vt_* as above, vt_col0 = 2 * rope_hl * 128).  Same arithmetic as rope_kv_kernel (attention.hip).
    const bf16_t* rope_cos = nullptr;   // [max_pos, 128]
    const bf16_t* rope_sin = nullptr;
    const int32_t* rope_pos = nullptr;  // [M] rope position of every row
    const int32_t* rope_slot = nullptr; // [M] cache slot of every row
    bf16_t* rope_kc = nullptr;          // [rope_hl, rope_smax, 128] of this layer and batch element
    bf16_t* rope_vc = nullptr;
    int rope_hl = 0, rope_smax = 0;
    // ---- K-sliced GEMM whose slice sum also applies the RMSNorm that follows (LLaMA prefill: o_proj -> post-attention norm,
    // down_proj -> the next layer's input norm; 256x256 tile with EVERY tile K-sliced): the fp32 slices are laid out row-major
    // over the whole output (partial[(ks * M + m) * N + n], slab_rows = 1) and ONE workgroup per row sums them in slice order,
    // applies the epilogue (C), and writes norm_out = bf16(norm_w * bf16(C * rsqrt(mean(C^2) + eps))) with rmsnorm_kernel's
    // arithmetic -- the reduce launch and the rmsnorm launch in one, bit-identical to the pair.  The K-sliced 256 x 128 lock-step
    // tile has the same form (ViT fc2 -> LayerNorm + residual).
    const bf16_t* norm_w = nullptr;     // [N]
    const bf16_t* norm_b = nullptr;     // [N] or null.  Non-null: LayerNorm (layernorm_kernel's arithmetic) instead of RMSNorm, and
    const bf16_t* norm_res = nullptr;   //   norm_out = bf16(norm_res + bf16(LN(C))) when norm_res is set (ViT post-norm block); C may be null then
    int norm_ldres = 0;
    bf16_t* norm_out = nullptr;         // [M, norm_ld]
    int norm_ld = 0;
    float norm_eps = 0.f;
    int slab_rows = 0;                  // set by launch_gemm
    // ---- successor prefetch: the weights of the launch that FOLLOWS this one on the stream (a model forward knows them).  Every
    // thread requests up to two 128-byte lines of [pf_ptr, pf_ptr + pf_bytes) -- one dword each, consecutive lines on consecutive
    // threads, results discarded -- ahead of its own first k tile, so the matrix travels HBM -> infinity cache while this GEMM runs
    // and the successor finds it there (a GEMM over weights last touched a whole denoise step ago starts 1.5-4.7 us late and stalls
    // on every k tile of a short K otherwise: profiles/r04_mall_prefetch_probe.log).  Used by the UNet's transformer blocks, whose
    // matrices (3-26 MB) fit the two lines per thread: same-run 25.07 -> 24.19 ms per denoise step.  NOT used where a matrix is larger
    // than that (ViT fc1 / fc2 55 MB, the LLaMA prefill): whole-matrix requests queue ahead of the wave's own first tile (ViT +4 %),
    // and a capped k-major cover of every row's first tiles measured +1 % (ViT) / +2 % (prefill) and only -2 % on the UNet
    // (scattered lines): profiles/r05_prefetch_ab_*.log.  nullptr = off; emu_gemm_tune bit 16 switches it off globally (A/B).
    const void* pf_ptr = nullptr;
    size_t pf_bytes = 0;
};
int launch_gemm(const GemmArgs& a, hipStream_t s);
// tools hook: where the next GEMM launches of a -DEMU_TRACE build write their per-workgroup timelines (nullptr = off)
void emu_gemm_trace_set(unsigned long long* buf);
unsigned long long* emu_gemm_trace_get();
void emu_gemm_trace_select(long n);     // trace only the n-th GEMM launch from now on (-1: every launch)
// fp8 x fp8 -> bf16 on the block-scaled MFMA (256x256 ping-pong tile only): K % 128 == 0, a.a_scale / a.w_scale set
int launch_gemm_fp8(const GemmArgs& a, hipStream_t s);
// the 256x256 ping-pong tile (gemm256.hip), dispatched by launch_gemm; tiles [0, full_tiles) whole-K, the rest in ksplit
// K-slices of EMU_GEMM256_SLICE_FLOATS fp32 each in a.partial (full_tiles < 0: no slicing)
int launch_gemm256(const GemmArgs& a, hipStream_t s, int full_tiles, int ksplit);
int gemm256_tiles(const GemmArgs& a);           // workgroups of a whole-K launch (the last row of tiles carries M % 256 <= 32)
bool gemm256_ok(const GemmArgs& a);             // K % 64 == 0 and operands within reach of 32-bit descriptor offsets
constexpr size_t EMU_GEMM256_SLICE_FLOATS = 288 * 256;
// scratch that always suffices: at most 256 slices (one per CU) of the largest tile (256 x 256 + 32 remainder rows, fp32)
constexpr size_t EMU_SPLITK_SCRATCH_FLOATS = (size_t)512 * 288 * 256;
// process-wide default split-K scratch for callers that do not pass one (the C-ABI primitives); caller-owned memory
void emu_gemm_set_splitk_scratch(float* ptr, size_t floats);
// test / bench hook: pin the tile configuration ('B', 'C', 'K', 'S', 'P'; 0 = heuristic)
void emu_gemm_force_config_set(int cfg);
// A/B switches of single dispatch decisions (bit 0: GLU GEMMs the hybrid would split run as one launch of 128 x 128 tiles)
void emu_gemm_tune_set(int mask);
int emu_gemm_tune_get();

// ---- launch profiler (engine.hip; bench.py's per-kernel table of the denoise leg): HIP events around the MFMA-carrying launches
// (GEMM / conv / attention) while enabled -- eager launches only.  klass: a short static string.
bool emu_prof_on();
void emu_prof_begin(hipStream_t s);
void emu_prof_end(hipStream_t s, const char* klass, int M, int N, int K, int tag, double flops);
void emu_prof_drop();                   // the wrapped launch returned non-zero (not launched: -95 probe, -22): record nothing

// ---- row-wise / elementwise (elementwise.hip)
int launch_rmsnorm(const bf16_t* x, const bf16_t* w, bf16_t* y, int rows, int cols, int ldx, int ldy, float eps, hipStream_t s);
// y = (res ? res : 0) + LayerNorm(x) * w + b       (ViT post-norm residual, eva_vit.py:298-300)
int launch_layernorm(const bf16_t* x, const bf16_t* w, const bf16_t* b, const bf16_t* res, bf16_t* y,
                     int rows, int cols, float eps, hipStream_t s);
// the same + per-row e4m3 quantisation of the bf16 result in one pass (cols <= 2048; y may be null): q [rows, cols] bytes, scale [rows]
int launch_layernorm_q8(const bf16_t* x, const bf16_t* w, const bf16_t* b, const bf16_t* res, bf16_t* y, uint8_t* q, float* scale,
                        int rows, int cols, float eps, hipStream_t s);
int launch_embed_gather(const int32_t* ids, const bf16_t* table, bf16_t* out, int n_tok, int hidden, int vocab, hipStream_t s);
// out[dst_rows[i], :] = src[i, :]
int launch_scatter_rows(const bf16_t* src, const int32_t* dst_rows, bf16_t* out, int n_rows, int hidden, hipStream_t s);
int launch_argmax(const bf16_t* logits, int ld, int rows, int vocab, int suppress_id, int32_t* out, hipStream_t s);
// NCHW fp32/bf16 image -> [B*gh*gw, Kpad] bf16 patch matrix (k = c*p*p + i*p + j, zero padded to Kpad)
int launch_patchify(const void* image, int image_is_f32, bf16_t* out, int B, int C, int HW, int p, int Kpad, hipStream_t s);
// tokens [B, 1+g*g, C] (cls dropped) -> [B, (g/s)^2, C] average over s x s windows (emu.py:82-89)
int launch_avgpool_tokens(const bf16_t* x, bf16_t* out, int B, int g, int C, int s, hipStream_t s_);
// x[b, 0, :] = cls + pos[0]; x[b, 1+t, :] = patches[b, t, :] + pos[1+t]   (eva_vit.py:406-409)
int launch_vit_assemble(const bf16_t* patches, const bf16_t* cls, const bf16_t* pos, bf16_t* x, int B, int T, int C, hipStream_t s);

// ---- attention (attention.hip)
struct RopeKvArgs {
    bf16_t* qkv;            // [B*T, 3*Hl*D] rows; q at col 0, k at Hl*D, v at 2*Hl*D  (q, k rotated in place)
    const bf16_t* cos;      // [max_pos, D] bf16 (cat(freqs, freqs) layout)
    const bf16_t* sin;
    const int32_t* pos;     // [B*T] rope position of every row
    const int32_t* slot;    // [B*T] kv-cache slot of every row (absolute index into S_max)
    bf16_t* kcache;         // [B, Hl, S_max, D]
    bf16_t* vcache;         // [B, Hl, S_max, D]
    int B, T, Hl, D, S_max;
};
int launch_rope_kv(const RopeKvArgs& a, hipStream_t s);

// Vt[b, h, d, s] = V[b, h, s, d] for s < S (zero for S <= s < S_pad); strided source
struct TransposeVArgs {
    const bf16_t* v; long v_sb, v_sh, v_ss;    // element strides (batch, head, token); d contiguous
    bf16_t* vt;                                // [B, H, D, S_pad]
    int B, H, S, D, S_pad;
};
int launch_transpose_v(const TransposeVArgs& a, hipStream_t s);

struct FlashArgs {
    const bf16_t* q; long q_sb, q_sh, q_ss;    // element strides; d contiguous
    const bf16_t* k; long k_sb, k_sh, k_ss;
    const bf16_t* vt;                          // [B, H, D, Sk_pad]
    bf16_t* o; long o_sb, o_sh, o_ss;
    const int32_t* kstart;                     // [B] or null: keys < kstart[b] are masked (left padding)
    int B, H, Sq, Sk, Sk_pad, D;               // D in {64, 128}
    int causal;                                // query i attends keys <= i + (Sk - Sq)
    float scale;
    int xcd_remap = 1;                         // set by launch_flash_attn: XCD-aware (head, query block) order
    int stage_o = 0;                           // set by launch_flash_attn: O leaves through LDS as 16-byte row-contiguous stores
    int heavy_first = 0;    // set by launch_flash_attn: causal prefill, XCDs walk their heads' query blocks from the longest down
};
int launch_flash_attn(const FlashArgs& a, hipStream_t s);

struct DecodeAttnArgs {
    const bf16_t* q; long q_sb, q_sh;          // [B, H, D] strided
    const bf16_t* kcache;                      // [B, H, S_max, D]
    const bf16_t* vcache;
    bf16_t* o; long o_sb, o_sh;
    const int32_t* kstart;                     // [B] or null
    const int32_t* ctx_ptr;                    // device int32 or null: overrides ctx (graph replay)
    float* ws;                                 // workspace: B*H*nsplit*(D+2) floats
    int B, H, D, S_max, ctx;                   // keys [0, ctx) valid
    int ctx_max;                               // sizes the launch (>= ctx); 0 -> ctx
    float scale;
};
int decode_attn_nsplit(int ctx);
// out_ids[step, b] = cur_ids[b]; pos[b]++, slot[b]++; ctx++, step++   (greedy loop state, all on device)
int launch_greedy_advance(const int32_t* cur_ids, int32_t* pos, int32_t* slot, int32_t* ctx, int32_t* step,
                          int32_t* out_ids, int B, hipStream_t s);
int launch_decode_attn(const DecodeAttnArgs& a, hipStream_t s);
// out_all[step[0]] = prev = src ([B, cols] bf16); pos[b]++, slot[b]++, step[0]++   (generate_image's loop state, on the device)
int launch_regress_advance(const bf16_t* src, bf16_t* out_all, bf16_t* prev, int32_t* pos, int32_t* slot, int32_t* step, int B,
                           int cols, hipStream_t s);

// Decode step, fused: RoPE of q and of the new k, KV-cache append of the new token, and single-query attention over
// the cache, in one launch (+ the split combine).  Context length of row b = slot[b] + 1, read on the device.
struct DecodeFusedArgs {
    const bf16_t* qkv;                         // [B, 3*H*D] projection rows (q | k | v), not yet rotated
    const bf16_t* cos; const bf16_t* sin;      // [max_pos, D]
    const int32_t* pos;                        // [B] RoPE position of the new token
    const int32_t* slot;                       // [B] cache slot of the new token
    bf16_t* kcache; bf16_t* vcache;            // [B, H, S_max, D]
    bf16_t* o; long o_sb, o_sh;
    const int32_t* kstart;                     // [B] or null
    float* ws;                                 // decode_fused_ws_floats(B, H, D, ctx_max)
    int B, H, D, S_max, ctx_max;               // ctx_max sizes the launch (>= max slot + 1)
    float scale;
    // Beam search: rows come in groups of share_nb beams of one prompt whose first share_len cache slots (the prompt) are
    // identical.  They are stored ONCE, in the group's first row, and read from there by every beam of the group (the
    // workgroups reading the same keys are dispatched onto one XCD so its L2 serves all but the first).  share_nb <= 1:
    // every row owns all of its slots.
    int share_nb = 0, share_len = 0;
    // Non-null (one-row path only): the last split workgroup of a head to arrive merges the head's live splits itself
    // (arrive[b * H + h]: zero between launches, the merging workgroup resets it) -- no decode_fused_combine_kernel launch.
    int* arrive = nullptr;
    // one-row path: leave the split states in ws and launch no combine -- the consumer merges them itself (launch_gemv_merge)
    bool skip_combine = false;
};
constexpr int DECODE_SHARE_MAX = 8;            // beams per group the shared-prefix path takes
size_t decode_fused_ws_floats(int B, int H, int D, int ctx_max);
int launch_decode_fused(const DecodeFusedArgs& a, hipStream_t s);

struct EmuP2p;
// ---- whole decoder layers of a single-row decode step in ONE launch (decode_layer.hip): qkv projection (+RMSNorm) -> RoPE / KV
// append / split attention + merge -> o_proj (+residual) -> [all-reduce] -> gate/up (+RMSNorm, SwiGLU) -> down (+residual) ->
// [all-reduce], for layers [layer0, layer0 + nlayers).  Workgroups are dealt roles by index (in dispatch order: every dependency
// points to a lower index), wait on arrival counters with their weight slice already requested, and hand results over with
// agent-scope (write-through) stores; every wait is bounded in wall-clock time (err counts give-ups; results are garbage then).
struct DecodeLayerPtrs { const bf16_t *wqkv, *wo, *wgu, *wdown, *ln1, *ln2; };
struct DecodeLayersArgs {
    const DecodeLayerPtrs* table;              // device memory, indexed by absolute layer
    int layer0, nlayers;
    bf16_t* hA;                                // [H] layer input / output (residual stream)
    bf16_t* hB;                                // [H] post-attention residual stream
    bf16_t* qkv;                               // [3 * Hl * 128]
    bf16_t* attn;                              // [Hl * 128]
    bf16_t* act;                               // [Fl]
    float* ws;                                 // decode_fused_ws_floats(1, Hl, 128, ctx_max)
    const bf16_t* cos; const bf16_t* sin;
    const int32_t* pos; const int32_t* slot; const int32_t* kstart;
    bf16_t* kcache; bf16_t* vcache;            // layer l at + l * kv_layer elements: [Hl, S_max, 128]
    size_t kv_layer;
    int H, Hl, Fl, S_max, ctx_max;
    float eps, scale;
    int epi_res;                               // 1: this rank adds the residual (rank 0, or no tensor parallelism)
    int* cnt;                                  // decode_layers_cnt_ints(layers, Hl) ints, ZERO at launch (all layers of the table)
    unsigned* err;                             // give-up counter (device)
    long long limit_ticks;                     // wait bound, 100 MHz ticks
    // tensor parallelism: the workgroup that completes o_proj / down_proj all-reduces the partial vector over the peers' comm
    // blocks (p2p.hip's protocol and sequence counters) before the consumers are released.  tp_n == 0: no all-reduce.
    char* tp_block[8];
    unsigned long long* tp_seq;
    int tp_n, tp_rank;
    // roles [role0, role1) of every layer run in this launch (0 q, 1 attention, 2 o_proj, 3 gate/up, 4 down; 0, 0 = all).  A partial
    // range covers ONE layer; its first role's input comes from the launch before (tensor parallelism without in-kernel all-reduce:
    // [q, attention, o_proj] -> all-reduce launch -> [gate/up, down] -> all-reduce launch).
    int role0, role1;
    unsigned long long* trace;                 // tools only (nullptr = off): 4 x u64 per workgroup {role | layer << 8, entry, input ready, exit}
    // set by launch_decode_layers
    int nQ, nA, nO, nG, nD, per_layer, cnt_stride, wave_od;
};
size_t decode_layers_cnt_ints(int layers, int Hl);
bool decode_layers_ok(const DecodeLayersArgs& a);
int launch_decode_layers(DecodeLayersArgs a, hipStream_t s);
// the comm blocks / sequence counters of an opened EmuP2p for DecodeLayersArgs::tp_* (false: peers not mapped)
bool emu_p2p_view(EmuP2p* p, char** block8, unsigned long long** seq, int* n, int* rank, long long* limit_ticks);
// the engine area of every mapped comm block (decode_engine.hip's in-launch all-reduce) and this rank's per-CU counters
bool emu_p2p_engine_view(EmuP2p* p, char** area8, unsigned int** seq, int* n, int* rank);

// ---- persistent weight-streaming engine for chains of one-row GEMVs (decode_engine.hip): one launch, one resident workgroup per CU
// (loader wave + three consumer waves on an LDS ring), op outputs handed between CUs through 4-byte {tag, bf16} granules
constexpr int ENG_MAX_OPS = 6;
constexpr int EMU_ENG_MAX_RANKS = 8;
constexpr int ENG_AR_BUFS = 4;                 // comm arrays per rank, reused round-robin (an array is rewritten four all-reduces later)
constexpr int ENG_AR_MAXLEN = 8192;            // granules per comm array
constexpr size_t EMU_P2P_ENG_BYTES = (size_t)ENG_AR_BUFS * ENG_AR_MAXLEN * 4;   // engine area at the end of every P2P comm block
struct EngOp {
    const bf16_t* W;                           // [N, K] row-major, K contiguous, K <= 6656
    int N, K;
    const bf16_t* gain;                        // RMSNorm gain [K] (fused prologue on the activation vector) or null
    float eps;
    int epi;                                   // EPI_NONE / EPI_RESID / EPI_SWIGLU (rows (2j, 2j + 1) = (gate_j, up_j))
    const bf16_t* res;                         // [N] (EPI_RESID)
    int vw;                                    // launch_gemv's column partition for this shape: 4 (block kernels) or 1 (wave kernel)
    // input vector: 0 = xg, a bf16 vector written BEFORE this launch; 1 = xgran, granules of an earlier op of this launch;
    // 2 = the tensor-parallel SUM of an earlier op's partial outputs: every CU reduces its share of the ranks' comm arrays (ar_k-th
    // all-reduce of the launch) in rank order (p2p.hip's arithmetic), publishes it to xgran, and gathers xgran like case 1
    int x_src;
    const bf16_t* xg;
    uint32_t* xgran;
    bf16_t* sum_out;                           // x_src 1 / 2: the gathered vector also goes here as plain bf16 (read by LATER launches), or null
    int keep_raw;                              // the gathered (un-normalised) vector stays in LDS for a later op's residual (res_src = 1)
    // output vector: 0 = out, plain bf16 (read by a LATER launch); 1 = ogran, granules (zero at launch); 2 = this rank's comm array
    // of the ar_k-th all-reduce of the launch (partial sums, read by every rank's reducers)
    int out_dst;
    bf16_t* out;
    uint32_t* ogran;
    int ar_k;                                  // out_dst 2 / x_src 2: which all-reduce of this launch
    int res_src;                               // EPI_RESID: 0 = res (global, written before this launch); 1 = the vector kept by keep_raw
    // set by launch_decode_engine: units (rows, or (gate, up) pairs) per CU = q (+ 1 for the first rem CUs), whole rows per 16 KiB fill,
    // bytes / LDS-DMA instructions per fill, fills that go to one consumer together, fills per CU (q / q + 1 units)
    int q, rem, rps, fill_bytes, ni, grp, nfills_lo, nfills_hi;
};
struct EngArgs {
    EngOp op[ENG_MAX_OPS];
    int nops;
    unsigned* err;                             // give-up counter (device)
    long long limit_ticks;                     // bound of every wait, 100 MHz ticks
    int ncu;                                   // workgroups = CUs of the device (all must be resident)
    // tensor parallelism (tp_n >= 1 with all-reduce ops): every rank's engine area inside its P2P comm block (p2p.hip), ENG_AR_BUFS
    // arrays of ar_len granules used round-robin by the all-reduces in launch order (a per-CU device counter numbers them: the tag)
    char* comm[EMU_ENG_MAX_RANKS];
    int tp_n, tp_rank, n_ar, ar_len;
    unsigned int* seq;                         // [ncu] all-reduces this CU has been through (device memory, owned by the comm block's creator)
    int nload;                                 // loader waves per workgroup (0 = default)
    int dbg;                                   // tools: bit 0 = consumers acknowledge fills without multiplying (loader ceiling)
    int nslot, xbytes;                         // set by launch_decode_engine
};
int launch_decode_engine(EngArgs a, hipStream_t s);
// which column partition launch_gemv's kernel uses for a one-row bf16 GEMV of this shape (the engine mirrors it bit for bit):
// 1 = gemv_wave_kernel (one wave per row), 4 = the 256-thread block kernels
int emu_gemv_partition(int N, int K, bool norm, int epi);

// One beam-search step on the device (beam.hip): see emu_beam_step_bf16 in include/emu_hip.h
struct BeamStepArgs {
    const bf16_t* logits;                      // row of (prompt b, beam j) = logits + b * ld_prompt + j * ld_beam
    long ld_prompt, ld_beam;                   // elements (step 0: ld_beam = 0, every beam continues the prompt)
    int V, B, nb, L, cur;                      // vocabulary, prompts, beams, max new tokens, tokens generated so far
    const int32_t* cur_dev;                    // non-null: the step index is read here instead (hipGraph replay); steps >= L are no-ops
    int min_len, eos_id;                       // EOS is masked while cur < min_len
    float length_penalty;
    int hf431;                                 // scorer conventions: 0 = transformers 5.x, 1 = transformers 4.31 (beam.hip)
    int32_t* running_seq; int32_t* sequences;  // [B, nb, L]
    float* running_scores; float* beam_scores; // [B, nb]
    unsigned char* finished; int32_t* seq_len; // [B, nb]
    unsigned char* heuristic_open;             // [B]
    int32_t* next_tok; long* beam_flat;        // [B * nb] out: token to feed / cache row it continues
};
// device-side step bookkeeping and KV re-order of a hipGraph-replayed beam step (beam.hip)
int launch_beam_advance(int32_t* cur_dev, int32_t* pos, int32_t* slot, const int32_t* pos0, int slot0, int rows, int L, int phase,
                        hipStream_t s);
int launch_beam_reorder(bf16_t* kc, bf16_t* vc, const long* beam_flat, const int32_t* cur_dev, int layers, int rows, int Hl, int s_max,
                        int D, int nb, int slot0, int L, hipStream_t s);
size_t beam_step_ws_floats(int B, int nb, int V);     // scratch of the two-launch step (chunk partials)
int launch_beam_step(const BeamStepArgs& a, float* ws, size_t ws_floats, hipStream_t s);

// ---- UNet denoise helpers (unet.hip); activations are NHWC: [B, H*W, C] bf16
// GroupNorm(groups, eps) (+ optional SiLU) over x [B, HW, C]: three launches (partial sums, finalize to per-(b, c)
// scale/shift, apply).  ws must hold gn_ws_floats(B, C, HW) floats.
size_t gn_ws_floats(int B, int C, int HW);
int launch_groupnorm(const bf16_t* x, const bf16_t* gamma, const bf16_t* beta, bf16_t* y, float* ws, int B, int HW, int C,
                     int groups, float eps, int silu, hipStream_t s);
// out[m, :C1] = a[m, :], out[m, C1:] = b[m, :]
int launch_concat_channels(const bf16_t* a, const bf16_t* b, bf16_t* out, int rows, int C1, int C2, hipStream_t s);
// latents NCHW [1, C, H, W] -> conv_in im2col matrix [2*H*W, Kpad] (both CFG halves identical), scaled by
// 1/sqrt(sigma[step]^2 + 1) (EulerDiscreteScheduler.scale_model_input); k = (ky*3 + kx)*C + c, zero padded
int launch_unet_prep_input(const bf16_t* latents, const float* sigmas, const int32_t* step, bf16_t* out, int C, int H, int W,
                           int Kpad, hipStream_t s);
// classifier-free guidance (cond first) + Euler step on NCHW latents, then step++ :
//   eps = u + g*(c - u); x += eps * (sigma[step+1] - sigma[step])   with the bf16 rounding points of the reference ops
int launch_cfg_euler_step(const bf16_t* eps_nhwc, bf16_t* latents, const float* sigmas, int32_t* step, float guidance, int C,
                          int HW, hipStream_t s);
// out[r, :] = bf16(silu(bf16(a[r, :] + b[r, :])))   (silu(emb + aug_emb) feeding every time_emb_proj)
int launch_add_silu(const bf16_t* a, const bf16_t* b, bf16_t* sum_out, bf16_t* silu_out, int n, hipStream_t s);
// out[r, :] = table[step[0], :] for r < rows
int launch_gather_step_row(const bf16_t* table, const int32_t* step, bf16_t* out, int rows, int cols, hipStream_t s);

// stand-alone successor prefetch (common.h::pf_touch): [ptr, ptr + bytes) -> infinity cache, workgroups x 256 threads
int launch_prefetch(const void* ptr, size_t bytes, int workgroups, hipStream_t s);
// in-place row softmax of x [rows, ld] over the first `cols` columns: x = bf16(softmax(float(x) * scale))
// (materialised-score attention for head dims the flash kernel does not cover: the VAE mid block, D = 512)
// per-row fp8 e4m3fn quantisation: scale[n] = amax_n / 448 (1 for an all-zero row), q = rne(w / scale)
int launch_quant_fp8_rows(const bf16_t* w, int ldw, uint8_t* q, int ldq, float* scale, int N, int K, hipStream_t s);
int launch_softmax_rows(bf16_t* x, const bf16_t* bias, int rows, int cols, int ld, int ld_bias, float scale, hipStream_t s);

// ---- one-shot peer-to-peer all-reduce over IPC-mapped comm blocks (p2p.hip)
constexpr int EMU_P2P_MAX_RANKS = 8;
constexpr size_t EMU_P2P_SLOT_BYTES = 256 * 1024;
// a message is cut into pieces of EMU_P2P_PIECE elements, each with its own sequence counter and flag per slot; the comm block is
// [slot 0 | slot 1 | flags: 2 x EMU_P2P_PIECES x u64]
constexpr int EMU_P2P_PIECE = 512 * 8;
constexpr int EMU_P2P_PIECES = (int)(EMU_P2P_SLOT_BYTES / (EMU_P2P_PIECE * 2));
struct EmuP2p;
EmuP2p* emu_p2p_create(int rank, int n, void* handle64_out);          // nullptr on failure
int emu_p2p_open(EmuP2p* p, const void* handles);                      // n x 64 bytes, rank order (own entry ignored)
void emu_p2p_destroy(EmuP2p* p);
void emu_p2p_set_timeout_ms(EmuP2p* p, int ms);
void emu_p2p_set_fenced(EmuP2p* p, int fenced);                        // 1 (default): system-scope fences around the exchange; 0: fence-free form
int emu_p2p_fenced(const EmuP2p* p);
int emu_p2p_allreduce(EmuP2p* p, bf16_t* x, size_t n, hipStream_t s);  // in place; > one slot goes through in chunks
unsigned int emu_p2p_giveups_read();
