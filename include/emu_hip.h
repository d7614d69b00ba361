/* emu_hip.h -- C ABI of libemu_hip.so, the MI355X (gfx950) compute library behind emu_amd.
 *
 * The reference (baaivision/Emu, Emu2/) has no FFI layer: its hot path is Python calling torch /
 * transformers / diffusers ops.  This header is the operator interface those call sites bind to when the
 * arithmetic is replaced by hand-written HIP kernels; every entry cites the reference lines it replaces
 * (paths relative to the reference root).  INTEGRATION.md shows the ctypes stubs.
 *
 * Conventions
 *  - plain C symbols, raw device pointers + sizes, no torch types; bf16 tensors are `uint16_t` bit patterns,
 *    row-major, innermost dimension contiguous; all "ld*"/stride arguments are in ELEMENTS.
 *  - every launch takes an explicit HIP stream (`emu_stream_t` = hipStream_t) and is asynchronous;
 *    nothing here allocates, frees or synchronises device memory (hipGraph / stream-capture safe).
 *  - return value: 0 = OK, negative = -errno style (-22 = EINVAL: unsupported shape/alignment),
 *    positive = hipError_t / ncclResult_t (offset by 1000) of the failing runtime call.
 *  - one context per (process, device); thread-compatible, not thread-safe per context.
 *  - the caller owns all tensors (e.g. torch allocator); the library owns only its small host-side tables.
 */
#ifndef EMU_HIP_H
#define EMU_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* emu_stream_t;
typedef struct emu_ctx emu_ctx;
typedef struct emu_llama emu_llama;
typedef struct emu_vit emu_vit;

enum { EMU_EPI_NONE = 0, EMU_EPI_RESID = 1, EMU_EPI_SWIGLU = 2, EMU_EPI_SILU = 3, EMU_EPI_GELU = 4, EMU_EPI_GEGLU = 5 };

/* ---- context / tensor-parallel communicator ------------------------------------------------------------
 * Replaces the reference's layer-placement "model parallel" (Emu2/emu/mixin.py:14-85, chat.py:235-283): the
 * decoder is tensor-parallel, one process per GPU, partial sums exchanged with RCCL all-reduce over xGMI. */
int emu_version(void);
int emu_ctx_create(int device, int tp_rank, int tp_size, emu_ctx** out);
void emu_ctx_destroy(emu_ctx* ctx);
const char* emu_last_error(const emu_ctx* ctx);
int emu_tp_unique_id(void* out128);                      /* rank 0: 128-byte RCCL unique id           */
int emu_tp_init(emu_ctx* ctx, const void* id128);        /* all ranks: ncclCommInitRank               */
int emu_allreduce_bf16(emu_ctx* ctx, void* buf, size_t n, emu_stream_t s);   /* in-place sum          */
/* One-shot peer-to-peer all-reduce for the decode-sized messages (13 KB, 120 per token), emu_amd/csrc/p2p.hip: every rank
 * exports a comm block over HIP IPC (create: 64-byte handle out), maps every peer's (open: tp_size x 64 bytes in rank
 * order; timeout_ms bounds every device-side wait, <= 0 keeps the 10 s default), and reads the peers' partial vectors
 * straight over xGMI.  emu_allreduce_bf16 uses it for messages of at most 256 KiB once emu_tp_p2p_enable(ctx, 1) was
 * called (and for every message when no RCCL communicator exists); the host enables it only after a self-test through
 * emu_tp_p2p_allreduce_bf16 on every rank.  emu_tp_p2p_giveups: device-side waits that timed out since process start --
 * non-zero means sums since then are invalid. */
int emu_tp_p2p_create(emu_ctx* ctx, void* handle64_out);
int emu_tp_p2p_open(emu_ctx* ctx, const void* handles, int timeout_ms);
int emu_tp_p2p_allreduce_bf16(emu_ctx* ctx, void* buf, size_t n, emu_stream_t s);
int emu_tp_p2p_enable(emu_ctx* ctx, int on);
/* The exchange has two forms: fenced = 1 (the state of a fresh comm block) brackets it with system-scope release / acquire fences --
 * the HIP memory model's own guarantee --; fenced = 0 relies on write-through stores + acknowledgement and cache-bypassing loads alone
 * (3.3 instead of 6.8 us per 13 KB all-reduce).  The host selects 0 only after a soak of that form passed on every rank of the job
 * (emu_amd/llama.py::_init_p2p).  emu_tp_p2p_fenced: the current form (-1: no comm block). */
int emu_tp_p2p_set_fenced(emu_ctx* ctx, int fenced);
int emu_tp_p2p_fenced(const emu_ctx* ctx);
unsigned int emu_tp_p2p_giveups(void);

/* Optional fp32 scratch for split-K GEMMs / convolutions issued through the primitives below (emu_linear_bf16,
 * emu_conv3x3_nhwc_bf16): few-tile long-K problems are cut into up to 4 K-slices that land here before a second launch
 * applies the epilogue.  Caller-owned device memory, >= 4 * M * N * 4 bytes for the shapes that should split; NULL / 0
 * (the default) never splits.  One stream at a time may use it.  The UNet engine carries its own inside its workspace. */
void emu_set_splitk_scratch(void* ptr, size_t bytes);
/* Test / bench hook: pin the tile configuration of every following GEMM / conv launch of this process.  0 (default) =
 * the shape heuristic; 'B' 128x128, 'C' 256(n)x128(m), 'K' 128x64 with two k-groups, 'S' 256x128 K-sliced,
 * 'P' 256x256 ping-pong (K-sliced under half a round of tiles), 'Q' the same never sliced (both: K % 64 == 0 and
 * operands within 2 GiB, else the heuristic), 'H' whole rounds of the 256x256 tile + the remaining columns as a second
 * GEMM (where the tile count allows, else the heuristic).  The parity tests walk every configuration over the
 * bench's true shapes with it (tests/test_gpu_ops.py); production code never calls it. */
void emu_gemm_force_config(int cfg);
/* Bench hook: A/B switches of single dispatch decisions (0 = the shipped heuristic).  Bit 0: GLU GEMMs that the heuristic
 * splits into whole rounds of the 256x256 tile + a remainder GEMM run as ONE launch of 128x128 tiles.  Bit 1: the K-slice
 * workgroups of a split GEMM are dealt tile by tile round-robin over the XCDs (the order before round 3) instead of the
 * XCD-aware slice-major order.  Bit 2: 4..16-row linears skip the LDS-DMA + MFMA stream (gemv_thin.hip) and run on the
 * v_dot2c / register-fed MFMA kernels as before round 3; bit 3: the GEMM / conv epilogues store straight from the accumulator
 * layout (8 bytes per lane to 32 rows per instruction, as before round 4) instead of through the LDS-staged, row-contiguous
 * 16-byte form; bit 4: the lock-step tiles keep the column-major XCD runs instead of 2-D tile blocks per XCD; bit 5: no 128 x 128
 * tile for one-round problems; bit 6: the attention kernel deals its workgroups in launch order (query blocks of a head on 8
 * different XCDs); bit 7: it stores O straight from the accumulator layout; bits 12-13: 1 / 2 = attention always on 4 / 8 waves;
 * bit 14: V^T tiles of a fused qkv projection stored straight from the accumulators instead of through the transposed staging; bit 15:
 * causal attention launches keep the (head, query block) order instead of walking every XCD's heads from the longest query block
 * down; bits 8-11: variant of the thin stream (tools/thin_ab.py); bit 16: no successor-weight prefetch from inside the GEMM kernels;
 * bit 17: the 256 x 256 tile keeps the column-major XCD runs instead of 2-D tile blocks per XCD (unsliced plain GEMMs), bit 18: blocks
 * only for launches of more than one round; bit 19 (opt-in): a short tensor-parallel shard's one-row step merges the decode
 * attention's splits in the o_proj launch's prologue instead of a combine launch ahead of it (gemv_merge.hip: bit-identical, measured level);
 * bits 21-23 (round 6, the four-wave 256x256 tile of gemm_w4.hip): 21 = never take it (the eight-wave ping-pong tile everywhere: the
 * same-run A/B twin), 22 = take it wherever it is instantiated (tests: bf16 plain GEMMs and convs of the 256x256 configuration,
 * whatever epilogue, slices and raggedness), 23 = its fp32 K-slices leave by direct stores instead of through LDS. */
void emu_gemm_tune(int mask);

/* Tools hook (tools/gemm_trace.py): per-workgroup timelines of the following GEMM launches -- 8 x uint64 per workgroup at
 * buf[blockIdx * 8]: entry, first k tile landed, main loop done, stores complete (s_memrealtime, 100 MHz), XCC / HW id, shader
 * clock at entry / loop end, unused.  Written only by a library built with -DEMU_TRACE (`python -m emu_amd.build --trace` ->
 * libemu_hip_trace.so; emu_gemm_trace_built() says which one is loaded); NULL = off.  The production library ignores it. */
void emu_gemm_trace(void* buf);
void emu_gemm_trace_nth(long n);     /* trace only the n-th GEMM / conv launch from now on (a launch INSIDE a model's forward); -1: all */
int emu_gemm_trace_built(void);

/* Measurement hook (bench.py roofline leg): HIP-event timing of every M<=8 weight-streaming GEMV launched while
 * enabled (eager launches only, not inside stream capture).  read: sum of launch durations (ms), algorithmic
 * weight bytes (2*N*K per launch) and launch count since the last enable. */
int emu_profile_gemv(int enable);
int emu_profile_gemv_read(double* total_ms, double* weight_bytes, long* launches);

/* Measurement hook (bench.py: per-kernel table of the denoise leg): HIP events around every MFMA-carrying launch -- GEMM, implicit-
 * GEMM convolution, attention -- issued while enabled (eager launches only, not inside stream capture).  read: the launches since
 * the last enable, aggregated by (class, M, N, K, tag): klass "gemm" / "conv" / "attn"; for attn M = Sq, N = Sk, K = head dim and
 * tag = batch x heads; for gemm / conv tag = epilogue | fused-epilogue mask << 8.  Returns the number of rows written (<= cap). */
typedef struct {
    char klass[16];
    int M, N, K, tag, launches;
    double ms, flops;
} emu_prof_row;
int emu_profile_launches(int enable);
int emu_profile_launches_read(emu_prof_row* rows, int cap);

/* ---- primitive operators -------------------------------------------------------------------------------*/
/* torch.nn.functional.linear on the hot path (eva_vit.py:106,112,198,250; LlamaAttention/LlamaMLP linears
 * reached from emu.py:133-138,213-229; project_up/down emu.py:53,55,131,147,201).
 * C[m,n] = epi(sum_k A[m,k] W[n,k] (+bias[n])).  M <= 8 streams W once (HBM-bound GEMV, optional fused
 * LLaMA RMSNorm of A via norm_w/eps); M > 8 runs the MFMA GEMM (norm_w must be NULL).
 * EPI_RESID: C = bf16(res + bf16(.)).  EPI_SWIGLU/GEGLU: W rows interleaved (2j, 2j+1), C has N/2 columns. */
int emu_linear_bf16(const void* A, const void* W, const void* bias, const void* res, const void* norm_w,
                    void* C, int M, int N, int K, int lda, int ldw, int ldres, int ldc, float eps, int epi,
                    emu_stream_t s);
/* emu_linear_bf16 with an fp8 (e4m3fn) weight stream: W8 [N, ldw] bytes, wscale fp32 [N]; M <= 2, K % 16 == 0,
 * epi in {NONE, RESID, SWIGLU}.  out = epi(bf16((sum_k fp8(W8[n,k]) * x[k]) * wscale[n] + bias[n])) */
int emu_quantize_fp8_rows(const void* w_bf16, int ldw, void* q_fp8, int ldq, float* scale, int N, int K,
                          emu_stream_t s);   /* scale[n] = amax_n / 448 (1 if the row is zero); q = rne_e4m3fn(w / scale) */
/* fp8 x fp8 GEMM on the block-scaled matrix instruction (v_mfma_scale_f32_32x32x64_f8f6f4, twice the bf16 MFMA rate) for
 * the MFMA-bound shapes of BASELINE configs[4] ("fp8 MFMA weights"): A8 [M, lda] and W8 [N, ldw] hold OCP e4m3fn bytes
 * (both from emu_quantize_fp8_rows: one fp32 scale per row), K % 128 == 0, lda / ldw % 16 == 0.
 *   C[m, n] = epi(bf16(a_scale[m] * w_scale[n] * sum_k fp8(A8[m,k]) * fp8(W8[n,k]) + bias[n]))   (fp32 accumulation)
 * epi in {NONE, RESID, SWIGLU, GELU, GEGLU}.  Not a reference feature (the reference computes in bf16). */
int emu_linear_fp8_bf16(const void* A8, const float* a_scale, const void* W8, const float* w_scale, const void* bias,
                        const void* res, void* C, int M, int N, int K, int lda, int ldw, int ldres, int ldc, int epi,
                        emu_stream_t s);
int emu_linear_fp8w_bf16(const void* A, const void* W8, const float* wscale, const void* bias, const void* res,
                         const void* norm_w, void* C, int M, int N, int K, int lda, int ldw, int ldres, int ldc,
                         float eps, int epi, emu_stream_t s);
/* emu_linear_bf16 (M > 8, epi in {NONE, RESID, GEGLU}) with the fused epilogues of the UNet transformer blocks -- what
 * diffusers' BasicTransformerBlock computes as LayerNorm -> Linear (norm1/2/3 ahead of attn1.to_q/k/v, attn2.to_q,
 * ff.net.0.proj; Emu2/emu/diffusion.py:136-141), without a LayerNorm launch in between:
 *   row_stats_out  producer side: besides C, per-row partial (sum, sum of squares) of the bf16 outputs, one fp32 pair per
 *                  128-column slot: row_stats_out[(slot * M + m) * 2 + {0,1}], slot = n / 128   (N % 128 == 0)
 *   ln_c/ln_d/ln_stats/ln_slots/ln_eps  consumer side: A is the UN-normalised activation, W = W0 * gamma, and
 *                  C[m,n] = epi(bf16(rstd_m * (sum_k A[m,k] W[n,k] - mean_m * ln_c[n]) + ln_d[n])) with mean / rstd of row m
 *                  from its ln_slots (= K / 128 <= 10) partial pairs, ln_c[n] = sum_k W[n,k], ln_d[n] = sum_k W0[n,k] beta[k] + bias[n]
 *   vt_out/vt_col0/vt_s/vt_spad  output columns n >= vt_col0 (the V heads of a fused qkv projection) are stored
 *                  key-contiguous, vt_out[(b * (N - vt_col0) + n - vt_col0) * vt_spad + s] for row m = b * vt_s + s,
 *                  instead of row-major (what the attention kernel's P.V MFMA reads; replaces a transpose launch)
 *   cross_k/cross_vt/...  this GEMM is attn2.to_q (head dim 64, N = heads * 64, no bias) and its epilogue runs the whole
 *                  cross-attention over the <= 64 cached context tokens: C receives softmax(q K^T * cross_scale) V per head
 *                  (what the attention kernel would have written from q), q itself is never stored.  cross_k [B * cross_n,
 *                  cross_ldk] K rows (head h at columns h * 64), cross_vt [B, heads, 64, cross_npad] V^T (zero beyond cross_n),
 *                  batch element of row m = m / cross_rows (cross_rows % 64 == 0).  Combines with ln_* only.
 * All pointers optional (NULL = feature off). */
typedef struct {
    float* row_stats_out;
    const float* ln_c;
    const float* ln_d;
    const float* ln_stats;
    int ln_slots;
    float ln_eps;
    void* vt_out;
    int vt_col0, vt_s, vt_spad;
    const void* cross_k;
    const void* cross_vt;
    int cross_ldk, cross_n, cross_npad, cross_rows;
    float cross_scale;
} emu_linear_fx;
int emu_linear_fused_bf16(const void* A, const void* W, const void* bias, const void* res, void* C, int M, int N, int K,
                          int lda, int ldw, int ldres, int ldc, int epi, const emu_linear_fx* fx, emu_stream_t s);

/* LlamaRMSNorm (transformers; emu.py:133-138): y = bf16(w * bf16(x * rsqrt(mean(x^2) + eps))) */
int emu_rmsnorm_bf16(const void* x, const void* w, void* y, int rows, int cols, int ldx, int ldy, float eps,
                     emu_stream_t s);
/* nn.LayerNorm + post-norm residual (eva_vit.py:298-300): y = (res ? res + : ) bf16(LN(x) * w + b) */
int emu_layernorm_bf16(const void* x, const void* w, const void* b, const void* res, void* y, int rows,
                       int cols, float eps, emu_stream_t s);
/* The same in one pass with the per-row e4m3 quantisation of the bf16 result (emu_quantize_fp8_rows' definition: scale = amax / 448,
 * 1 for a zero row), bit-identical to the two launches: q [rows, cols] bytes, scale [rows]; y may be null; cols <= 2048.  The
 * LayerNorm in front of a GEMM of the W8A8 modes (emu_vit_use_fp8, emu_unet_use_fp8) hands over the GEMM's fp8 operand itself. */
int emu_layernorm_q8_bf16(const void* x, const void* w, const void* b, const void* res, void* y, void* q, float* scale, int rows,
                          int cols, float eps, emu_stream_t s);
/* in-place row softmax x = bf16(softmax(x * scale [+ bias])) over [rows, cols] (row stride ld): materialised-score
 * attention for the VAE mid block (AutoencoderKL attention, 1 head of 512; Emu2/emu/diffusion.py:216) and, with the additive
 * bf16 bias [rows, cols] (relative-position bias + causal mask), T5Attention of Emu1's CausalFormer
 * (Emu1/models/modeling_t5.py:629-666) */
int emu_softmax_rows_bf16(void* x, const void* bias, int rows, int cols, int ld, int ld_bias, float scale, emu_stream_t s);
/* A chain of one-row projections  y_i = epi_i(W_i . norm_i(x_i))  in ONE persistent launch (emu_amd/csrc/decode_engine.hip: a loader
 * wave per CU streams every op's weights through an LDS ring ahead of the activations, op outputs cross the chip as 4-byte
 * granules): the LlamaDecoderLayer linears of a decode step (transformers, reached from Emu2/emu/emu.py:213-229) between two
 * all-reduces of a tensor-parallel shard.  x_i is either a bf16 vector written before the call (x_from_prev = 0) or the output of
 * op i - 1 (x_from_prev = 1; that op's output then exists only as granules in `granules`, caller-owned device memory of
 * emu_gemv_chain_granule_bytes(ops, nops) bytes, zeroed by this call).  Bit-identical to the same ops through emu_linear_bf16.
 * K <= 6656, N >= 2 x CUs, at most 6 ops; -95 = shape not covered (use the launches); err: device counter of bounded waits that gave
 * up (non-zero = invalid results).  Needs the device to itself (one resident workgroup per CU). */
typedef struct {
    const void* W; int N, K;
    const void* gain; float eps;       /* fused RMSNorm on the input vector, or NULL */
    int epi;                           /* EMU_EPI_NONE / EMU_EPI_RESID / EMU_EPI_SWIGLU */
    const void* res;                   /* [N] for EMU_EPI_RESID */
    const void* x;                     /* [K] input vector (x_from_prev = 0) */
    int x_from_prev;
    void* out;                         /* [N] (SWIGLU: N / 2) bf16, or NULL when the next op takes it (x_from_prev = 1 there) */
} emu_chain_op;
size_t emu_gemv_chain_granule_bytes(const emu_chain_op* ops, int nops);
int emu_gemv_chain_bf16(emu_ctx* ctx, const emu_chain_op* ops, int nops, void* granules, size_t granule_bytes, unsigned int* err,
                        emu_stream_t s);
/* Touch [ptr, ptr + bytes) -- the weight matrix of a launch that FOLLOWS on the stream -- into the 256 MB infinity cache with
 * `workgroups` x 256 threads, one dword per 128-byte line, nothing waited for.  The stand-alone form of the successor prefetch the
 * one-row decode step carries inside its own launches (emu_llama_set_decode_prefetch): the reference's nn.Linear weights
 * (transformers LlamaDecoderLayer reached from Emu2/emu/emu.py:213-229) are read once per token, and every kernel of a layer that
 * leaves HBM idle (attention, split merge, all-reduce) pulls the next projection's rows on-die meanwhile.  tools / tests. */
int emu_prefetch(const void* ptr, size_t bytes, int workgroups, emu_stream_t s);
/* embed_tokens (emu.py:119,193) and the masked row overwrite text_embeds[ids == IMAGE] = ... (emu.py:202-203) */
int emu_embed_gather_bf16(const int32_t* ids, const void* table, void* out, int n_tok, int hidden, int vocab,
                          emu_stream_t s);
int emu_scatter_rows_bf16(const void* src, const int32_t* dst_rows, void* out, int n_rows, int hidden,
                          emu_stream_t s);
/* greedy next-token selection of lm.generate(num_beams=1, do_sample=False) (emu.py:213-229); suppress_id =
 * EOS while fewer than min_length tokens exist, else -1.  First index wins ties. */
int emu_argmax_bf16(const void* logits, int ld, int rows, int vocab, int suppress_id, int32_t* out,
                    emu_stream_t s);
/* PatchEmbed conv as im2col (eva_vit.py:327-335): image NCHW (fp32 or bf16) -> [B*g*g, Kpad] bf16 */
int emu_patchify(const void* image, int image_is_f32, void* out, int B, int C, int HW, int patch, int Kpad,
                 emu_stream_t s);
/* cat(cls, patches) + pos_embed (eva_vit.py:406-409) */
int emu_vit_assemble_bf16(const void* patches, const void* cls, const void* pos, void* x, int B, int T, int C,
                          emu_stream_t s);
/* encode_image pooling (emu.py:82-89): tokens [B, 1+g*g, C] -> [B, (g/s)^2, C] */
int emu_avgpool_tokens_bf16(const void* x, void* out, int B, int g, int C, int stride, emu_stream_t s);
/* apply_rotary_pos_emb on q,k in place + KV-cache append (transformers LlamaAttention; emu.py:213-229).
 * qkv rows [B*T, 3*H*D] (q | k | v); cos/sin tables [max_pos, D] bf16; caches [B, H, S_max, D]. */
int emu_rope_kv_append_bf16(void* qkv, const void* cos, const void* sin, const int32_t* pos,
                            const int32_t* slot, void* kcache, void* vcache, int B, int T, int H, int D,
                            int S_max, emu_stream_t s);
/* Vt[b,h,d,s] = V[b,h,s,d], zero padded to S_pad (multiple of 64) */
int emu_transpose_v_bf16(const void* v, long v_sb, long v_sh, long v_ss, void* vt, int B, int H, int S, int D,
                         int S_pad, emu_stream_t s);
/* softmax(Q K^T * scale + causal/left-pad mask) V, fused (LlamaAttention prefill; eva_vit.py:227-248;
 * diffusers attention).  D in {64, 128}.  kstart[b]: keys < kstart[b] masked (left padding), may be NULL. */
int emu_flash_attn_bf16(const void* q, long q_sb, long q_sh, long q_ss, const void* k, long k_sb, long k_sh,
                        long k_ss, const void* vt, void* o, long o_sb, long o_sh, long o_ss,
                        const int32_t* kstart, int B, int H, int Sq, int Sk, int Sk_pad, int D, int causal,
                        float scale, emu_stream_t s);
/* one-query attention over the KV cache (decode step).  ctx_dev (device int32, may be NULL) overrides ctx so a
 * captured graph can be replayed while the context grows; ctx_max sizes the launch. */
size_t emu_decode_attn_ws_bytes(int B, int H, int D, int ctx_max);
int emu_decode_attn_bf16(const void* q, long q_sb, long q_sh, const void* kcache, const void* vcache, void* o,
                         long o_sb, long o_sh, const int32_t* kstart, const int32_t* ctx_dev, int ctx,
                         int ctx_max, void* ws, int B, int H, int D, int S_max, float scale, emu_stream_t s);

/* ---- LLaMA decoder engine ------------------------------------------------------------------------------
 * transformers LlamaModel / LlamaForCausalLM as used by EmuModel.generate / generate_image
 * (emu.py:133-138, 213-229).  Weights are passed as raw pointers to PACKED tensors (see emu_amd/llama.py):
 *   wqkv  [3*Hl*D, hidden]   rows = q heads | k heads | v heads of this TP shard
 *   wo    [hidden, Hl*D]
 *   wgu   [2*Fl, hidden]     gate/up interleaved: row 2j = gate_j, row 2j+1 = up_j
 *   wdown [hidden, Fl]
 * KV cache: [layers, B, Hl, S_max, D] for K and V. */
typedef struct {
    int hidden, heads_local, head_dim, ffn_local, layers, vocab, max_pos;
    float rms_eps;
} emu_llama_cfg;
int emu_llama_create(emu_ctx* ctx, const emu_llama_cfg* cfg, emu_llama** out);
void emu_llama_destroy(emu_llama* m);
int emu_llama_set_layer(emu_llama* m, int layer, const void* wqkv, const void* wo, const void* wgu,
                        const void* wdown, const void* ln1, const void* ln2);
/* Optional fp8 decode stream (not in the reference, which runs bf16 end to end -- Emu2/emu/chat.py:199-213; this is
 * the weight-only quantised serving mode of BASELINE.json configs[5]).  W8: OCP e4m3fn bytes in the SAME packed row
 * layout as the bf16 weight; scale: fp32 [rows], W ~= fp8 * scale[row].  emu_llama_use_fp8(m, 1): decode rows (B*T <= 2)
 * stream them, prefill keeps the bf16 weights; (m, 2): prefill rows (B*T > 16) run W8A8 GEMMs as well -- activations
 * quantised per row ahead of every GEMM, emu_linear_fp8_bf16 -- when hidden, heads_local*head_dim and ffn_local are
 * multiples of 128 (BASELINE.json configs[4], "fp8 MFMA weights"); (m, 0): bf16.  Fails if the fp8 tensors were not registered. */
int emu_llama_set_layer_fp8(emu_llama* m, int layer, const void* wqkv8, const float* sqkv, const void* wo8,
                            const float* so, const void* wgu8, const float* sgu, const void* wdown8,
                            const float* sdown);
int emu_llama_set_head_fp8(emu_llama* m, const void* lm_head8, const float* lm_scale);
int emu_llama_use_fp8(emu_llama* m, int enable);
/* Vocabulary-sharded lm_head under tensor parallelism (the reference's lm_head is lm.lm_head of LlamaForCausalLM reached
 * from Emu2/emu/emu.py:213-229): the lm_head pointer of emu_llama_set_head holds rows [row0, row0 + rows) of the vocabulary only;
 * emu_llama_logits fills this rank's columns of the caller's [M, vocab] rows, zeroes the others and all-reduces the rows, so every
 * rank ends with the full, bit-identical logits.  rows < 0 = whole matrix (the default). */
int emu_llama_set_head_shard(emu_llama* m, int row0, int rows);
int emu_llama_set_head(emu_llama* m, const void* final_norm, const void* lm_head, const void* embed,
                       const void* rope_cos, const void* rope_sin);
/* kcache == vcache == NULL detaches the caches (the caller freed them): emu_llama_forward answers -22 until new ones are set. */
int emu_llama_set_kv(emu_llama* m, void* kcache, void* vcache, int batch, int s_max);
/* Beam search (the reference's default decoding mode: lm.generate(num_beams=5), Emu2/emu/emu.py:163-172,213-229; transformers
 * replicates the prompt's cache per beam with expand_inputs / _reorder_cache).  Here cache rows come in groups of `beams`
 * consecutive rows of one prompt whose first `shared_slots` slots (the prompt) are stored ONCE, in the group's first row:
 * single-token steps read them from there (the beams' workgroups for one 128-key split run on one XCD, whose L2 then
 * serves all but the first), and only slots >= shared_slots are per-row.  beams <= 1 switches it off
 * (emu_llama_set_kv does too).  -22 for beams > 8, a batch the group size does not divide, or slots beyond the capacity. */
int emu_llama_set_kv_share(emu_llama* m, int beams, int shared_slots);
/* Parity hook: emu_llama_forward runs decoder layers [l0, l1) only (l1 < 0: through the last; default 0, -1 = all), each on the
 * KV-cache plane of its own index.  The full-size tests feed ONE layer the fp32 oracle's input of that layer (a common input per
 * layer instead of 60 layers of accumulated bf16 rounding); production code never calls it. */
int emu_llama_set_layer_range(emu_llama* m, int l0, int l1);

/* Prefill fusion (off by default), a PER-CALL promise: a caller that sets it promises that its NEXT emu_llama_forward call with
 * T > 1 rows passes slot[i] = i (the rows of the one batch element are the whole context, in order -- what EmuModel's prefill
 * does); that call consumes the promise, so any other caller of emu_llama_forward (rows in another slot order) gets the
 * unfused sequence, which honours slot[] everywhere, unless it renews the promise itself.  Then, for
 * B = 1, T == ctx, head_dim 128 and heads_local * 128 a multiple of 256, the qkv projection applies RoPE to q and k, appends
 * k / v to the cache and writes V^T for the attention kernel from its own epilogue (same arithmetic and rounding points as the
 * three launches it replaces: bit-identical hidden states and caches); every other call runs the unfused sequence. */
int emu_llama_set_prefill_fusion(emu_llama* m, int enable);

/* Tensor-parallel prefill with the all-reduces overlapped (the reference has no tensor parallelism: its multi-GPU scheme
 * is layer placement, Emu2/emu/mixin.py:44-81, Emu2/emu/chat.py:250-283).  min_rows > 0: a B = 1, T == ctx forward under the
 * slot-order promise (emu_llama_set_prefill_fusion) with at least min_rows rows (clamped up to 512) on a context with tp_size > 1
 * (or a 1-rank communicator / comm block) cuts the prompt's rows once, at a multiple of 256, into two halves that run as two
 * CONCURRENT LANES: causal attention makes the first half independent of the second, and the second needs only the first's keys /
 * values of the same layer, so each half walks all layers as its own chain (norm, qkv, attention, o_proj, all-reduce, norm, gate/up,
 * down, all-reduce) on its own stream -- the first on the caller's, the second on a stream owned by the context -- with one event
 * per layer (the first half's K / V rows are in the cache) the only edge between them.  While one lane's partial sums are on the
 * wire, the other lane's GEMMs have the CUs.  The caller's stream has joined the second one when the call returns (the whole
 * forward is capturable into one hipGraph).  All-reduces run inside their lane's stream where RCCL takes them; without a
 * communicator (rank processes sharing a device) every peer-to-peer all-reduce additionally waits for the one issued before it.
 * Same kernels and rounding points as the serial schedule; results agree to bf16 rounding, not bit for bit (a half-size launch may
 * take another tile configuration than the whole prompt's).  Cost on ONE GPU with a no-op all-reduce (what the cut itself costs):
 * +17 % (TP = 8 shard) ... +21 % (TP = 2) per prefill under graph replay; what it hides -- 2 x layers all-reduces of [S, hidden] --
 * needs a multi-GPU node to be measured.  min_rows = 0 keeps every all-reduce on the caller's stream between the GEMMs.  The first
 * call with min_rows > 0 creates the stream and events (never inside a forward); the workspace grows by the second lane's V^T and
 * K-slice scratch (emu_llama_workspace_bytes follows the setting). */
int emu_llama_set_tp_overlap(emu_llama* m, int min_rows);
/* forwards that took the two-half schedule since the engine was created (tests, tools) */
long emu_llama_tp_overlap_count(const emu_llama* m);

/* Decode attention in ONE launch (OFF by default): the last split workgroup of a head to arrive merges the head's splits itself
 * (agent-scope stores / loads of the split states and a relaxed arrival counter: no fence, no spinning) instead of a second,
 * combine launch; same arithmetic, bit-identical outputs.  Measured 0.4 % slower than the two launches on MI355X (three
 * dependent fabric round trips in the tail cost what the launch does), so it stays an option for A/B timing. */
int emu_llama_set_decode_tail(emu_llama* m, int enable);

/* Whole decoder layers of a one-row decode step in ONE launch (csrc/decode_layer.hip): with enable != 0, emu_llama_forward calls
 * with B == 1, T == 1, bf16 weights, head_dim 128 and no shared-prefix KV rows run layers_per_launch layers (0 = all layers of the
 * call) per launch instead of six launches (+ two all-reduces) per layer.  The workgroups of the launches it replaces become roles
 * of one grid that wait on arrival counters with their weight slice already requested; results are bit-identical to the launches
 * (same rows per workgroup and summation order).  Under tensor parallelism, enable == 1 cuts every layer at its two all-reduces
 * ([q, attention, o_proj] | emu_allreduce_bf16 | [gate/up, down] | emu_allreduce_bf16: four launches instead of eight), and
 * enable == 2 runs the all-reduces INSIDE the launch over the P2P comm blocks (emu_tp_p2p_enable must be on, else as 1): the
 * workgroup that completes o_proj / down_proj runs the peer exchange while the consumers' weight slices are already in flight.
 * enable == 3 (tensor parallelism with the P2P path on; otherwise as 1): the two RMSNorm-fronted weight streams stay stand-alone
 * launches, the attention (its split merge done by the head's last split) and the two row-sharded projections run as single-role
 * launches whose LAST workgroup to arrive runs the all-reduce in its tail -- 5 launches per layer instead of 8, and nothing waits
 * inside a launch except that one workgroup for its peers, so this mode is safe for rank processes sharing a device.
 * Mode 2 needs every rank on its own GPU (a launch that waits for a peer holds its CUs), and every fused mode needs the DEVICE TO
 * ITSELF (modes 1 and 2; also mode 1's cut layers wait inside their launches): processes that share a GPU can starve each other's producer workgroups of CU slots (observed with eight rank processes on
 * one device: a time-out, then garbage and a non-zero give-up count).  Every in-kernel wait is bounded in wall-clock time (2 s, or
 * the peer-to-peer time-out where longer).
 * enable == 4: the persistent weight-streaming engine (emu_amd/csrc/decode_engine.hip; see emu_gemv_chain_bf16): per layer the
 * attention launches, then ONE launch  o_proj -> all-reduce -> RMSNorm + gate/up (SwiGLU) -> down -> all-reduce -> RMSNorm + the next
 * layer's qkv projection, whose loader waves stream the layer's weights through an LDS ring ahead of the four hand-offs and whose
 * all-reduces run over the ranks' comm arrays (tags numbered by a per-CU device counter).  Tensor-parallel shards whose rows are at
 * most 13 KiB (hidden, heads_local * 128, ffn_local <= 6656: TP >= 4 at the LLaMA-33B shape) with the fence-free P2P path on; anything
 * else keeps the launches.  Same bits as the launches; needs every rank on its own GPU and the device to itself.
 * The first fused forward after a weight change uploads a pointer table and must therefore run outside stream capture (-16).
 * Replaces: the per-layer module calls of the LlamaDecoderLayer loop (Emu2/emu/emu.py:133-138, :213-229) and the device hops of
 * Emu2/emu/mixin.py:44-81. */
int emu_llama_set_decode_fused(emu_llama* m, int enable, int layers_per_launch);
/* giveups: in-kernel waits that ran into their time limit since creation (non-zero: outputs are garbage, treat as an error);
 * forwards: emu_llama_forward calls that took the fused path.  Either pointer may be NULL.  Synchronises the device. */
int emu_llama_decode_fused_stats(emu_llama* m, unsigned int* giveups, long* forwards);
/* Tools hook (tools/decode_trace.py): device buffer of 4 x u64 per workgroup of the LAST fused launch of a forward -- {role | layer << 8,
 * entry, input ready (0: no wait), exit} in 100 MHz ticks; written only by a library built with -DEMU_TRACE; NULL = off. */
int emu_llama_set_decode_trace(emu_llama* m, void* buf);
size_t emu_llama_workspace_bytes(const emu_llama* m, int B, int T);
/* all decoder layers over B*T rows (T > 1: prefill with MFMA GEMMs + flash attention; T == 1: decode with
 * weight-streaming GEMVs).  hidden [B*T, hidden] is the residual stream, updated in place (NOT final-normed).
 * pos/slot: device int32 [B*T] (RoPE position / cache slot per row); kstart: device int32 [B] or NULL;
 * ctx_dev: device int32 [1] = number of valid cache slots AFTER this call (decode only, may be NULL);
 * ctx: the same value on the host (exact for prefill; upper bound used when ctx_dev is given). */
int emu_llama_forward(emu_llama* m, void* hidden, int B, int T, const int32_t* pos, const int32_t* slot,
                      const int32_t* kstart, const int32_t* ctx_dev, int ctx, void* workspace, size_t ws_bytes,
                      emu_stream_t s);
/* final RMSNorm (+ lm_head): rows [M, hidden] -> normed [M, hidden] / logits [M, ld] */
int emu_llama_final_norm(emu_llama* m, const void* hidden, void* out, int rows, emu_stream_t s);
int emu_llama_logits(emu_llama* m, const void* hidden, int ldh, int M, void* logits, int ld, void* workspace,
                     size_t ws_bytes, emu_stream_t s);
/* one greedy decode step entirely on the device (graph-capturable): embed(cur_ids) -> layers -> logits ->
 * argmax -> out_ids[step_dev[0], :] = cur_ids = next; pos/slot/ctx/step advance by one.
 * state: device int32 arrays cur_ids[B], pos[B], slot[B], ctx[1], step[1]; out_ids [max_new, B]. */
int emu_llama_greedy_step(emu_llama* m, int B, int32_t* cur_ids, int32_t* pos, int32_t* slot,
                          const int32_t* kstart, int32_t* ctx_dev, int32_t* step_dev, int32_t* out_ids,
                          int ctx_upper, void* hidden, void* logits, int ld_logits, void* workspace,
                          size_t ws_bytes, emu_stream_t s);

/* One step of transformers' beam search (lm.generate(num_beams=N), Emu2/emu/emu.py:163-172,213-229: the library's BeamSearchScorer
 * logic in its vectorised form), deterministic mode, in two launches: log-softmax of the N beams' logit rows of every prompt, the 2N
 * best continuations over beams x vocabulary of (log p + running score) with EOS masked while cur < min_len, then the bookkeeping --
 * the N best non-finished candidates keep running; candidates among the first N that hit EOS or the length limit L compete with the
 * kept results at score / len ** length_penalty; the early-stopping heuristic (early_stopping=False) clears heuristic_open[b] when
 * no continuation can beat the worst kept result any more.  hf431 selects the scorer conventions: 0 = transformers 5.x (the library
 * installed beside this repo, which the golden fixtures pin: every ending hypothesis is scored over cur + 1 tokens, the heuristic
 * looks at the best running beam); 1 = transformers 4.31, the release the reference pins (Emu2/requirements.txt:2;
 * BeamHypotheses.add divides an EOS hypothesis by the cur tokens ahead of the EOS, is_done looks at the best of all 2N candidates,
 * finalize adds the running beams at the length limit unless the prompt is done) -- restated from that release, not runnable here.
 * cur: tokens generated so far; cur_dev (device int32, may be NULL) overrides it so that a captured hipGraph replays the step for
 * every token (steps with cur >= L do nothing).
 * logits row of (prompt b, beam j) = logits + b * ld_prompt + j * ld_beam elements (step 0: ld_beam = 0).  State, updated in
 * place (device): running_seq / sequences int32 [B, nb, L], running_scores / beam_scores f32 [B, nb] (initial: running (0, -1e9, ...),
 * kept -1e9), finished u8 [B, nb], seq_len int32 [B, nb], heuristic_open u8 [B] (initial 1).  Out: next_tok int32 [B * nb] (the tokens
 * to feed), beam_flat int64 [B * nb] (cache row b * nb + beam each of them continues).  nb <= 8, L <= 256, cur < L.
 * workspace: emu_beam_step_workspace_bytes(B, nb, V) of device scratch (per-chunk partial results of the first of the two launches). */
size_t emu_beam_step_workspace_bytes(int B, int nb, int V);
int emu_beam_step_bf16(const void* logits, long ld_prompt, long ld_beam, int V, int B, int nb, int L, int cur, const int32_t* cur_dev,
                       int min_len, int eos_id, float length_penalty, int hf431, int32_t* running_seq, int32_t* sequences,
                       float* running_scores, float* beam_scores, unsigned char* finished, int32_t* seq_len,
                       unsigned char* heuristic_open, int32_t* next_tok, long* beam_flat, void* workspace, size_t ws_bytes,
                       emu_stream_t s);

/* Loop state of EmuModel.generate_image (emu.py:92-153, KV-cached form: step j feeds project_up(project_down(h_{j-1}))) on the
 * device, so that {project_up, decoder step, final norm, project_down, this} is captured once and replayed n_query - 1 times:
 * out_all[step_dev[0]] = prev = src ([B, cols] bf16: the step's visual embedding), pos[b]++, slot[b]++, step_dev[0]++. */
int emu_regress_advance_bf16(const void* src, void* out_all, void* prev, int32_t* pos, int32_t* slot, int32_t* step_dev, int B,
                             int cols, emu_stream_t s);

/* The host work of a beam step moved to the device, so that {advance, re-order, embed, decoder step, logits, emu_beam_step_bf16} can
 * be captured once and replayed for every token (the reference's default decoding mode must not run at the speed of the host):
 * emu_beam_advance phase 0: slot[i] = slot0 + cur - 1, pos[i] = pos0[i] + cur - 1 for the step that feeds token cur - 1 (cur read
 *   from cur_dev; rows = prompts x beams); phase 1: cur_dev += 1.  Nothing happens once cur >= L.
 * emu_llama_beam_reorder_kv: transformers' _reorder_cache for the generated slots [slot0, slot0 + cur - 1) of the engine's current
 *   cache (rows in groups of `beams`; the shared prompt slots never move): row r takes the slots of row beam_flat[r], in place. */
int emu_beam_advance(int32_t* cur_dev, int32_t* pos, int32_t* slot, const int32_t* pos0, int slot0, int rows, int L, int phase,
                     emu_stream_t s);
int emu_llama_beam_reorder_kv(emu_llama* m, const long* beam_flat, const int32_t* cur_dev, int beams, int slot0, int L,
                              emu_stream_t s);

/* ---- EVA-CLIP ViT engine -------------------------------------------------------------------------------
 * EVAVisionTransformer.forward_features (eva_vit.py:402-431), post-norm blocks (:296-300), naive attention
 * (:182-252) with heads zero-padded from head_width to 128 at pack time (see emu_amd/vit.py):
 *   wqkv [3*Hh*128, C] (+ bqkv [3*Hh*128] = q_bias | 0 | v_bias), wproj [C, Hh*128], fc1 [F, C], fc2 [C, F]. */
typedef struct {
    int image_size, patch_size, width, layers, heads, head_width, mlp_hidden, kpad;
    float ln_eps;
    int prenorm;   /* 0: Emu2 post-norm blocks (eva_vit.py:298-300); 1: Emu1 EVA-CLIP-g pre-norm blocks
                      (Emu1/models/eva_vit_model.py:409-416: x += attn(LN1(x)); x += mlp(LN2(x))) */
} emu_vit_cfg;
int emu_vit_create(emu_ctx* ctx, const emu_vit_cfg* cfg, emu_vit** out);
void emu_vit_destroy(emu_vit* m);
int emu_vit_set_stem(emu_vit* m, const void* wpatch, const void* bpatch, const void* cls, const void* pos);
int emu_vit_set_block(emu_vit* m, int layer, const void* wqkv, const void* bqkv, const void* wproj,
                      const void* bproj, const void* ln1w, const void* ln1b, const void* fc1w, const void* fc1b,
                      const void* fc2w, const void* fc2b, const void* ln2w, const void* ln2b);
/* Optional W8A8 mode of the encoder blocks (not a reference feature -- the reference runs bf16 end to end; BASELINE.json
 * configs[4] names an fp8 MFMA path): e4m3 copies of the four matrices of a block (emu_quantize_fp8_rows of the PACKED bf16
 * matrices registered with emu_vit_set_block: bytes [N, K] + one fp32 scale per output row).  With emu_vit_use_fp8(m, 1) every
 * GEMM of a block quantises its activation rows (per-row e4m3) and runs on emu_linear_fp8_bf16's kernels; LayerNorm, attention
 * and the stem stay bf16.  width and mlp_hidden must be multiples of 128.  Registering changes emu_vit_workspace_bytes. */
int emu_vit_set_block_fp8(emu_vit* m, int layer, const void* wqkv8, const float* sqkv, const void* wproj8, const float* sproj,
                          const void* fc1w8, const float* sfc1, const void* fc2w8, const float* sfc2);
int emu_vit_use_fp8(emu_vit* m, int enable);
/* Launch fusions of the encoder blocks (all on by default; 0 = the launch sequence of rounds 1-3, for A/B timing and parity):
 * bit 0: with one image the V heads leave the qkv projection key-contiguous (no transpose launch); bit 1: the K-slice sum of a
 * post-norm block's fc2 applies bias, LayerNorm and the residual add row-wise in the same launch (no reduce + layernorm launches).
 * Bit-identical tokens either way. */
int emu_vit_set_fusion(emu_vit* m, int mask);
size_t emu_vit_workspace_bytes(const emu_vit* m, int B);
/* image NCHW (fp32 or bf16) -> tokens [B, 1+g*g, C] bf16 (raw block output incl. cls, eva_vit.py:433-445) */
int emu_vit_forward(emu_vit* m, const void* image, int image_is_f32, int B, void* out_tokens, void* workspace,
                    size_t ws_bytes, emu_stream_t s);

/* Parity hook: blocks [l0, l1) of the encoder in place on tokens [B, 1+g*g, C] bf16 (what emu_vit_forward runs after the stem
 * for l0 = 0, l1 = layers); the full-size tests compare every block on a common input.  Same workspace as emu_vit_forward. */
int emu_vit_blocks(emu_vit* m, void* tokens, int B, int l0, int l1, void* workspace, size_t ws_bytes, emu_stream_t s);

/* ---- SDXL-style UNet denoise engine ---------------------------------------------------------------------
 * One call = one iteration of EmuVisualGeneration's denoising loop (Emu2/emu/diffusion.py:130-149):
 * cat([latents]*2) -> scheduler.scale_model_input -> UNet2DConditionModel (conf/diffusion_config/unet/config.json)
 * -> chunk(cond, uncond) -> guidance -> EulerDiscreteScheduler.step (conf/.../scheduler_config.json).
 * Weights are registered by PACKED name (see emu_amd/unet.py: conv weights [Cout, 3,3,Cin], fused attn1 qkv / attn2 kv,
 * interleaved GEGLU rows, all time_emb_proj concatenated into temb_proj_all).  Parity of this path is UNPINNED
 * (diffusers is not available to run; see oracle/unet_ref.py). */
/* GroupNorm(groups, eps)(+SiLU) over an NHWC activation [B, HW, C] (diffusers ResnetBlock2D / Transformer2DModel norms) */
size_t emu_groupnorm_ws_bytes(int B, int HW, int C);
int emu_groupnorm_nhwc_bf16(const void* x, const void* gamma, const void* beta, void* y, void* ws, int B, int HW, int C,
                            int groups, float eps, int silu, emu_stream_t s);
/* 3x3 convolution, padding 1, as an implicit GEMM over NHWC input [B,Hin,Win,Cin] (Cin % 64 == 0), weights
 * [Cout, 3, 3, Cin].  mode 1: stride 1; 2: stride 2 (Downsample2D); 3: nearest x2 upsample fused (Upsample2D).
 * y = bf16(res + bf16(bf16(conv + bias) + bias2[b]))  (bias2: per-batch rows, e.g. the time embedding projection) */
int emu_conv3x3_nhwc_bf16(const void* x, const void* w, const void* bias, const void* bias2, int ld_bias2, const void* res,
                          void* y, int B, int Hin, int Win, int Cin, int Cout, int mode, emu_stream_t s);

typedef struct emu_unet emu_unet;
typedef struct {
    int in_ch, out_ch;
    int ch[3];
    int layers_per_block;
    int depth[3], heads[3], attn[3];
    int cross_dim, groups;
    float gn_eps;
    int temb_dim, kpad_in;
} emu_unet_cfg;
int emu_unet_create(emu_ctx* ctx, const emu_unet_cfg* cfg, emu_unet** out);
void emu_unet_destroy(emu_unet* u);
int emu_unet_set_weight(emu_unet* u, const char* name, const void* ptr);
int emu_unet_finalize(emu_unet* u);                       /* -2 + emu_last_error: first missing tensor        */
/* Launch fusions of the transformer blocks (default: every one whose packed tensors are registered).  Bit 0: LayerNorm folded
 * into the consumer GEMM (needs "<block>attn1.qkv.wln/.c/.d", "attn2.q.wln/.c/.d", "ff.geglu.wln/.c/.d": W * gamma, its fp32
 * row sums, W @ beta + bias); bit 1: V^T written by the qkv projection's epilogue (no transpose launch); bit 2: the cross-attention
 * over the (<= 64) prompt tokens runs inside the attn2.to_q projection's epilogue (no q tensor, no attention launch).  0 = the
 * unfused launch sequence (A/B timing, parity of fused vs unfused).  Returns the mask in effect. */
int emu_unet_set_fusion(emu_unet* u, int mask);
/* Optional W8A8 mode of the transformer blocks (not a reference feature; BASELINE.json configs[4] names an fp8 MFMA path).
 * Register every block's six packed matrices (attn1.qkv.w, attn1.out.w, attn2.q.w, attn2.out.w, ff.geglu.w, ff.out.w) a second
 * time with emu_unet_set_weight as "<name>.fp8" (the e4m3 bytes of emu_quantize_fp8_rows) and "<name>.fp8s" (fp32 row
 * scales), then emu_unet_use_fp8(u, 1): the blocks' GEMMs run on emu_linear_fp8_bf16's kernels, the three LayerNorms of a
 * block emit the consumer's fp8 rows themselves, the attention outputs and the GEGLU product are quantised per row by a launch
 * of their own; convs, GroupNorm, proj_in / proj_out, attention and the scheduler stay bf16.  Of emu_unet_set_fusion's bits the
 * V^T stores (1) and the cross-attention epilogue (2) stay in effect -- they work on finished sums -- the LayerNorm fold (0) has
 * no fp8 form.  Changes emu_unet_workspace_bytes; returns -2 when a copy is missing. */
int emu_unet_use_fp8(emu_unet* u, int enable);
int emu_unet_temb_total(const emu_unet* u);               /* rows of temb_proj_all (sum of resnet out channels) */
size_t emu_unet_workspace_bytes(const emu_unet* u, int H, int W);
size_t emu_unet_context_bytes(const emu_unet* u, int n_ctx);
/* once per prompt: cross-attention K / V^T of every block from ctx_tokens [2, n_ctx, cross] (cond FIRST,
 * diffusion.py:202,210) and the text_time embedding from add_in [2, add_dim] = cat(mean(prompt), Timesteps(time_ids)) */
int emu_unet_set_context(emu_unet* u, const void* ctx_tokens, int n_ctx, const void* add_in, int add_dim, void* cache,
                         size_t cache_bytes, void* workspace, size_t ws_bytes, emu_stream_t s);
/* one denoise step in place on latents NCHW [1, 4, H, W] bf16.  temb_table [steps, ch0] bf16 = Timesteps(t_i);
 * sigmas [steps+1] fp32; step_dev: device int32 step index, incremented by the call (graph replay advances it). */
int emu_unet_step(emu_unet* u, void* latents, int H, int W, const void* temb_table, const void* sigmas, int32_t* step_dev,
                  float guidance, void* workspace, size_t ws_bytes, emu_stream_t s);
/* bare UNet forward (parity tests): eps_out NHWC [2*H*W, 4] for cat([latents]*2) / sqrt(sigmas[step]^2 + 1) */
int emu_unet_forward(emu_unet* u, const void* latents, int H, int W, const void* temb_table, const void* sigmas,
                     const int32_t* step_dev, void* eps_out, void* workspace, size_t ws_bytes, emu_stream_t s);
/* Classifier-free guidance split over a rank pair (Emu2/emu/diffusion.py:131-145 runs cat([latents] * 2) through the
 * UNet and chunk(2)s the prediction): half = 0 / 1 makes emu_unet_forward compute ONLY the cond / uncond row of the pair (batch 1;
 * eps_out is then [H*W, 4]) against the context emu_unet_set_context cached for both rows; -1 (default) restores the pair.  The
 * ranks exchange their halves (131 KB per step at 128 x 128 latents) and each applies emu_unet_cfg_euler_step -- guidance + Euler
 * update of emu_unet_step on eps_pair [2*H*W, 4] (cond first), advancing step_dev.  emu_unet_step refuses while a half is set. */
int emu_unet_set_cfg_half(emu_unet* u, int half);
int emu_unet_cfg_euler_step(emu_unet* u, const void* eps_pair, void* latents, int H, int W, const void* sigmas, int32_t* step_dev,
                            float guidance, emu_stream_t s);


#ifdef __cplusplus
}
#endif
#endif /* EMU_HIP_H */
