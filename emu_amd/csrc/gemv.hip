// Decode-shape (M <= 16 rows) weight streams:  out[m, n] = epilogue( sum_k xeff[m, k] * W[n, k] )
//
// HBM-bound: every weight byte is read exactly once per call (PMC: traffic / algorithmic = 1.0013) with non-temporal
// loads; the activations (a few KB) come from L1/L2.  The family, dispatched by launch_gemv():
//   gemv_kernel      1..8 rows.  One 256-thread workgroup owns R (16/8/4/2) consecutive weight rows and splits K across its
//                    4 waves (thread t owns 16-byte vectors t, t+256, ...), N/R workgroups >> 256 CUs.  Inner product on
//                    v_dot2c_f32_bf16 (weights consumed unpacked), wave reductions on DPP.  Optional fused prologue: LLaMA
//                    RMSNorm of x (fp32 variance, the reference's bf16 rounding points: xeff = bf16(g * bf16(x * rinv))).
//                    Fused epilogues: bias, residual add, SwiGLU over interleaved (gate, up) row pairs, SiLU, GELU.
//   gemv_rt_kernel   M = 1, single HBM round trip per workgroup (all loads issued up front): down_proj, small shards.
//   gemv_wave_kernel M = 1, K <= 2560: one wave per 4 whole rows, no LDS / barrier (TP-shard o_proj / down_proj).
//   gemv_mfma_kernel 9..16 rows on v_mfma_f32_16x16x32_bf16.
//   gemv_fp8_*       optional e4m3 weight stream (per-row scale).
//
// Replaces (reference call sites): LlamaDecoderLayer linears + RMSNorm reached from Emu2/emu/emu.py:133-138
// and :213-229 at S=1 (greedy: 1 row; beam search: num_beams rows), project_up/project_down emu.py:131,147.
// Algorithmic bytes per call = 2*N*K (fp8: N*K).
#include "common.h"
#include <type_traits>
#include "kernels.h"

namespace {

// PRE > 0: all (<= PRE) 16-byte weight chunks of this thread are requested BEFORE the RMSNorm prologue, so the HBM
// stream is already in flight while the block computes mean(x^2); requires K/8 <= 256*PRE.  PRE == 0: generic loop.
template <int R, int MB, bool NORM, int EPI, int PRE>
__global__ __launch_bounds__(256) void gemv_kernel(const GemvArgs a) {
    __shared__ float red[4][R * MB];
    __shared__ float fin[R * MB];
    __shared__ float scratch[4];
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int KV = a.K >> 3;                       // 16-byte vectors per row
    const int n0 = blockIdx.x * R;

    const bf16_t* wrow[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int n = n0 + r;
        n = n < a.N ? n : a.N - 1;                 // tail rows: clamp loads, mask stores
        wrow[r] = a.W + (size_t)n * a.ldw;
    }
    // PRE path (MB == 1): x first (L2 hits, returned first because loads complete in order), then the whole weight
    // slice of this thread, so HBM is streaming while the RMSNorm statistics are reduced.
    u32x4 pre[PRE > 0 ? PRE : 1][R];
    u32x4 xr[PRE > 0 ? PRE : 1];
    if constexpr (PRE > 0) {
        static_assert(PRE == 0 || MB == 1, "preload form is for the single-row decode case");
#pragma unroll
        for (int c = 0; c < PRE; ++c) {
            const int vi = tid + 256 * c;
            const u32x4 z = {0u, 0u, 0u, 0u};
            xr[c] = vi < KV ? ld16(a.x + vi * 8) : z;
        }
#pragma unroll
        for (int c = 0; c < PRE; ++c) {
            const int vi = tid + 256 * c;
            const int vc = vi < KV ? vi : KV - 1;  // clamped: x is zero there, so the product vanishes
#pragma unroll
            for (int r = 0; r < R; ++r) pre[c][r] = ld_stream(reinterpret_cast<const u32x4*>(wrow[r] + vc * 8));
        }
    }

    float rinv[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) rinv[m] = 1.f;
    // PRE < 0 ("head" form, MB == 1, fused norm, K <= 8192): the activation vector and the gain (<= 4 chunks per thread
    // each, L2 hits, returned first because loads complete in order) and the first trip of the weight stream are requested
    // before the RMSNorm statistics are reduced, and the reduction goes through a raw s_barrier (a __syncthreads drains
    // vmcnt, i.e. waits for the weights): HBM streams during the prologue instead of after it.  The normalised activations
    // are packed once into registers, so the remaining trips issue nothing but weight loads.
    u32x4 hx[4], hg[4];
    u32x4 hw[R];
    if constexpr (PRE < 0) {
        static_assert(PRE >= 0 || (MB == 1 && NORM), "head form: single row with the fused norm");
        __shared__ float ssp[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int vi = tid + 256 * c;
            const int vc = vi < KV ? vi : KV - 1;
            hx[c] = vi < KV ? ld16(a.x + vi * 8) : u32x4{0u, 0u, 0u, 0u};
            hg[c] = ld16(a.norm_w + vc * 8);
        }
#pragma unroll
        for (int r = 0; r < R; ++r)                                    // short rows (K < 2048): clamped, against a zero operand
            hw[r] = ld_stream(reinterpret_cast<const u32x4*>(wrow[r] + (tid < KV ? tid : KV - 1) * 8));
        float ss = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float f[8];
            unpack8(hx[c], f);
#pragma unroll
            for (int j = 0; j < 8; ++j) ss += f[j] * f[j];
        }
        ss = wave_sum(ss);
        if (lane == 0) *reinterpret_cast<volatile float*>(&ssp[wave]) = ss;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const volatile float* sp = ssp;
        rinv[0] = rsqrtf((sp[0] + sp[1] + sp[2] + sp[3]) / (float)a.K + a.eps);
#pragma unroll
        for (int c = 0; c < 4; ++c) {                                  // hx <- bf16(g * bf16(x * rinv)), the dot2 operand
            float xf[8], g[8];
            unpack8(hx[c], xf);
            unpack8(hg[c], g);
#pragma unroll
            for (int j = 0; j < 8; ++j) xf[j] = g[j] * bfround(xf[j] * rinv[0]);
            hx[c] = pack8(xf);
        }
    } else
    if constexpr (NORM) {
        float ss[MB];
#pragma unroll
        for (int m = 0; m < MB; ++m) ss[m] = 0.f;
        if constexpr (PRE > 0) {
#pragma unroll
            for (int c = 0; c < PRE; ++c) {
                float f[8];
                unpack8(xr[c], f);
#pragma unroll
                for (int j = 0; j < 8; ++j) ss[0] += f[j] * f[j];
            }
        } else
        for (int vi = tid; vi < KV; vi += 256) {
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                if (m < a.M) {
                    float f[8];
                    unpack8(ld16(a.x + (size_t)m * a.ldx + vi * 8), f);
#pragma unroll
                    for (int j = 0; j < 8; ++j) ss[m] += f[j] * f[j];
                }
            }
        }
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            const float t = block_sum<4>(ss[m], scratch);
            rinv[m] = rsqrtf(t / (float)a.K + a.eps);
        }
    }

    float acc[R][MB];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int m = 0; m < MB; ++m) acc[r][m] = 0.f;

    // The inner product runs on v_dot2c_f32_bf16: weights are consumed as they arrive (no unpack) against the activation
    // vector re-packed to bf16 pairs once per 16-byte column group -- exact, because the reference rounds the normalised
    // activations to bf16 anyway.  4 VALU ops per 8 weights instead of 16 (unpack + fma).
    auto consume = [&](int vi, const u32x4 (&wv)[R], const u32x4* xpre) {
        float g[8];
        if constexpr (NORM) unpack8(ld16(a.norm_w + vi * 8), g);
        u32x4 xp[MB];
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            if (m < a.M) {
                xp[m] = xpre ? *xpre : ld16(a.x + (size_t)m * a.ldx + vi * 8);
                if constexpr (NORM) {
                    float xf[8];
                    unpack8(xp[m], xf);
#pragma unroll
                    for (int j = 0; j < 8; ++j) xf[j] = g[j] * bfround(xf[j] * rinv[m]);
                    xp[m] = pack8(xf);                               // the pack is the reference's second rounding
                }
            } else {
                xp[m] = u32x4{0u, 0u, 0u, 0u};
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                float t = acc[r][m];
                t = bf16_dot2(wv[r].x, xp[m].x, t);
                t = bf16_dot2(wv[r].y, xp[m].y, t);
                t = bf16_dot2(wv[r].z, xp[m].z, t);
                t = bf16_dot2(wv[r].w, xp[m].w, t);
                acc[r][m] = t;
            }
        }
    };

    if constexpr (PRE < 0) {
        auto dots = [&](const u32x4 (&wv)[R], const u32x4& xp) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float t = acc[r][0];
                t = bf16_dot2(wv[r].x, xp.x, t);
                t = bf16_dot2(wv[r].y, xp.y, t);
                t = bf16_dot2(wv[r].z, xp.z, t);
                t = bf16_dot2(wv[r].w, xp.w, t);
                acc[r][0] = t;
            }
        };
        // trip 0 from the head registers; trips 1..3 stream behind it through two register sets (two trips in flight)
        u32x4 w1[R];
        auto fetch = [&](u32x4 (&wv)[R], int c) {
            const int vi = tid + 256 * c;
            if (vi < KV) {
#pragma unroll
                for (int r = 0; r < R; ++r) wv[r] = ld_stream(reinterpret_cast<const u32x4*>(wrow[r] + vi * 8));
            }
        };
        fetch(w1, 1);
        dots(hw, hx[0]);
        __builtin_amdgcn_sched_barrier(0);                             // a third register set would cost a wave of occupancy
        fetch(hw, 2);
        if (tid + 256 < KV) dots(w1, hx[1]);
        __builtin_amdgcn_sched_barrier(0);
        fetch(w1, 3);
        if (tid + 512 < KV) dots(hw, hx[2]);
        if (tid + 768 < KV) dots(w1, hx[3]);
    } else if constexpr (PRE > 0) {
#pragma unroll
        for (int c = 0; c < PRE; ++c) {
            const int vi = tid + 256 * c;
            if (vi < KV) consume(vi, pre[c], &xr[c]);
        }
    } else {
#pragma unroll 2
        for (int vi = tid; vi < KV; vi += 256) {
            u32x4 wv[R];
#pragma unroll
            for (int r = 0; r < R; ++r) wv[r] = ld_stream(reinterpret_cast<const u32x4*>(wrow[r] + vi * 8));
            consume(vi, wv, nullptr);
        }
    }

#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            const float v = wave_sum(acc[r][m]);
            if (lane == 0) red[wave][r * MB + m] = v;
        }
    __syncthreads();
    if (tid < R * MB) fin[tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
    __syncthreads();

    if constexpr (EPI == EPI_SWIGLU) {
        // rows (2j, 2j+1) = (gate_j, up_j); out[m, n0/2 + j] = bf16(bf16(silu(gate)) * up)
        if (tid < (R / 2) * MB) {
            const int j = tid / MB, m = tid % MB;
            const int n = n0 + 2 * j;
            if (m < a.M && n + 1 < a.N) {
                const float gt = bfround(fin[(2 * j) * MB + m] + (a.bias ? bf2f(a.bias[n]) : 0.f));
                const float up = bfround(fin[(2 * j + 1) * MB + m] + (a.bias ? bf2f(a.bias[n + 1]) : 0.f));
                const float act = bfround(silu(gt));
                a.out[(size_t)m * a.ldo + (n >> 1)] = f2bf(act * up);
            }
        }
    } else {
        if (tid < R * MB) {
            const int r = tid / MB, m = tid % MB;
            const int n = n0 + r;
            if (m < a.M && n < a.N) {
                float v = fin[tid];
                if (a.bias) v += bf2f(a.bias[n]);
                v = bfround(v);
                if constexpr (EPI == EPI_SILU) v = bfround(silu(v));
                if constexpr (EPI == EPI_GELU) v = bfround(gelu_erf(v));
                if constexpr (EPI == EPI_RESID) v = v + bf2f(a.res[(size_t)m * a.ldres + n]);
                a.out[(size_t)m * a.ldo + n] = f2bf(v);
            }
        }
    }
}

// Single-round-trip GEMV (decode, M = 1): every load of the block -- x, gain, and the block's WHOLE weight slice
// (R rows x KIT 256-lane trips, fully unrolled into registers) -- is issued before anything waits, so a block pays one
// HBM latency instead of one per loop trip plus one for the RMSNorm prologue.  The prologue's cross-wave reduction uses
// a raw s_barrier (a __syncthreads would drain vmcnt and serialise the weight loads behind it).  FP8: 8-byte loads of
// e4m3 weights with a per-row fp32 scale.  KIT = ceil(K / 8 / 256) rounded up to a built value.
template <int R, int KIT, bool NORM, int EPI, bool FP8>
__global__ __launch_bounds__(256) void gemv_rt_kernel(const GemvArgs a) {
    using WT = typename std::conditional<FP8, u32x2, u32x4>::type;
    __shared__ float red[4][R];
    __shared__ float fin[R];
    __shared__ float ssp[4];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int KV = a.K >> 3;
    const int n0 = blockIdx.x * R;
    constexpr int WB = FP8 ? 1 : 2;                                   // bytes per weight
    const unsigned char* W8 = reinterpret_cast<const unsigned char*>(a.W);

    u32x4 xv[KIT], gv[KIT];
#pragma unroll
    for (int it = 0; it < KIT; ++it) {
        const int vi = tid + 256 * it;
        xv[it] = u32x4{0u, 0u, 0u, 0u};
        gv[it] = u32x4{0u, 0u, 0u, 0u};
        if (vi < KV) {
            xv[it] = ld16(a.x + vi * 8);
            if constexpr (NORM) gv[it] = ld16(a.norm_w + vi * 8);
        }
    }
    WT wv[KIT][R];
#pragma unroll
    for (int it = 0; it < KIT; ++it) {
        const int vi = tid + 256 * it;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            int n = n0 + r;
            n = n < a.N ? n : a.N - 1;
            wv[it][r] = WT{};
            if (vi < KV)
                wv[it][r] = __builtin_nontemporal_load(
                    reinterpret_cast<const WT*>(W8 + ((size_t)n * a.ldw + (size_t)vi * 8) * WB));
        }
    }

    float rinv = 1.f;
    if constexpr (NORM) {
        float ss = 0.f;
#pragma unroll
        for (int it = 0; it < KIT; ++it) {
            float f[8];
            unpack8(xv[it], f);
#pragma unroll
            for (int j = 0; j < 8; ++j) ss += f[j] * f[j];
        }
        ss = wave_sum(ss);
        if (lane == 0) *reinterpret_cast<volatile float*>(&ssp[wave]) = ss;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const volatile float* sp = ssp;
        rinv = rsqrtf((sp[0] + sp[1] + sp[2] + sp[3]) / (float)a.K + a.eps);
    }

    float acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.f;
#pragma unroll
    for (int it = 0; it < KIT; ++it) {
        float xf[8];
        u32x4 xp = xv[it];                                           // bf16 pairs for the dot2 path
        if constexpr (NORM || FP8) unpack8(xv[it], xf);
        if constexpr (NORM) {
            float g[8];
            unpack8(gv[it], g);
#pragma unroll
            for (int j = 0; j < 8; ++j) xf[j] = g[j] * bfround(xf[j] * rinv);
            xp = pack8(xf);                                          // the reference's second rounding
            if constexpr (FP8) unpack8(xp, xf);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if constexpr (FP8) {
                f32x2_t a2 = {0.f, 0.f};
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const f32x2_t lo = __builtin_amdgcn_cvt_pk_f32_fp8(wv[it][r][q], false);
                    const f32x2_t hi = __builtin_amdgcn_cvt_pk_f32_fp8(wv[it][r][q], true);
                    a2 = __builtin_elementwise_fma(lo, f32x2_t{xf[4 * q], xf[4 * q + 1]}, a2);       // v_pk_fma_f32
                    a2 = __builtin_elementwise_fma(hi, f32x2_t{xf[4 * q + 2], xf[4 * q + 3]}, a2);
                }
                acc[r] += a2[0] + a2[1];
            } else {
                float t = acc[r];
                t = bf16_dot2(wv[it][r].x, xp.x, t);
                t = bf16_dot2(wv[it][r].y, xp.y, t);
                t = bf16_dot2(wv[it][r].z, xp.z, t);
                t = bf16_dot2(wv[it][r].w, xp.w, t);
                acc[r] = t;
            }
        }
    }

#pragma unroll
    for (int r = 0; r < R; ++r) {
        const float v = wave_sum(acc[r]);
        if (lane == 0) red[wave][r] = v;
    }
    __syncthreads();
    if (tid < R) {
        float t = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
        if constexpr (FP8) { const int n = n0 + tid; t *= a.wscale[n < a.N ? n : a.N - 1]; }
        fin[tid] = t;
    }
    __syncthreads();
    if constexpr (EPI == EPI_SWIGLU) {
        if (tid < R / 2) {
            const int n = n0 + 2 * tid;
            if (n + 1 < a.N) {
                const float gt = bfround(fin[2 * tid] + (a.bias ? bf2f(a.bias[n]) : 0.f));
                const float up = bfround(fin[2 * tid + 1] + (a.bias ? bf2f(a.bias[n + 1]) : 0.f));
                a.out[n >> 1] = f2bf(bfround(silu(gt)) * up);
            }
        }
    } else {
        if (tid < R) {
            const int n = n0 + tid;
            if (n < a.N) {
                float v = fin[tid];
                if (a.bias) v += bf2f(a.bias[n]);
                v = bfround(v);
                if constexpr (EPI == EPI_SILU) v = bfround(silu(v));
                if constexpr (EPI == EPI_GELU) v = bfround(gelu_erf(v));
                if constexpr (EPI == EPI_RESID) v = v + bf2f(a.res[n]);
                a.out[n] = f2bf(v);
            }
        }
    }
}

template <int R, int KIT, bool FP8>
int launch_rt(const GemvArgs& a, hipStream_t s) {
    const dim3 grid((a.N + R - 1) / R), block(256);
    const bool norm = a.norm_w != nullptr;
#define EMU_RT_CASE(E)                                                                                               \
    case E:                                                                                                          \
        if (norm) hipLaunchKernelGGL((gemv_rt_kernel<R, KIT, true, E, FP8>), grid, block, 0, s, a);                  \
        else hipLaunchKernelGGL((gemv_rt_kernel<R, KIT, false, E, FP8>), grid, block, 0, s, a);                      \
        break;
    switch (a.epi) {
        EMU_RT_CASE(EPI_NONE)
        EMU_RT_CASE(EPI_RESID)
        EMU_RT_CASE(EPI_SWIGLU)
        EMU_RT_CASE(EPI_SILU)
        EMU_RT_CASE(EPI_GELU)
        default: return -22;
    }
#undef EMU_RT_CASE
    EMU_CHECK_LAUNCH();
    return 0;
}

// Short rows (K <= 2560: the o_proj / down_proj of a tensor-parallel shard, K = 896 / 2240 at TP = 8): splitting such
// a row over 256 threads leaves most lanes idle and pays two barriers for a cross-wave sum.  Here every wave owns RW whole
// rows -- all loads up front, v_dot2c, one DPP wave reduction, lane 0 stores -- no LDS, no barrier: the launch is pure
// latency, so the shortest dependency chain wins.
template <int RW, int KITW, int EPI>
__global__ __launch_bounds__(256) void gemv_wave_kernel(const GemvArgs a) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int KV = a.K >> 3;
    const int n0 = (blockIdx.x * 4 + wave) * RW;
    u32x4 xv[KITW], wv[KITW][RW];
#pragma unroll
    for (int it = 0; it < KITW; ++it) {
        const int vi = lane + 64 * it;
        const int vc = vi < KV ? vi : KV - 1;                         // clamped; the activation is zeroed instead
        xv[it] = vi < KV ? ld16(a.x + vi * 8) : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            int n = n0 + r;
            n = n < a.N ? n : a.N - 1;
            wv[it][r] = ld_stream(reinterpret_cast<const u32x4*>(a.W + (size_t)n * a.ldw + vc * 8));
        }
    }
    float acc[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) acc[r] = 0.f;
#pragma unroll
    for (int it = 0; it < KITW; ++it)
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            float t = acc[r];
            t = bf16_dot2(wv[it][r].x, xv[it].x, t);
            t = bf16_dot2(wv[it][r].y, xv[it].y, t);
            t = bf16_dot2(wv[it][r].z, xv[it].z, t);
            t = bf16_dot2(wv[it][r].w, xv[it].w, t);
            acc[r] = t;
        }
#pragma unroll
    for (int r = 0; r < RW; ++r) acc[r] = wave_sum(acc[r]);
    if (lane == 0) {
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const int n = n0 + r;
            if (n < a.N) {
                float v = acc[r];
                if (a.bias) v += bf2f(a.bias[n]);
                v = bfround(v);
                if constexpr (EPI == EPI_RESID) v = v + bf2f(a.res[n]);
                a.out[n] = f2bf(v);
            }
        }
    }
}

template <int RW, int KITW>
int launch_wave(const GemvArgs& a, hipStream_t s) {
    const dim3 grid((a.N + 4 * RW - 1) / (4 * RW)), block(256);
    if (a.epi == EPI_RESID) hipLaunchKernelGGL((gemv_wave_kernel<RW, KITW, EPI_RESID>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((gemv_wave_kernel<RW, KITW, EPI_NONE>), grid, block, 0, s, a);
    EMU_CHECK_LAUNCH();
    return 0;
}

// returns 1 when the shape is not covered
int try_launch_wave(const GemvArgs& a, hipStream_t s) {
    if (a.M != 1 || a.norm_w || a.wscale || (a.epi != EPI_NONE && a.epi != EPI_RESID)) return 1;
    const int kitw = ((a.K >> 3) + 63) / 64;
    if (kitw > 5 || a.N < 1024) return 1;                            // K <= 2560; tiny N stays on the block kernels
    switch (kitw) {
        case 1: return launch_wave<4, 1>(a, s);
        case 2: return launch_wave<4, 2>(a, s);
        case 3: return launch_wave<4, 3>(a, s);
        case 4: return launch_wave<4, 4>(a, s);
        default: return launch_wave<4, 5>(a, s);
    }
}

// M = 1 dispatch onto the single-round-trip kernel; returns 1 when the shape is not covered.
int try_launch_rt(const GemvArgs& a, hipStream_t s) {
    if (a.M != 1) return 1;
    if (a.epi == EPI_SWIGLU && (a.N & 1)) return 1;
    const int kit = ((a.K >> 3) + 255) / 256;
    const bool f8 = a.wscale != nullptr;
    if (f8 && (a.epi == EPI_SILU || a.epi == EPI_GELU)) return 1;
    // measured (tools/kbench.py, profiles/r01_gemv_variants.log): one round trip wins for matrices small enough
    // that latency, not bandwidth, sets the time (TP shards, tiny models) and for long rows without the RMSNorm
    // prologue (down_proj: 6.3 vs 5.85 TB/s); big fused-norm matrices amortise the prologue better over 8 rows
    const size_t bytes = (size_t)a.N * a.K * (f8 ? 1 : 2);
    const bool small = bytes < ((size_t)(f8 ? 32 : 64) << 20);
    const bool long_rows = !f8 && kit >= 5 && a.norm_w == nullptr;
    if (!small && !long_rows) return 1;
    if (kit <= 4) return f8 ? launch_rt<8, 4, true>(a, s) : launch_rt<4, 4, false>(a, s);
    if (kit <= 9) return f8 ? launch_rt<4, 9, true>(a, s) : launch_rt<2, 9, false>(a, s);
    return 1;
}

// Skinny-M weight stream on the matrix cores (2 <= M <= 16 rows: beam search, CFG pairs).  The FMA kernel above costs
// M FMAs + an unpack per weight and goes VALU-bound past M = 2 (M = 5: 0.9 TB/s); here one v_mfma_f32_16x16x32_bf16
// consumes a 16-row x 32-k weight fragment exactly as it arrives from HBM (A operand: lane -> row l & 15, 8 consecutive k
// at 8 * (l >> 4)) against the activations as the B operand (col = activation row, same k), so the VALU does nothing in
// the loop.  A block owns 16 weight rows; its 4 waves interleave over 32-k blocks (together 256 contiguous bytes per
// row per step) and reduce through LDS.  Activations are re-read from L2 per fragment (no LDS staging, any K % 32 == 0).
template <int EPI, int RG>
__global__ __launch_bounds__(256) void gemv_mfma_kernel(const GemvArgs a) {
    typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
    __shared__ f32x4_t part[4][RG][64];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int i = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * (16 * RG);
    const int mrow = i < a.M ? i : a.M - 1;                            // columns >= M duplicate the last row (discarded)
    const bf16_t* wp[RG];
#pragma unroll
    for (int rg = 0; rg < RG; ++rg) {
        const int n = n0 + 16 * rg + i;
        wp[rg] = a.W + (size_t)(n < a.N ? n : a.N - 1) * a.ldw + g * 8;
    }
    const bf16_t* xp = a.x + (size_t)mrow * a.ldx + g * 8;
    const int KB = a.K >> 5;
    f32x4_t acc[RG];
#pragma unroll
    for (int rg = 0; rg < RG; ++rg) acc[rg] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    constexpr int U = RG >= 4 ? 4 : 8;                                 // fragments in flight per row group
    int kb = wave;
    for (; kb + 4 * (U - 1) < KB; kb += 4 * U) {
        u32x4 wv[RG][U], xv[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            xv[j] = ld16(xp + (kb + 4 * j) * 32);                      // one activation fragment serves RG weight fragments
#pragma unroll
            for (int rg = 0; rg < RG; ++rg) wv[rg][j] = ld_stream(reinterpret_cast<const u32x4*>(wp[rg] + (kb + 4 * j) * 32));
        }
#pragma unroll
        for (int j = 0; j < U; ++j)
#pragma unroll
            for (int rg = 0; rg < RG; ++rg)
                acc[rg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wv[rg][j]),
                                                                  __builtin_bit_cast(bf16x8_t, xv[j]), acc[rg], 0, 0, 0);
    }
    for (; kb < KB; kb += 4) {
        const u32x4 xv = ld16(xp + kb * 32);
#pragma unroll
        for (int rg = 0; rg < RG; ++rg) {
            const u32x4 wv = ld_stream(reinterpret_cast<const u32x4*>(wp[rg] + kb * 32));
            acc[rg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wv),
                                                              __builtin_bit_cast(bf16x8_t, xv), acc[rg], 0, 0, 0);
        }
    }
#pragma unroll
    for (int rg = 0; rg < RG; ++rg) part[wave][rg][lane] = acc[rg];
    __syncthreads();
    // wave w finishes row groups w, w + 4, ...: lane holds C[weight row 4g + r][activation row i], r = 0..3
    const int m = i;
    for (int rg = wave; rg < RG; rg += 4) {
        f32x4_t v = part[0][rg][lane];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const f32x4_t t = part[w][rg][lane];
            v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3];
        }
        if (m >= a.M) continue;
        const int nb = n0 + 16 * rg + 4 * g;                           // first of this lane's 4 output columns
        if constexpr (EPI == EPI_SWIGLU) {
#pragma unroll
            for (int r = 0; r < 4; r += 2) {
                const int n = nb + r;
                if (n + 1 < a.N) {
                    float gt = v[r], up = v[r + 1];
                    if (a.bias) { gt += bf2f(a.bias[n]); up += bf2f(a.bias[n + 1]); }
                    gt = bfround(gt); up = bfround(up);
                    a.out[(size_t)m * a.ldo + (n >> 1)] = f2bf(bfround(silu(gt)) * up);
                }
            }
        } else {
            float o[4];
            const bool full = nb + 3 < a.N;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = nb + r;
                float t = v[r];
                if (n < a.N) {
                    if (a.bias) t += bf2f(a.bias[n]);
                    t = bfround(t);
                    if constexpr (EPI == EPI_SILU) t = bfround(silu(t));
                    if constexpr (EPI == EPI_GELU) t = bfround(gelu_erf(t));
                    if constexpr (EPI == EPI_RESID) t = t + bf2f(a.res[(size_t)m * a.ldres + n]);
                }
                o[r] = t;
            }
            bf16_t* dst = a.out + (size_t)m * a.ldo + nb;
            if (full && ((reinterpret_cast<size_t>(dst) & 7) == 0)) {
                uint2 pk;
                pk.x = packbf(o[0], o[1]); pk.y = packbf(o[2], o[3]);
                *reinterpret_cast<uint2*>(dst) = pk;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (nb + r < a.N) dst[r] = f2bf(o[r]);
            }
        }
    }
}

template <int RG>
int launch_gemv_mfma_rg(const GemvArgs& a, hipStream_t s) {
    const dim3 grid((a.N + 16 * RG - 1) / (16 * RG)), block(256);
#define EMU_MF_CASE(E) case E: hipLaunchKernelGGL((gemv_mfma_kernel<E, RG>), grid, block, 0, s, a); break;
    switch (a.epi) {
        EMU_MF_CASE(EPI_NONE)
        EMU_MF_CASE(EPI_RESID)
        EMU_MF_CASE(EPI_SWIGLU)
        EMU_MF_CASE(EPI_SILU)
        EMU_MF_CASE(EPI_GELU)
        default: return -22;
    }
#undef EMU_MF_CASE
    EMU_CHECK_LAUNCH();
    return 0;
}

int launch_gemv_mfma(const GemvArgs& a, hipStream_t s) {
    // Row groups per workgroup (each activation fragment, re-read from L2, then serves RG weight fragments): measured
    // flat at M = 5 (RG 1 / 2 / 4: 69.5 / 71.7 / 82.9 us on qkv), so the activation re-read is not what holds the kernel
    // at 3.8 TB/s -- the 64-byte-per-row fragment loads are; RG = 1 keeps the most workgroups in flight.
    return launch_gemv_mfma_rg<1>(a, s);
}

// fp8 (OCP e4m3fn) weight stream: half the HBM bytes per token.  One 16-byte load = 16 weights of one row; the per-row
// fp32 scale is applied once to the fp32 dot product.  Same fused RMSNorm prologue / epilogues as the bf16 kernel.
template <int R, int MB, bool NORM, int EPI, int NW>
__global__ __launch_bounds__(NW * 64) void gemv_fp8_kernel(const GemvArgs a) {
    __shared__ float red[NW][R * MB];
    __shared__ float red_ss[NW][MB];
    __shared__ float fin[R * MB];
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int KV = a.K >> 4;                       // 16-element groups per row
    const int n0 = blockIdx.x * R;
    const uint8_t* W8 = reinterpret_cast<const uint8_t*>(a.W);
    const uint8_t* wrow[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int n = n0 + r;
        n = n < a.N ? n : a.N - 1;
        wrow[r] = W8 + (size_t)n * a.ldw;
    }
    // fused RMSNorm without a prologue: y = rinv * sum_k w[k] * (g[k] * x[k]), rinv = rsqrt(mean(x^2) + eps).  The sum of
    // squares rides along the one pass over x (every column belongs to exactly one lane of the workgroup), so the weight
    // stream starts at once instead of behind a load -> reduce -> barrier round trip; the reference's two bf16 rounding
    // points inside the norm are not reproduced (2^-9 relative, far below the e4m3 weight error of this optional stream).
    float ss[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) ss[m] = 0.f;
    float acc[R][MB];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int m = 0; m < MB; ++m) acc[r][m] = 0.f;

#pragma unroll 2
    for (int vi = tid; vi < KV; vi += NW * 64) {
        u32x4 wv[R];
#pragma unroll
        for (int r = 0; r < R; ++r) wv[r] = ld_stream(reinterpret_cast<const u32x4*>(wrow[r] + vi * 16));
        float xf[MB][16];
        float g[16];
        if constexpr (NORM) { unpack8(ld16(a.norm_w + vi * 16), g); unpack8(ld16(a.norm_w + vi * 16 + 8), g + 8); }
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            if (m < a.M) {
                unpack8(ld16(a.x + (size_t)m * a.ldx + vi * 16), xf[m]);
                unpack8(ld16(a.x + (size_t)m * a.ldx + vi * 16 + 8), xf[m] + 8);
                if constexpr (NORM) {
#pragma unroll
                    for (int j = 0; j < 16; ++j) { ss[m] = fmaf(xf[m][j], xf[m][j], ss[m]); xf[m][j] *= g[j]; }
                }
            } else {
#pragma unroll
                for (int j = 0; j < 16; ++j) xf[m][j] = 0.f;
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            f32x2_t wf[8];                                            // v_cvt_pk_f32_fp8 pairs feed v_pk_fma_f32
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                wf[2 * q] = __builtin_amdgcn_cvt_pk_f32_fp8(wv[r][q], false);
                wf[2 * q + 1] = __builtin_amdgcn_cvt_pk_f32_fp8(wv[r][q], true);
            }
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                f32x2_t a2 = {0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    a2 = __builtin_elementwise_fma(wf[j], f32x2_t{xf[m][2 * j], xf[m][2 * j + 1]}, a2);
                acc[r][m] += a2[0] + a2[1];
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            const float v = wave_sum(acc[r][m]);
            if (lane == 0) red[wave][r * MB + m] = v;
        }
    if constexpr (NORM) {
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            const float v = wave_sum(ss[m]);
            if (lane == 0) red_ss[wave][m] = v;
        }
    }
    __syncthreads();
    if (tid < R * MB) {
        const int n = n0 + tid / MB;
        float t = 0.f, rinv = 1.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) t += red[w][tid];
        if constexpr (NORM) {
            float q = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) q += red_ss[w][tid % MB];
            rinv = rsqrtf(q / (float)a.K + a.eps);
        }
        fin[tid] = t * rinv * a.wscale[n < a.N ? n : a.N - 1];
    }
    __syncthreads();
    if constexpr (EPI == EPI_SWIGLU) {
        if (tid < (R / 2) * MB) {
            const int j = tid / MB, m = tid % MB;
            const int n = n0 + 2 * j;
            if (m < a.M && n + 1 < a.N) {
                const float gt = bfround(fin[(2 * j) * MB + m] + (a.bias ? bf2f(a.bias[n]) : 0.f));
                const float up = bfround(fin[(2 * j + 1) * MB + m] + (a.bias ? bf2f(a.bias[n + 1]) : 0.f));
                a.out[(size_t)m * a.ldo + (n >> 1)] = f2bf(bfround(silu(gt)) * up);
            }
        }
    } else {
        if (tid < R * MB) {
            const int r = tid / MB, m = tid % MB;
            const int n = n0 + r;
            if (m < a.M && n < a.N) {
                float v = fin[tid];
                if (a.bias) v += bf2f(a.bias[n]);
                v = bfround(v);
                if constexpr (EPI == EPI_RESID) v = v + bf2f(a.res[(size_t)m * a.ldres + n]);
                a.out[(size_t)m * a.ldo + n] = f2bf(v);
            }
        }
    }
}

template <int R, int MB, bool NORM, int EPI, int NW>
__global__ __launch_bounds__(NW * 64) void gemv_fp8v8_kernel(const GemvArgs a) {
    __shared__ float red[NW][R * MB];
    __shared__ float red_ss[NW][MB];
    __shared__ float fin[R * MB];
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int KV = a.K >> 3;                       // 8-element groups per row (8-byte weight loads)
    const int n0 = blockIdx.x * R;
    const uint8_t* W8 = reinterpret_cast<const uint8_t*>(a.W);
    const uint8_t* wrow[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int n = n0 + r;
        n = n < a.N ? n : a.N - 1;
        wrow[r] = W8 + (size_t)n * a.ldw;
    }
    float ss[MB];                                  // sum of squares rides along the pass over x (see gemv_fp8_kernel)
#pragma unroll
    for (int m = 0; m < MB; ++m) ss[m] = 0.f;
    float acc[R][MB];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int m = 0; m < MB; ++m) acc[r][m] = 0.f;

#pragma unroll 2
    for (int vi = tid; vi < KV; vi += NW * 64) {
        u32x2 wv[R];
#pragma unroll
        for (int r = 0; r < R; ++r) wv[r] = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(wrow[r] + vi * 8));
        float xf[MB][8];
        float g[8];
        if constexpr (NORM) unpack8(ld16(a.norm_w + vi * 8), g);
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            if (m < a.M) {
                unpack8(ld16(a.x + (size_t)m * a.ldx + vi * 8), xf[m]);
                if constexpr (NORM) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) { ss[m] = fmaf(xf[m][j], xf[m][j], ss[m]); xf[m][j] *= g[j]; }
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) xf[m][j] = 0.f;
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            f32x2_t wf[4];                                            // v_cvt_pk_f32_fp8 pairs feed v_pk_fma_f32
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                wf[2 * q] = __builtin_amdgcn_cvt_pk_f32_fp8(wv[r][q], false);
                wf[2 * q + 1] = __builtin_amdgcn_cvt_pk_f32_fp8(wv[r][q], true);
            }
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                f32x2_t a2 = {0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    a2 = __builtin_elementwise_fma(wf[j], f32x2_t{xf[m][2 * j], xf[m][2 * j + 1]}, a2);
                acc[r][m] += a2[0] + a2[1];
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            const float v = wave_sum(acc[r][m]);
            if (lane == 0) red[wave][r * MB + m] = v;
        }
    if constexpr (NORM) {
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            const float v = wave_sum(ss[m]);
            if (lane == 0) red_ss[wave][m] = v;
        }
    }
    __syncthreads();
    if (tid < R * MB) {
        const int n = n0 + tid / MB;
        float t = 0.f, rinv = 1.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) t += red[w][tid];
        if constexpr (NORM) {
            float q = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) q += red_ss[w][tid % MB];
            rinv = rsqrtf(q / (float)a.K + a.eps);
        }
        fin[tid] = t * rinv * a.wscale[n < a.N ? n : a.N - 1];
    }
    __syncthreads();
    if constexpr (EPI == EPI_SWIGLU) {
        if (tid < (R / 2) * MB) {
            const int j = tid / MB, m = tid % MB;
            const int n = n0 + 2 * j;
            if (m < a.M && n + 1 < a.N) {
                const float gt = bfround(fin[(2 * j) * MB + m] + (a.bias ? bf2f(a.bias[n]) : 0.f));
                const float up = bfround(fin[(2 * j + 1) * MB + m] + (a.bias ? bf2f(a.bias[n + 1]) : 0.f));
                a.out[(size_t)m * a.ldo + (n >> 1)] = f2bf(bfround(silu(gt)) * up);
            }
        }
    } else {
        if (tid < R * MB) {
            const int r = tid / MB, m = tid % MB;
            const int n = n0 + r;
            if (m < a.M && n < a.N) {
                float v = fin[tid];
                if (a.bias) v += bf2f(a.bias[n]);
                v = bfround(v);
                if constexpr (EPI == EPI_RESID) v = v + bf2f(a.res[(size_t)m * a.ldres + n]);
                a.out[(size_t)m * a.ldo + n] = f2bf(v);
            }
        }
    }
}

template <int R, int MB, int NW>
int launch_fp8v8(const GemvArgs& a, hipStream_t s) {
    static_assert(R * MB <= NW * 64, "epilogue needs one thread per (row, m)");
    const dim3 grid((a.N + R - 1) / R), block(NW * 64);
    const bool norm = a.norm_w != nullptr;
#define EMU_FP8_CASE(E)                                                                                              \
    case E:                                                                                                          \
        if (norm) hipLaunchKernelGGL((gemv_fp8v8_kernel<R, MB, true, E, NW>), grid, block, 0, s, a);                 \
        else hipLaunchKernelGGL((gemv_fp8v8_kernel<R, MB, false, E, NW>), grid, block, 0, s, a);                     \
        break;
    switch (a.epi) {
        EMU_FP8_CASE(EPI_NONE)
        EMU_FP8_CASE(EPI_RESID)
        EMU_FP8_CASE(EPI_SWIGLU)
        default: return -22;
    }
#undef EMU_FP8_CASE
    EMU_CHECK_LAUNCH();
    return 0;
}

template <int R, int MB, int NW>
int launch_fp8(const GemvArgs& a, hipStream_t s) {
    static_assert(R * MB <= NW * 64, "epilogue needs one thread per (row, m)");
    const dim3 grid((a.N + R - 1) / R), block(NW * 64);
    const bool norm = a.norm_w != nullptr;
#define EMU_FP8_CASE(E)                                                                                              \
    case E:                                                                                                          \
        if (norm) hipLaunchKernelGGL((gemv_fp8_kernel<R, MB, true, E, NW>), grid, block, 0, s, a);                       \
        else hipLaunchKernelGGL((gemv_fp8_kernel<R, MB, false, E, NW>), grid, block, 0, s, a);                           \
        break;
    switch (a.epi) {
        EMU_FP8_CASE(EPI_NONE)
        EMU_FP8_CASE(EPI_RESID)
        EMU_FP8_CASE(EPI_SWIGLU)
        default: return -22;
    }
#undef EMU_FP8_CASE
    EMU_CHECK_LAUNCH();
    return 0;
}

template <int R, int MB, bool NORM, int PRE>
int launch_epi(const GemvArgs& a, hipStream_t s) {
    const dim3 grid((a.N + R - 1) / R), block(256);
    switch (a.epi) {
        case EPI_NONE:   hipLaunchKernelGGL((gemv_kernel<R, MB, NORM, EPI_NONE, PRE>), grid, block, 0, s, a); break;
        case EPI_RESID:  hipLaunchKernelGGL((gemv_kernel<R, MB, NORM, EPI_RESID, PRE>), grid, block, 0, s, a); break;
        case EPI_SWIGLU: hipLaunchKernelGGL((gemv_kernel<R, MB, NORM, EPI_SWIGLU, PRE>), grid, block, 0, s, a); break;
        case EPI_SILU:   hipLaunchKernelGGL((gemv_kernel<R, MB, NORM, EPI_SILU, PRE>), grid, block, 0, s, a); break;
        case EPI_GELU:   hipLaunchKernelGGL((gemv_kernel<R, MB, NORM, EPI_GELU, PRE>), grid, block, 0, s, a); break;
        default: return -22;
    }
    EMU_CHECK_LAUNCH();
    return 0;
}

template <int R, int MB>
int launch_norm(const GemvArgs& a, hipStream_t s) {
    // the preload-everything form only for the single-row decode case (register budget: PRE*R*4 VGPRs)
    // measured: the preload form wins only for plain streams (o_proj); with the RMSNorm prologue it loses 20 %
    // (clamped tail chunks + lower occupancy), so those keep the rolling loop.
    if constexpr (MB == 1 && R <= 4) {
        if (!a.norm_w && (a.K >> 3) <= 1024) return launch_epi<R, MB, false, 4>(a, s);
    }
    if constexpr (MB == 1) {
        if (a.norm_w && (a.K >> 3) <= 1024) return launch_epi<R, MB, true, -1>(a, s);
    }
    return a.norm_w ? launch_epi<R, MB, true, 0>(a, s) : launch_epi<R, MB, false, 0>(a, s);
}

template <int R>
int launch_mb(const GemvArgs& a, hipStream_t s) {
    if (a.M <= 1) return launch_norm<R, 1>(a, s);
    if (a.M <= 2) return launch_norm<R, 2>(a, s);
    if (a.M <= 3) return launch_norm<R, 3>(a, s);
    if (a.M <= 4) return launch_norm<R, 4>(a, s);
    if (a.M <= 5) return launch_norm<R, 5>(a, s);             // 5 beams: the reference's default decoding mode
    return launch_norm<R, 8>(a, s);
}

}  // namespace

int emu_gemv_rows_per_block(int N, int K, bool norm) {
    // measured (tools/kbench.py, profiles/): plain streams are fastest with 2 rows per workgroup (more, smaller workgroups
    // balance the 256 CUs better).  Kernels with the fused RMSNorm prologue wanted 8-16 rows so the prologue was amortised;
    // in the head form (K <= 8192: the weight stream starts before the statistics are reduced) the prologue no longer
    // stalls the stream and 4 rows win: decode 93.1 (16 rows) / 93.6 (8) / 95.0 (4) tokens/s, 2 rows lose 12 %.
    int R = norm ? ((K >> 3) <= 1024 ? 4 : 16) : 2;
    while (R > 2 && (N + R - 1) / R < (R == 16 ? 1024 : 512)) R >>= 1;
    return R;
}

// rows per workgroup for 2..8 activation rows (beam search, CFG pairs): 8 amortises the activation unpack best
int emu_gemv_rows_per_block_multi(int N) {
    int R = 8;
    while (R > 2 && (N + R - 1) / R < 512) R >>= 1;
    return R;
}

int emu_gemv_partition(int N, int K, bool norm, int epi) {
    // try_launch_wave's conditions (M = 1, bf16): no fused norm, EPI_NONE / EPI_RESID, K <= 2560, N >= 1024
    const int kitw = ((K >> 3) + 63) / 64;
    if (!norm && (epi == EPI_NONE || epi == EPI_RESID) && kitw <= 5 && N >= 1024) return 1;
    return 4;                                  // gemv_rt_kernel and gemv_kernel alike: thread t owns 16-byte columns t, t + 256, ...
}

int launch_gemv(const GemvArgs& a, hipStream_t s) {
    if (a.M < 1 || a.M > 16 || (a.K & 7) || a.N < 1) return -22;
    if (a.epi == EPI_SWIGLU && (a.N & 1)) return -22;
    // 9..16 rows: the 16x16x32 MFMA stream.  Up to 8 rows the v_dot2c block kernel is faster (M = 2: 6.4 vs 4.1 TB/s,
    // M = 5: 4.5 vs 3.9, M = 8: 3.7 vs 3.5; tools/kbench.py --filter rows): its row-contiguous 1 KiB loads use HBM better
    // than the MFMA fragment's 64 bytes per row.
    // 4..16 rows with whole 256-wide k stages: LDS-DMA stages + MFMA (gemv_thin.hip).  Measured on the LLaMA-33B shapes
    // (tools/thin_ab.py, profiles/r03_thin_stream_ab.log): 5 rows 42.7 / 17.9 / 74.6 / 42.6 us (qkv / o / gate-up / down) against
    // 46.4 / 21.7 / 77.4 / 47.6 on the v_dot2c kernel below, 8 rows 43.7 vs 60.7, 16 rows 49.2 vs 92.0 (register-fed MFMA); at 2-3
    // rows the v_dot2c kernel is level or ahead (qkv 40.6 vs 41.0).  emu_gemm_tune bit 2 switches it off (A/B).
    if (a.M >= 4 && !(emu_gemm_tune_get() & 4) && gemv_thin_ok(a)) return launch_gemv_thin(a, s);
    if (a.M > 8 && !a.wscale && !a.norm_w && (a.K & 31) == 0 && (a.ldw & 7) == 0 &&
        (a.ldx & 7) == 0)
        return launch_gemv_mfma(a, s);
    if (a.M > 8) return -22;
    if (a.wscale && ((a.K & 15) || a.M > 2)) return -22;
    { const int st = try_launch_wave(a, s); if (st != 1) return st; }
    { const int st = try_launch_rt(a, s); if (st != 1) return st; }
    if (a.wscale) {                                  // fp8 weight stream (decode, batch <= 2 built)
        // 16 weights per 16-byte load: K = 6656 is only 416 groups, so short rows run 2-wave blocks (3.25 trips per
        // lane, like the bf16 kernel) and long rows (down_proj, K = 17920) 4-wave blocks
        if (a.K <= 8192) return a.M <= 1 ? launch_fp8v8<8, 1, 4>(a, s) : launch_fp8v8<8, 2, 4>(a, s);   // 8-byte loads: 3.25 trips per lane at K = 6656
        const bool small = (a.N + 7) / 8 < 512;
        if (a.M <= 1) return small ? launch_fp8<4, 1, 2>(a, s) : launch_fp8<4, 1, 4>(a, s);   // down_proj: 4 rows x 4 waves (21.8 vs 23.1 us)
        return small ? launch_fp8<4, 2, 2>(a, s) : launch_fp8<8, 2, 2>(a, s);
    }
    int R = a.rows_per_block > 0 ? a.rows_per_block
                                 : (a.M > 1 ? emu_gemv_rows_per_block_multi(a.N)
                                            : emu_gemv_rows_per_block(a.N, a.K, a.norm_w != nullptr));
    if (a.M > 1 && R > 8) R = 8;                   // 16 rows per workgroup only pays at M = 1 (accumulator registers)
    switch (R) {
        case 2: return launch_mb<2>(a, s);
        case 4: return launch_mb<4>(a, s);
        case 8: return launch_mb<8>(a, s);
        case 16: return a.M <= 1 ? launch_norm<16, 1>(a, s) : launch_mb<8>(a, s);
        default: return -22;
    }
}
