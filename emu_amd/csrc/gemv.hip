// Decode-shape (M <= 8 rows) weight-streaming GEMV:  out[m, n] = epilogue( sum_k xeff[m, k] * W[n, k] )
//
// HBM-bound: every weight byte is read exactly once per call with non-temporal 16-byte loads; x (a few
// KB) is re-read from L1/L2.  One 256-thread workgroup owns R consecutive weight rows and splits K across
// its 4 waves (thread t owns 16-byte vectors t, t+256, ...), so N/R workgroups (>> 256 CUs) stream
// concurrently.  Optional fused prologue: LLaMA RMSNorm of x (fp32 variance, bf16 rounding points of the
// reference kept: xeff = bf16(g * bf16(x * rsqrt(mean(x^2)+eps)))).  Fused epilogues: bias, residual add,
// SwiGLU over interleaved (gate, up) row pairs.
//
// Replaces (reference call sites): LlamaDecoderLayer linears + RMSNorm reached from Emu2/emu/emu.py:133-138
// and :213-229 at S=1, project_up/project_down emu.py:131,147.  Algorithmic bytes per call = 2*N*K.
#include <cstdlib>

#include "common.h"
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

    auto consume = [&](int vi, const u32x4 (&wv)[R], const u32x4* xpre) {
        float g[8];
        if constexpr (NORM) unpack8(ld16(a.norm_w + vi * 8), g);
        float xf[MB][8];
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            if (m < a.M) {
                if (xpre) unpack8(*xpre, xf[m]);
                else unpack8(ld16(a.x + (size_t)m * a.ldx + vi * 8), xf[m]);
                if constexpr (NORM) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) xf[m][j] = bfround(g[j] * bfround(xf[m][j] * rinv[m]));
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) xf[m][j] = 0.f;
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float wf[8];
            unpack8(wv[r], wf);
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[r][m] = fmaf(wf[j], xf[m][j], acc[r][m]);
        }
    };

    if constexpr (PRE > 0) {
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
                const float gt = bfround(fin[(2 * j) * MB + m]);
                const float up = bfround(fin[(2 * j + 1) * MB + m]);
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
    static const bool no_pre = getenv("EMU_GEMV_NOPRE") != nullptr;
    // measured: the preload form wins only for plain streams (o_proj); with the RMSNorm prologue it loses 20 %
    // (clamped tail chunks + lower occupancy), so those keep the rolling loop.
    if constexpr (MB == 1 && R <= 4) {
        if (!no_pre && !a.norm_w && (a.K >> 3) <= 1024) return launch_epi<R, MB, false, 4>(a, s);
    }
    return a.norm_w ? launch_epi<R, MB, true, 0>(a, s) : launch_epi<R, MB, false, 0>(a, s);
}

template <int R>
int launch_mb(const GemvArgs& a, hipStream_t s) {
    if (a.M <= 1) return launch_norm<R, 1>(a, s);
    if (a.M <= 2) return launch_norm<R, 2>(a, s);
    if (a.M <= 4) return launch_norm<R, 4>(a, s);
    return launch_norm<R, 8>(a, s);
}

}  // namespace

int emu_gemv_rows_per_block(int N, int K, bool norm) {
    // measured (tools/kbench.py, profiles/): kernels with the fused RMSNorm prologue want 8 rows per workgroup so the
    // prologue is amortised; plain streams are fastest with 2 rows per workgroup (more, smaller workgroups balance the
    // 256 CUs better).
    (void)K;
    int R = norm ? 8 : 2;
    while (R > 2 && (N + R - 1) / R < 512) R >>= 1;
    return R;
}

int launch_gemv(const GemvArgs& a, hipStream_t s) {
    if (a.M < 1 || a.M > 8 || (a.K & 7) || a.N < 1) return -22;
    if (a.epi == EPI_SWIGLU && (a.N & 1)) return -22;
    static const char* force_r = getenv("EMU_GEMV_R");     // A/B runs
    int R = a.rows_per_block > 0 ? a.rows_per_block
                                 : (force_r ? atoi(force_r) : emu_gemv_rows_per_block(a.N, a.K, a.norm_w != nullptr));
    if (a.M > 4 && R > 4) R = 4;                   // bound the accumulator register file
    switch (R) {
        case 2: return launch_mb<2>(a, s);
        case 4: return launch_mb<4>(a, s);
        case 8: return launch_mb<8>(a, s);
        default: return -22;
    }
}
