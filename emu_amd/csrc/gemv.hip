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
#include "common.h"
#include "kernels.h"

namespace {

template <int R, int MB, bool NORM, int EPI>
__global__ __launch_bounds__(256) void gemv_kernel(const GemvArgs a) {
    __shared__ float red[4][R * MB];
    __shared__ float fin[R * MB];
    __shared__ float scratch[4];
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int KV = a.K >> 3;                       // 16-byte vectors per row
    const int n0 = blockIdx.x * R;

    float rinv[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) rinv[m] = 1.f;
    if constexpr (NORM) {
        float ss[MB];
#pragma unroll
        for (int m = 0; m < MB; ++m) ss[m] = 0.f;
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
            float t = block_sum<4>(ss[m], scratch);
            rinv[m] = rsqrtf(t / (float)a.K + a.eps);
        }
    }

    float acc[R][MB];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int m = 0; m < MB; ++m) acc[r][m] = 0.f;

    const bf16_t* wrow[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int n = n0 + r;
        n = n < a.N ? n : a.N - 1;                 // tail rows: clamp loads, mask stores
        wrow[r] = a.W + (size_t)n * a.ldw;
    }

#pragma unroll 2
    for (int vi = tid; vi < KV; vi += 256) {
        u32x4 wv[R];
#pragma unroll
        for (int r = 0; r < R; ++r) wv[r] = ld_stream(reinterpret_cast<const u32x4*>(wrow[r] + vi * 8));
        float g[8];
        if constexpr (NORM) unpack8(ld16(a.norm_w + vi * 8), g);
        float xf[MB][8];
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            if (m < a.M) {
                unpack8(ld16(a.x + (size_t)m * a.ldx + vi * 8), xf[m]);
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
    }

#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            float v = wave_sum(acc[r][m]);
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

template <int R, int MB, bool NORM>
int launch_epi(const GemvArgs& a, hipStream_t s) {
    const dim3 grid((a.N + R - 1) / R), block(256);
    switch (a.epi) {
        case EPI_NONE:   hipLaunchKernelGGL((gemv_kernel<R, MB, NORM, EPI_NONE>), grid, block, 0, s, a); break;
        case EPI_RESID:  hipLaunchKernelGGL((gemv_kernel<R, MB, NORM, EPI_RESID>), grid, block, 0, s, a); break;
        case EPI_SWIGLU: hipLaunchKernelGGL((gemv_kernel<R, MB, NORM, EPI_SWIGLU>), grid, block, 0, s, a); break;
        case EPI_SILU:   hipLaunchKernelGGL((gemv_kernel<R, MB, NORM, EPI_SILU>), grid, block, 0, s, a); break;
        case EPI_GELU:   hipLaunchKernelGGL((gemv_kernel<R, MB, NORM, EPI_GELU>), grid, block, 0, s, a); break;
        default: return -22;
    }
    EMU_CHECK_LAUNCH();
    return 0;
}

template <int R, int MB>
int launch_norm(const GemvArgs& a, hipStream_t s) {
    return a.norm_w ? launch_epi<R, MB, true>(a, s) : launch_epi<R, MB, false>(a, s);
}

template <int R>
int launch_mb(const GemvArgs& a, hipStream_t s) {
    if (a.M <= 1) return launch_norm<R, 1>(a, s);
    if (a.M <= 2) return launch_norm<R, 2>(a, s);
    if (a.M <= 4) return launch_norm<R, 4>(a, s);
    return launch_norm<R, 8>(a, s);
}

}  // namespace

int emu_gemv_rows_per_block(int N, int K) {
    // K=6656-class rows are short: take 8 rows per workgroup so each lane keeps >= 8 loads in flight;
    // long rows (down_proj K=17920) use 4.  Small N (TP shards) drops to 2 to keep >= 2 blocks per CU.
    int R = K >= 12288 ? 4 : 8;
    while (R > 2 && (N + R - 1) / R < 1024) R >>= 1;
    return R;
}

int launch_gemv(const GemvArgs& a, hipStream_t s) {
    if (a.M < 1 || a.M > 8 || (a.K & 7) || a.N < 1) return -22;
    if (a.epi == EPI_SWIGLU && (a.N & 1)) return -22;
    int R = a.rows_per_block > 0 ? a.rows_per_block : emu_gemv_rows_per_block(a.N, a.K);
    if (a.M > 4 && R > 4) R = 4;                   // bound the accumulator register file
    switch (R) {
        case 2: return launch_mb<2>(a, s);
        case 4: return launch_mb<4>(a, s);
        case 8: return launch_mb<8>(a, s);
        default: return -22;
    }
}
