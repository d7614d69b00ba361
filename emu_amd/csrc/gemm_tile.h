// Device code shared by the MFMA GEMM kernels (gemm.hip: 128x128 / 256x128 / 128x64 tiles; gemm256.hip: the 256x256
// ping-pong tile): the fused epilogue of one accumulator quad, LDS-DMA and counted-wait helpers, the implicit-GEMM 3x3
// gather and the split-K reduce launch.
#pragma once
#include "common.h"
#include "kernels.h"

namespace emu_gemm {

// 16 zero bytes: source of LDS-DMA chunks that lie outside the image (conv taps) or beyond K (ragged last k tile)
static __device__ __attribute__((aligned(16))) const uint32_t g_zero16[4] = {0u, 0u, 0u, 0u};

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// one LDS-DMA instruction: 16 bytes per lane, written lane-linear at (wave-uniform) lds_wave_base + lane * 16
__device__ __forceinline__ void glds16(const void* src, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// input pixel of output pixel (py, px) under filter tap (ky, kx); false when the tap falls outside the image
__device__ __forceinline__ bool conv_tap(const ConvGeom& g, int py, int px, int ky, int kx, int& yi, int& xi) {
    if (g.mode == CONV_3X3_S2) {
        yi = 2 * py + ky - 1; xi = 2 * px + kx - 1;
        return yi >= 0 && yi < g.Hin && xi >= 0 && xi < g.Win;
    }
    if (g.mode == CONV_3X3_UP2) {                       // nearest x2 upsample fused into the gather
        const int yu = py + ky - 1, xu = px + kx - 1;
        yi = yu >> 1; xi = xu >> 1;
        return yu >= 0 && yu < 2 * g.Hin && xu >= 0 && xu < 2 * g.Win;
    }
    yi = py + ky - 1; xi = px + kx - 1;
    return yi >= 0 && yi < g.Hin && xi >= 0 && xi < g.Win;
}

// Epilogue for one accumulator quad: lane-local 4 consecutive output columns nb..nb+3 of row m.
template <int EPI>
__device__ __forceinline__ void store_quad(const GemmArgs& a, int m, int nb, float (&v)[4]) {
    // 8-byte accesses need every row start 8-byte aligned: ldc (and ldres) % 4 == 0; other strides (a [M, 32274] logits
    // buffer) take the scalar path below
    // (GLU epilogues store one 4-byte pair per quad: an even ldc is enough, and launch_gemm requires it)
    constexpr bool GLU = EPI == EPI_SWIGLU || EPI == EPI_GEGLU;
    const bool full = (nb + 3) < a.N && (GLU || ((a.ldc | (EPI == EPI_RESID ? a.ldres : 0)) & 3) == 0);
    if (full) {
        if (a.bias) {
            const u32x2 bv = *reinterpret_cast<const u32x2*>(a.bias + nb);
            v[0] += bflo(bv.x); v[1] += bfhi(bv.x); v[2] += bflo(bv.y); v[3] += bfhi(bv.y);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = bfround(v[e]);
        if (a.bias2) {
            const u32x2 bv = *reinterpret_cast<const u32x2*>(a.bias2 + (size_t)(m / a.rows_per_batch) * a.ld_bias2 + nb);
            v[0] = bfround(v[0] + bflo(bv.x)); v[1] = bfround(v[1] + bfhi(bv.x));
            v[2] = bfround(v[2] + bflo(bv.y)); v[3] = bfround(v[3] + bfhi(bv.y));
        }
        if constexpr (EPI == EPI_SWIGLU || EPI == EPI_GEGLU) {
            // interleaved rows (2j, 2j+1): SwiGLU = (gate, up) -> bf16(bf16(silu(gate)) * up)
            //                               GEGLU  = (hidden, gate) -> bf16(hidden * bf16(gelu(gate)))
            float o0, o1;
            if constexpr (EPI == EPI_SWIGLU) {
                o0 = bfround(silu(v[0])) * v[1];
                o1 = bfround(silu(v[2])) * v[3];
            } else {
                o0 = v[0] * bfround(gelu_erf(v[1]));
                o1 = v[2] * bfround(gelu_erf(v[3]));
            }
            *reinterpret_cast<uint32_t*>(a.C + (size_t)m * a.ldc + (nb >> 1)) = packbf(o0, o1);
        } else {
            if constexpr (EPI == EPI_SILU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = bfround(silu(v[e]));
            }
            if constexpr (EPI == EPI_GELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = bfround(gelu_erf(v[e]));
            }
            if constexpr (EPI == EPI_RESID) {
                const u32x2 rv = *reinterpret_cast<const u32x2*>(a.res + (size_t)m * a.ldres + nb);
                v[0] += bflo(rv.x); v[1] += bfhi(rv.x); v[2] += bflo(rv.y); v[3] += bfhi(rv.y);
            }
            u32x2 ov;
            ov.x = packbf(v[0], v[1]);
            ov.y = packbf(v[2], v[3]);
            *reinterpret_cast<u32x2*>(a.C + (size_t)m * a.ldc + nb) = ov;
        }
        return;
    }
    // ragged last columns (N % 4 != 0): scalar path
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (nb + e < a.N) {
            if (a.bias) v[e] += bf2f(a.bias[nb + e]);
            v[e] = bfround(v[e]);
            if (a.bias2) v[e] = bfround(v[e] + bf2f(a.bias2[(size_t)(m / a.rows_per_batch) * a.ld_bias2 + nb + e]));
        }
    }
    if constexpr (EPI == EPI_SWIGLU || EPI == EPI_GEGLU) {
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
            if (nb + e + 1 < a.N) {
                const float o = (EPI == EPI_SWIGLU) ? bfround(silu(v[e])) * v[e + 1] : v[e] * bfround(gelu_erf(v[e + 1]));
                a.C[(size_t)m * a.ldc + ((nb + e) >> 1)] = f2bf(o);
            }
        }
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (nb + e < a.N) {
                float t = v[e];
                if constexpr (EPI == EPI_SILU) t = bfround(silu(t));
                if constexpr (EPI == EPI_GELU) t = bfround(gelu_erf(t));
                if constexpr (EPI == EPI_RESID) t += bf2f(a.res[(size_t)m * a.ldres + nb + e]);
                a.C[(size_t)m * a.ldc + nb + e] = f2bf(t);
            }
        }
    }
}

// second launch of a split-K GEMM: sum the K-slices of every sliced tile in slice order (deterministic) and apply the
// fused epilogue.  SPLITK_RED_Y workgroups per tile (a handful of tiles must still fill the chip).
constexpr int SPLITK_RED_Y = 16;
template <int EPI, int BMv, int BNv>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmArgs a) {
    const int wg = a.full_tiles + blockIdx.x;
    const int tiles_m = (a.M + BMv - 1) / BMv;
    const int n0 = (wg / tiles_m) * BNv, m0 = (wg % tiles_m) * BMv;
    const float* base = a.partial + (size_t)blockIdx.x * a.ksplit * (BMv * BNv);
    constexpr int QN = BNv / 4;
    constexpr int PER = BMv * QN / SPLITK_RED_Y;                      // quads per workgroup (grid.y chunks of a tile)
    for (int q = blockIdx.y * PER + threadIdx.x; q < (blockIdx.y + 1) * PER; q += 256) {
        const int lm = q / QN, lq = q - lm * QN;
        const int m = m0 + lm, nb = n0 + lq * 4;
        if (m >= a.M || nb >= a.N) continue;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        for (int ks = 0; ks < a.ksplit; ++ks) {
            const f32x4_t t = *reinterpret_cast<const f32x4_t*>(base + (size_t)ks * (BMv * BNv) + (size_t)lm * BNv + lq * 4);
            v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3];
        }
        store_quad<EPI>(a, m, nb, v);
    }
}

}  // namespace emu_gemm
