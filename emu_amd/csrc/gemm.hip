// bf16 MFMA GEMM for prefill / ViT / UNet shapes:  C[m, n] = epilogue( sum_k A[m, k] * W[n, k] )
// ("NT": both operands K-contiguous, i.e. torch.nn.functional.linear(A, W)).
//
// CDNA4 design: 128(n) x 128(m) x 64(k) workgroup tile, 4 waves in a 2x2 grid, each wave owns a 64x64
// sub-tile as 2x2 v_mfma_f32_32x32x16_bf16 accumulators.  The WEIGHT tile is the MFMA A operand, so the
// accumulator registers of one lane run along n: 4 consecutive output columns per register quad ->
// 8-byte stores, bias/activation/SwiGLU pairs stay lane-local.  Tiles are staged global -> registers ->
// LDS (double-buffered, next tile's loads issued before the current tile's MFMAs), 128-byte LDS rows with
// a 16-byte-slot XOR swizzle (slot ^= (row>>1)&7) so every ds_read_b128 lane group hits 16 distinct slots.
// Workgroup ids are remapped so each XCD walks a contiguous range of tiles (weight tile reuse in its L2).
//
// Replaces the torch Linear calls on the reference hot path: Emu2/emu/eva_vit.py:106,112,198,250 (ViT),
// transformers LlamaAttention/LlamaMLP reached from Emu2/emu/emu.py:133-138,213-229 (prefill),
// project_up/down emu.py:201,147.  Algorithmic FLOPs = 2*M*N*K.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 128 * BK * 2;     // 16 KiB per operand tile

__device__ __forceinline__ int lds_off(int row, int chunk) {       // 128-byte rows, 8 slots of 16 B
    return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}

template <int EPI, bool CONV>
__global__ __launch_bounds__(256) void gemm_nt_kernel(const GemmArgs a) {
    __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];   // [buf][W | A]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave >> 1, wm = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;

    // XCD-aware bijective remap: XCD x gets the contiguous tile range it would get from a blocked split
    const int nwg = gridDim.x, b = blockIdx.x;
    const int xcd = b & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (b >> 3);
    const int tiles_m = (a.M + BM - 1) / BM;
    const int n0 = (wg / tiles_m) * BN, m0 = (wg % tiles_m) * BM;

    // staging: thread t moves 16-byte chunks q = t + 256*i of each 128x64 tile (row = q>>3, chunk = q&7)
    const bf16_t* gW[4];
    const bf16_t* gA[4];
    int soff[4], kc[4];
    int pb[4], py[4], px[4];                     // CONV: output pixel of each staged row
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = tid + 256 * i, row = q >> 3, c = q & 7;
        int gn = n0 + row; gn = gn < a.N ? gn : a.N - 1;
        int gm = m0 + row; gm = gm < a.M ? gm : a.M - 1;
        gW[i] = a.W + (size_t)gn * a.ldw + c * 8;
        gA[i] = a.A + (size_t)gm * a.lda + c * 8;
        soff[i] = lds_off(row, c);
        kc[i] = c * 8;
        if constexpr (CONV) {
            const int hw = a.conv.Hout * a.conv.Wout;
            pb[i] = gm / hw;
            const int r = gm - pb[i] * hw;
            py[i] = r / a.conv.Wout;
            px[i] = r - py[i] * a.conv.Wout;
        }
    }
    const int nk = (a.K + BK - 1) / BK;
    u32x4 rw[4], ra[4];
    auto gload = [&](int kt) {
        const int k0 = kt * BK;
        const u32x4 z = {0u, 0u, 0u, 0u};
        if constexpr (CONV) {
            // a 64-wide k tile lies inside one filter tap because Cin % 64 == 0
            const int tap = k0 / a.conv.Cin, ci0 = k0 - tap * a.conv.Cin;
            const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                rw[i] = ld16(gW[i] + k0);
                int yi, xi;
                bool ok;
                if (a.conv.mode == CONV_3X3_S2) {
                    yi = 2 * py[i] + ky - 1; xi = 2 * px[i] + kx - 1;
                    ok = yi >= 0 && yi < a.conv.Hin && xi >= 0 && xi < a.conv.Win;
                } else if (a.conv.mode == CONV_3X3_UP2) {       // nearest x2 upsample fused into the gather
                    const int yu = py[i] + ky - 1, xu = px[i] + kx - 1;
                    ok = yu >= 0 && yu < 2 * a.conv.Hin && xu >= 0 && xu < 2 * a.conv.Win;
                    yi = yu >> 1; xi = xu >> 1;
                } else {
                    yi = py[i] + ky - 1; xi = px[i] + kx - 1;
                    ok = yi >= 0 && yi < a.conv.Hin && xi >= 0 && xi < a.conv.Win;
                }
                const size_t off = (((size_t)pb[i] * a.conv.Hin + yi) * a.conv.Win + xi) * a.conv.Cin + ci0 + kc[i];
                ra[i] = ok ? ld16(a.A + off) : z;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool ok = (k0 + kc[i]) < a.K;
                rw[i] = ok ? ld16(gW[i] + k0) : z;
                ra[i] = ok ? ld16(gA[i] + k0) : z;
            }
        }
    };
    auto sstore = [&](int buf) {
        char* base = smem + buf * 2 * TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            st16(base + soff[i], rw[i]);
            st16(base + TILE_BYTES + soff[i], ra[i]);
        }
    };

    f32x16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    gload(0);
    sstore(0);
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = (kt + 1) < nk;
        if (more) gload(kt + 1);
        const char* sW = smem + cur * 2 * TILE_BYTES;
        const char* sA = sW + TILE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int ch = kk * 2 + hi;
            bf16x8_t wf[2], af[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = wn * 64 + i * 32 + l31;
                wf[i] = *reinterpret_cast<const bf16x8_t*>(sW + lds_off(row, ch));
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = wm * 64 + j * 32 + l31;
                af[j] = *reinterpret_cast<const bf16x8_t*>(sA + lds_off(row, ch));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i], af[j], acc[i][j], 0, 0, 0);
        }
        if (more) sstore(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // epilogue: lane holds, per accumulator, column m and rows n = nb + (r & 3), nb = .. + 8*(r>>2) + 4*hi
    const bool pair = (EPI == EPI_SWIGLU || EPI == EPI_GEGLU);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int m = m0 + wm * 64 + j * 32 + l31;
        if (m >= a.M) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nb = n0 + wn * 64 + i * 32 + 8 * g + 4 * hi;
                if (nb >= a.N) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e];
                const bool full = (nb + 3) < a.N;
                if (a.bias) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (full || nb + e < a.N) v[e] += bf2f(a.bias[nb + e]);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = bfround(v[e]);
                if (a.bias2) {
                    const bf16_t* b2 = a.bias2 + (size_t)(m / a.rows_per_batch) * a.ld_bias2 + nb;
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (full || nb + e < a.N) v[e] = bfround(v[e] + bf2f(b2[e]));
                }
                if constexpr (pair) {
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
                    bf16_t* dst = a.C + (size_t)m * a.ldc + (nb >> 1);
                    if (full) {
                        *reinterpret_cast<uint32_t*>(dst) = packbf(o0, o1);
                    } else {
                        if (nb + 1 < a.N) dst[0] = f2bf(o0);
                    }
                } else {
                    if constexpr (EPI == EPI_SILU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = bfround(silu(v[e]));
                    }
                    if constexpr (EPI == EPI_GELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = bfround(gelu_erf(v[e]));
                    }
                    bf16_t* dst = a.C + (size_t)m * a.ldc + nb;
                    if (full) {
                        if constexpr (EPI == EPI_RESID) {
                            const u32x2 rv = *reinterpret_cast<const u32x2*>(a.res + (size_t)m * a.ldres + nb);
                            v[0] += bflo(rv.x); v[1] += bfhi(rv.x); v[2] += bflo(rv.y); v[3] += bfhi(rv.y);
                        }
                        u32x2 ov;
                        ov.x = packbf(v[0], v[1]);
                        ov.y = packbf(v[2], v[3]);
                        *reinterpret_cast<u32x2*>(dst) = ov;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (nb + e < a.N) {
                                float t = v[e];
                                if constexpr (EPI == EPI_RESID) t += bf2f(a.res[(size_t)m * a.ldres + nb + e]);
                                dst[e] = f2bf(t);
                            }
                        }
                    }
                }
            }
        }
    }
}

}  // namespace

int launch_gemm(const GemmArgs& a, hipStream_t s) {
    if (a.M < 1 || a.N < 1 || (a.K & 7) || (a.ldw & 7) || (a.ldc & 3)) return -22;
    if ((a.epi == EPI_SWIGLU || a.epi == EPI_GEGLU) && (a.N & 1)) return -22;
    if (a.epi == EPI_RESID && (a.ldres & 3)) return -22;
    if (a.bias2 && a.rows_per_batch < 1) return -22;
    const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    const dim3 grid(tiles), block(256);
    if (a.conv.mode != CONV_NONE) {
        const ConvGeom& g = a.conv;
        if ((g.Cin & 63) || a.K != 9 * g.Cin || a.M % (g.Hout * g.Wout)) return -22;
        if (g.mode == CONV_3X3 && (g.Hout != g.Hin || g.Wout != g.Win)) return -22;
        if (g.mode == CONV_3X3_S2 && (g.Hout != (g.Hin + 1) / 2 || g.Wout != (g.Win + 1) / 2)) return -22;
        if (g.mode == CONV_3X3_UP2 && (g.Hout != 2 * g.Hin || g.Wout != 2 * g.Win)) return -22;
        switch (a.epi) {
            case EPI_NONE:  hipLaunchKernelGGL((gemm_nt_kernel<EPI_NONE, true>), grid, block, 0, s, a); break;
            case EPI_RESID: hipLaunchKernelGGL((gemm_nt_kernel<EPI_RESID, true>), grid, block, 0, s, a); break;
            default: return -22;
        }
        EMU_CHECK_LAUNCH();
        return 0;
    }
    if (a.lda & 7) return -22;
    switch (a.epi) {
        case EPI_NONE:   hipLaunchKernelGGL((gemm_nt_kernel<EPI_NONE, false>), grid, block, 0, s, a); break;
        case EPI_RESID:  hipLaunchKernelGGL((gemm_nt_kernel<EPI_RESID, false>), grid, block, 0, s, a); break;
        case EPI_SWIGLU: hipLaunchKernelGGL((gemm_nt_kernel<EPI_SWIGLU, false>), grid, block, 0, s, a); break;
        case EPI_SILU:   hipLaunchKernelGGL((gemm_nt_kernel<EPI_SILU, false>), grid, block, 0, s, a); break;
        case EPI_GELU:   hipLaunchKernelGGL((gemm_nt_kernel<EPI_GELU, false>), grid, block, 0, s, a); break;
        case EPI_GEGLU:  hipLaunchKernelGGL((gemm_nt_kernel<EPI_GEGLU, false>), grid, block, 0, s, a); break;
        default: return -22;
    }
    EMU_CHECK_LAUNCH();
    return 0;
}
