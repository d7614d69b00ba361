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
#include <cstdlib>

#include "common.h"
#include "kernels.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 128 * BK * 2;     // 16 KiB per operand tile

__device__ __forceinline__ int lds_off(int row, int chunk) {       // 128-byte rows, 8 slots of 16 B
    return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}


// Epilogue for one accumulator quad: lane-local 4 consecutive output columns nb..nb+3 of row m.
template <int EPI>
__device__ __forceinline__ void store_quad(const GemmArgs& a, int m, int nb, float (&v)[4]) {
    const bool full = (nb + 3) < a.N;
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
        if (nb + 1 < a.N) {
            const float o0 = (EPI == EPI_SWIGLU) ? bfround(silu(v[0])) * v[1] : v[0] * bfround(gelu_erf(v[1]));
            a.C[(size_t)m * a.ldc + (nb >> 1)] = f2bf(o0);
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
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int m = m0 + wm * 64 + j * 32 + l31;
        if (m >= a.M) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nb = n0 + wn * 64 + i * 32 + 8 * g + 4 * hi;
                if (nb >= a.N) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e];
                store_quad<EPI>(a, m, nb, v);
            }
    }
}


// ------------------------------------------------------------------------------------------------ v2 pipeline
// Same tile math as gemm_nt_kernel, different memory pipeline: operands go global -> LDS directly with
// global_load_lds (16 B per lane, no VGPR staging, no ds_write pass) into a 4-stage LDS ring, and the loads of tiles
// t+1, t+2 stay in flight ACROSS the per-tile barrier (counted s_waitcnt vmcnt(N), raw s_barrier), so HBM/L2 latency
// (~2-3k cycles under load) is covered by three tiles of MFMA work instead of one.  An LDS-DMA instruction writes
// lane-linear (wave base + lane*16), so the bank-conflict swizzle is applied to the per-lane SOURCE address and undone
// by the same XOR on the ds_read side.  MF = 32-row m-fragments per wave: 2 -> 128x128 tile, 1 -> 128(n) x 64(m) tile
// for problems with too few 128x128 tiles to fill 256 CUs.  Implicit-GEMM conv taps that fall outside the image read a
// 16-byte zero buffer instead of the activation.
__device__ __attribute__((aligned(16))) const uint32_t g_zero16[4] = {0u, 0u, 0u, 0u};


template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ void glds16(const bf16_t* src, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// Tile configuration: WN x WM waves, each owning NF x MF 32x32 accumulators; NSTG-deep LDS ring of 64-wide k tiles.
// KG > 1: KG groups of WN x WM waves split the four 16-wide k-steps of every tile between them (intra-workgroup split-K:
// twice the waves per SIMD for the same tile, partial accumulators summed through LDS in the epilogue).
template <int WN_, int WM_, int NF_, int MF_, int NSTG_, int KG_ = 1, int BK_ = 64>
struct TileCfg {
    static constexpr int WN = WN_, WM = WM_, NF = NF_, MF = MF_, NSTG = NSTG_, KG = KG_, BKv = BK_;
    static constexpr int THREADS = WN * WM * KG * 64;
    static_assert(KG == 1 || KG == 2, "k-groups: 1 or 2");
    static_assert(BKv == 64 || BKv == 32, "k tile: 64 (128-byte LDS rows) or 32 (64-byte rows, twice the ring depth)");
    static constexpr int BNv = WN * NF * 32, BMv = WM * MF * 32;
    static constexpr int ROWB = BKv * 2;                                      // bytes per LDS row
    static constexpr int SLOTS = ROWB / 16;                                   // 16-byte slots per row: 8 or 4
    static constexpr int RPI = 64 / SLOTS;                                    // rows one wave LDS-DMA instruction fills: 8 or 16
    static constexpr int W_BYTES = BNv * ROWB, A_BYTES = BMv * ROWB, ST_BYTES = W_BYTES + A_BYTES;
    static constexpr int NLW = (BNv * SLOTS) / THREADS, NLA = (BMv * SLOTS) / THREADS, LPT = NLW + NLA;   // LDS-DMA per thread per tile
    static_assert((BNv * SLOTS) % THREADS == 0 && (BMv * SLOTS) % THREADS == 0, "tile rows must split evenly over the waves");
    static_assert(NSTG * ST_BYTES <= 160 * 1024, "LDS ring exceeds 160 KiB");
    static_assert((NSTG - 2) * LPT <= 63, "vmcnt field is 6 bits");
    // conflict-free ds_read_b128 of MFMA fragments: 128-byte rows XOR the slot with (row >> 1) & 7, 64-byte rows with
    // (row >> 2) & 3 (16 consecutive rows then cover all 64 banks exactly once)
    __device__ static __forceinline__ int swz(int row) { return SLOTS == 8 ? ((row >> 1) & 7) : ((row >> 2) & 3); }
    __device__ static __forceinline__ int off(int row, int chunk) { return row * ROWB + ((chunk ^ swz(row)) << 4); }
};

template <int EPI, bool CONV, class T>
__global__ __launch_bounds__(T::THREADS) void gemm2_kernel(const GemmArgs a) {
    constexpr int NW = T::WN * T::WM * T::KG;           // waves per workgroup
    constexpr int RED_BYTES = T::KG > 1 ? T::WN * T::WM * T::NF * T::MF * 16 * 64 * 4 : 0;
    constexpr int SMEM_BYTES = T::NSTG * T::ST_BYTES > RED_BYTES ? T::NSTG * T::ST_BYTES : RED_BYTES;
    __shared__ __attribute__((aligned(16))) char smem[SMEM_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = wave / (T::WN * T::WM), wtile = wave % (T::WN * T::WM);
    const int wn = wtile / T::WM, wm = wtile % T::WM;
    const int l31 = lane & 31, hi = lane >> 5;

    // Workgroups [0, full_tiles) own whole tiles (XCD-aware bijective remap so the 8 L2s each see a compact set of
    // tiles); the rest are K-slices of the tail tiles (wave-quantisation fix: a 2.13-round problem would otherwise pay
    // for 3 rounds).  full_tiles == number of tiles and ksplit == 1 for an unsplit launch.
    const int b = blockIdx.x;
    int wg, ks = 0, nsl = 1;
    if (b < a.full_tiles) {
        const int nwg = a.full_tiles;
        const int xcd = b & 7, q8 = nwg >> 3, r8 = nwg & 7;
        wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (b >> 3);
    } else {
        const int j = b - a.full_tiles;
        wg = a.full_tiles + j / a.ksplit;
        ks = j - (wg - a.full_tiles) * a.ksplit;
        nsl = a.ksplit;
    }
    const int tiles_m = (a.M + T::BMv - 1) / T::BMv;
    const int n0 = (wg / tiles_m) * T::BNv, m0 = (wg % tiles_m) * T::BMv;

    // per-lane source of every LDS-DMA instruction: LDS row r = (i*NW + wave)*8 + lane/8, slot p = lane%8 receives global
    // chunk c = p ^ ((r >> 1) & 7)
    const bf16_t* gW[T::NLW];
    const bf16_t* gA[T::NLA];
    int pb[T::NLA], py[T::NLA], px[T::NLA], ac[T::NLA];
#pragma unroll
    for (int i = 0; i < T::NLW; ++i) {
        const int r = (i * NW + wave) * T::RPI + lane / T::SLOTS, c = (lane % T::SLOTS) ^ T::swz(r);
        int gn = n0 + r; gn = gn < a.N ? gn : a.N - 1;
        gW[i] = a.W + (size_t)gn * a.ldw + c * 8;
    }
#pragma unroll
    for (int i = 0; i < T::NLA; ++i) {
        const int r = (i * NW + wave) * T::RPI + lane / T::SLOTS, c = (lane % T::SLOTS) ^ T::swz(r);
        int gm = m0 + r; gm = gm < a.M ? gm : a.M - 1;
        gA[i] = a.A + (size_t)gm * a.lda + c * 8;
        ac[i] = c * 8;
        if constexpr (CONV) {
            const int hw = a.conv.Hout * a.conv.Wout;
            pb[i] = gm / hw;
            const int rr = gm - pb[i] * hw;
            py[i] = rr / a.conv.Wout;
            px[i] = rr - py[i] * a.conv.Wout;
        }
    }
    // split-K: this workgroup owns K-tiles [kt0, kt0 + nk) of slice blockIdx.y
    const int nk_all = a.K / T::BKv;
    const int kt0 = (int)((long)ks * nk_all / nsl);
    const int nk = (int)((long)(ks + 1) * nk_all / nsl) - kt0;
    auto issue = [&](int kt, int stage) {
        kt = kt < nk ? kt : nk - 1;                    // past-the-end tiles re-load the last one (keeps vmcnt counts uniform)
        const int k0 = (kt0 + kt) * T::BKv;
        char* base = smem + stage * T::ST_BYTES + wave * 1024;
#pragma unroll
        for (int i = 0; i < T::NLW; ++i) glds16(gW[i] + k0, base + i * NW * 1024);
        if constexpr (CONV) {
            const int tap = k0 / a.conv.Cin, ci0 = k0 - tap * a.conv.Cin;
            const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
            for (int i = 0; i < T::NLA; ++i) {
                int yi, xi;
                bool ok;
                if (a.conv.mode == CONV_3X3_S2) {
                    yi = 2 * py[i] + ky - 1; xi = 2 * px[i] + kx - 1;
                    ok = yi >= 0 && yi < a.conv.Hin && xi >= 0 && xi < a.conv.Win;
                } else if (a.conv.mode == CONV_3X3_UP2) {
                    const int yu = py[i] + ky - 1, xu = px[i] + kx - 1;
                    ok = yu >= 0 && yu < 2 * a.conv.Hin && xu >= 0 && xu < 2 * a.conv.Win;
                    yi = yu >> 1; xi = xu >> 1;
                } else {
                    yi = py[i] + ky - 1; xi = px[i] + kx - 1;
                    ok = yi >= 0 && yi < a.conv.Hin && xi >= 0 && xi < a.conv.Win;
                }
                const size_t off = (((size_t)pb[i] * a.conv.Hin + yi) * a.conv.Win + xi) * a.conv.Cin + ci0 + ac[i];
                const bf16_t* src = ok ? a.A + off : reinterpret_cast<const bf16_t*>(g_zero16);
                glds16(src, base + T::W_BYTES + i * NW * 1024);
            }
        } else {
#pragma unroll
            for (int i = 0; i < T::NLA; ++i) glds16(gA[i] + k0, base + T::W_BYTES + i * NW * 1024);
        }
    };

    f32x16_t acc[T::NF][T::MF];
#pragma unroll
    for (int i = 0; i < T::NF; ++i)
#pragma unroll
        for (int j = 0; j < T::MF; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#pragma unroll
    for (int t = 0; t < T::NSTG - 1; ++t) issue(t, t);
    for (int kt = 0; kt < nk; ++kt) {
        wait_vmcnt<(T::NSTG - 2) * T::LPT>();          // this wave's share of tile kt has landed
        __builtin_amdgcn_s_barrier();                  // ... and everyone's; everyone is also done reading tile kt-1
        issue(kt + T::NSTG - 1, (kt + T::NSTG - 1) % T::NSTG);
        const char* sW = smem + (kt % T::NSTG) * T::ST_BYTES;
        const char* sA = sW + T::W_BYTES;
        // fragments are double-buffered in registers: the ds_reads of k-step kk+1 are in flight under the MFMAs of kk
        bf16x8_t wf[2][T::NF], af[2][T::MF];
        auto frags = [&](int kk, int buf) {
            const int ch = kk * 2 + hi;
#pragma unroll
            for (int i = 0; i < T::NF; ++i)
                wf[buf][i] = *reinterpret_cast<const bf16x8_t*>(sW + T::off((wn * T::NF + i) * 32 + l31, ch));
#pragma unroll
            for (int j = 0; j < T::MF; ++j)
                af[buf][j] = *reinterpret_cast<const bf16x8_t*>(sA + T::off((wm * T::MF + j) * 32 + l31, ch));
        };
        constexpr int KSTEPS = (T::BKv / 16) / T::KG;  // k-steps of this wave's k-group
        const int kk0 = kg * KSTEPS;
        frags(kk0, 0);
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk) {
            if (kk < KSTEPS - 1) frags(kk0 + kk + 1, (kk + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);         // keep the next k-step's ds_reads ABOVE this k-step's MFMAs
#pragma unroll
            for (int i = 0; i < T::NF; ++i)
#pragma unroll
                for (int j = 0; j < T::MF; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[kk & 1][i], af[kk & 1][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    wait_vmcnt<0>();                                   // drain the tail LDS-DMA before the LDS is released

    if constexpr (T::KG > 1) {
        // sum the two k-groups' partial accumulators through LDS (the ring is dead now): group 1 writes, group 0 adds
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem) + (size_t)wtile * T::NF * T::MF * 16 * 64;
        if (kg == 1) {
#pragma unroll
            for (int i = 0; i < T::NF; ++i)
#pragma unroll
                for (int j = 0; j < T::MF; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) red[((i * T::MF + j) * 16 + r) * 64 + lane] = acc[i][j][r];
        }
        __syncthreads();
        if (kg == 1) return;
#pragma unroll
        for (int i = 0; i < T::NF; ++i)
#pragma unroll
            for (int j = 0; j < T::MF; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] += red[((i * T::MF + j) * 16 + r) * 64 + lane];
    }

#pragma unroll
    for (int j = 0; j < T::MF; ++j) {
        const int m = m0 + (wm * T::MF + j) * 32 + l31;
        if (m >= a.M) continue;
#pragma unroll
        for (int i = 0; i < T::NF; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nb = n0 + (wn * T::NF + i) * 32 + 8 * g + 4 * hi;
                if (nb >= a.N) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e];
                if (nsl > 1) {                         // raw fp32 slice tile; splitk_reduce_kernel applies the epilogue
                    float* dst = a.partial + ((size_t)(wg - a.full_tiles) * nsl + ks) * (T::BMv * T::BNv) +
                                 (size_t)(m - m0) * T::BNv + (nb - n0);
                    *reinterpret_cast<f32x4_t*>(dst) = f32x4_t{v[0], v[1], v[2], v[3]};
                } else {
                    store_quad<EPI>(a, m, nb, v);
                }
            }
    }
}

using CfgA = TileCfg<2, 2, 2, 2, 4>;     // 128 x 128, 4 waves, 4-stage ring (128 KiB)
using CfgB = TileCfg<2, 2, 2, 2, 2>;     // 128 x 128, 4 waves, 2 stages (64 KiB, 2 workgroups per CU)
using CfgC = TileCfg<4, 2, 2, 2, 3>;     // 256(n) x 128(m), 8 waves, 3 stages (144 KiB)
using CfgD = TileCfg<4, 2, 2, 4, 2>;     // 256 x 256, 8 waves of 64(n) x 128(m), 2 stages (128 KiB)
using CfgE = TileCfg<2, 2, 2, 1, 4>;     // 128(n) x 64(m), 4 waves, 4 stages (96 KiB): few-tile problems
using CfgF = TileCfg<2, 4, 2, 2, 3>;     // 128(n) x 256(m), 8 waves, 3 stages (144 KiB)
using CfgG = TileCfg<2, 2, 1, 1, 3>;     // 64 x 64, 4 waves, 3 stages (48 KiB, 3 workgroups per CU): latency-bound small GEMMs
using CfgH = TileCfg<2, 2, 2, 1, 3>;     // 128(n) x 64(m), 4 waves, 3 stages (72 KiB, 2 workgroups per CU)
using CfgI = TileCfg<2, 2, 2, 1, 6>;     // 128(n) x 64(m), 4 waves, 6 stages (144 KiB): 5 tiles (120 KiB) in flight per CU
using CfgJ = TileCfg<2, 2, 2, 2, 4>;     // 128 x 128, 4 waves, 4 stages (128 KiB): 3 tiles (96 KiB) in flight per CU
using CfgK = TileCfg<2, 2, 2, 1, 3, 2>;  // 128(n) x 64(m), 2 k-groups x 4 waves, 3 stages (72 KiB, 2 workgroups per CU)
using CfgL = TileCfg<2, 2, 2, 2, 2, 2>;  // 128 x 128, 2 k-groups x 4 waves, 2 stages (64 KiB, 2 workgroups per CU)
// 32-wide k tiles (BK_ = 32: 64-byte LDS rows, twice the ring depth) measured (profiles/r01_gemm_tilecfg_sweep_bk32.log):
// 256 x 256 x 32, 4 stages: 887 TF/s at M = 1544 (C: 842) but 25 % row padding at M = 770 / 1025 makes it lose there;
// 256 x 128 x 32 with 6 stages: below C everywhere.  Not dispatched.
// measured and dropped (profiles/r01_gemm_tilecfg_sweep_MN.log): 4 waves of 128(n) x 64(m) on a 256 x 128 tile
// (0.75 KB of LDS reads per MFMA instead of 1 KB): -5 % vs C; 4 waves of 128 x 128 on 256 x 256 (AGPR accumulators): -40 %

// second launch of a split-K GEMM: sum the K-slices of every tail tile in slice order (deterministic) and apply the
// fused epilogue.  SPLITK_RED_Y workgroups per tail tile (a handful of tail tiles must still fill the chip).
constexpr int SPLITK_RED_Y = 16;
template <int EPI, class T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmArgs a) {
    const int wg = a.full_tiles + blockIdx.x;
    const int tiles_m = (a.M + T::BMv - 1) / T::BMv;
    const int n0 = (wg / tiles_m) * T::BNv, m0 = (wg % tiles_m) * T::BMv;
    const float* base = a.partial + (size_t)blockIdx.x * a.ksplit * (T::BMv * T::BNv);
    constexpr int QN = T::BNv / 4;
    constexpr int PER = T::BMv * QN / SPLITK_RED_Y;                   // quads per workgroup (grid.y chunks of a tile)
    for (int q = blockIdx.y * PER + threadIdx.x; q < (blockIdx.y + 1) * PER; q += 256) {
        const int lm = q / QN, lq = q - lm * QN;
        const int m = m0 + lm, nb = n0 + lq * 4;
        if (m >= a.M || nb >= a.N) continue;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        for (int ks = 0; ks < a.ksplit; ++ks) {
            const f32x4_t t = *reinterpret_cast<const f32x4_t*>(base + (size_t)ks * (T::BMv * T::BNv) + (size_t)lm * T::BNv + lq * 4);
            v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3];
        }
        store_quad<EPI>(a, m, nb, v);
    }
}

float* g_splitk_scratch = nullptr;
size_t g_splitk_floats = 0;

// full_tiles whole-K workgroups followed by (tiles - full_tiles) * ksplit slice workgroups, one launch (+ the reduce)
template <int EPI, bool CONV, class T>
void launch_cfg(const GemmArgs& a, hipStream_t s, int full_tiles = -1, int ksplit = 1) {
    const int tiles = ((a.M + T::BMv - 1) / T::BMv) * ((a.N + T::BNv - 1) / T::BNv);
    GemmArgs b = a;
    b.full_tiles = full_tiles < 0 ? tiles : full_tiles;
    b.ksplit = ksplit;
    const int tail = tiles - b.full_tiles;
    hipLaunchKernelGGL((gemm2_kernel<EPI, CONV, T>), dim3(b.full_tiles + tail * ksplit), dim3(T::THREADS), 0, s, b);
    if (tail > 0) hipLaunchKernelGGL((splitk_reduce_kernel<EPI, T>), dim3(tail, SPLITK_RED_Y), dim3(256), 0, s, b);
}

inline int tiles_of(const GemmArgs& a, int bn, int bm) { return ((a.M + bm - 1) / bm) * ((a.N + bn - 1) / bn); }

template <int EPI, bool CONV>
void launch_v2(const GemmArgs& a, hipStream_t s) {
    static const char* force = getenv("EMU_GEMM_CFG");      // A/B runs: force one configuration
    char cfg = force ? force[0] : 0;
    // split-K, whole problem or tail only.  The 256x128 tile moves the fewest bytes per FLOP through L2 (the binding
    // resource of these kernels) but runs one workgroup per CU, so a problem of T tiles costs ceil(T / 256) rounds: a
    // 2048 x 1280 output is 80 tiles (0.31 round), the LLaMA qkv prefill 546 (2.13 -> 3 rounds).  The tiles beyond the
    // last full round (all of them when T < 256) are cut into K-slices so they fill the CUs once more: fp32 slice tiles
    // land in a scratch, a second launch sums them in order and applies the epilogue.  Needs N % 4 == 0, a non-GLU
    // epilogue (the pair lives in one quad, fine, but GLU halves the output row -- kept simple), >= 8 K-tiles per slice.
    if (!cfg || cfg == 'S') {
        static const char* sk_env = getenv("EMU_GEMM_SPLITK");         // A/B: 0 disables
        GemmArgs b = a;
        if (!b.partial) { b.partial = g_splitk_scratch; b.partial_floats = g_splitk_floats; }
        const int tc = tiles_of(a, 256, 128), nk = a.K / BK, slots = 256;
        const int full = (tc / slots) * slots, tail = tc - full;
        int ksplit = tail > 0 ? slots / tail : 1;
        if (ksplit > 8) ksplit = 8;
        static const char* mk_env = getenv("EMU_GEMM_SPLITK_MINK");     // A/B: minimum K-tiles per slice
        const int min_k = mk_env ? atoi(mk_env) : (full ? 8 : 16);
        while (ksplit > 1 && nk / ksplit < min_k) --ksplit;
        // Only the whole-problem case (fewer tiles than CUs) pays.  Slicing just the tail of a multi-round problem
        // (546 tiles = 2.13 rounds: 512 whole tiles + 34 x 7 slices in the same launch) measured flat (331 vs 333 us on
        // the LLaMA qkv prefill, ViT fc1 slightly worse): these kernels are bound by aggregate L2 bandwidth, not by
        // rounds of workgroups -- a thin last round simply runs faster.  EMU_GEMM_SPLITK=2 enables the tail form (A/B).
        const bool tail_ok = sk_env && atoi(sk_env) == 2 && tail * 2 <= slots && tc < 2048;
        const bool want = ksplit > 1 && (full == 0 || tail_ok);
        const bool epi_ok = EPI != EPI_SWIGLU && EPI != EPI_GEGLU;
        if (want && epi_ok && b.partial && (a.N & 3) == 0 &&
            (size_t)tail * ksplit * (CfgC::BMv * CfgC::BNv) <= b.partial_floats && !(sk_env && atoi(sk_env) == 0)) {
            launch_cfg<EPI, CONV, CfgC>(b, s, full, ksplit);
            return;
        }
        if (cfg == 'S') cfg = 0;
    }
    if (!cfg) {
        // 128x128 tiles are L1/TA-bandwidth-bound (64 FLOP/B needs ~64 B/clk/CU), so the largest problems take the
        // 256(n) x 128(m) tile; mid-size GEMMs 128x128 with two workgroups per CU; few-tile / long-K problems (UNet 32x32
        // level, implicit-GEMM convs, skinny ViT fc2) take 128 x 64 tiles with two k-groups of waves (intra-workgroup
        // split-K) and two workgroups per CU.  Thresholds from tools/kbench.py sweeps (profiles/r01_gemm_tilecfg_*).
        const int tc = tiles_of(a, 256, 128);
        static const char* glu_env = getenv("EMU_GEMM_GLU_CFG");      // A/B for the GEGLU / SwiGLU GEMMs only
        if (glu_env && (EPI == EPI_GEGLU || EPI == EPI_SWIGLU)) cfg = glu_env[0];
        else if (tc >= 1024) cfg = 'C';
        else if ((EPI == EPI_GEGLU || EPI == EPI_SWIGLU) && tc >= 512) cfg = 'C';   // in situ (UNet step) +1.5 % over 128x128
        else if (!CONV && tc >= 180 && tc < 400) cfg = 'C';           // ~one 256x128 tile per CU: LLaMA o/down prefill, ViT qkv
        else if (!CONV && tiles_of(a, 128, 128) >= 400) cfg = 'B';
        else {
            static const char* small_env = getenv("EMU_GEMM_SMALL_CFG");    // A/B for the few-tile fallback
            cfg = small_env ? small_env[0] : 'K';
        }
    }
    switch (cfg) {
        case 'C': launch_cfg<EPI, CONV, CfgC>(a, s); break;
        case 'K': launch_cfg<EPI, CONV, CfgK>(a, s); break;
        case 'L': launch_cfg<EPI, CONV, CfgL>(a, s); break;
        default:  launch_cfg<EPI, CONV, CfgB>(a, s); break;
    }
}


}  // namespace

void emu_gemm_set_splitk_scratch(float* ptr, size_t floats) { g_splitk_scratch = ptr; g_splitk_floats = floats; }

int launch_gemm(const GemmArgs& a, hipStream_t s) {
    if (a.M < 1 || a.N < 1 || (a.K & 7) || (a.ldw & 7) || (a.ldc & 3)) return -22;
    if ((a.epi == EPI_SWIGLU || a.epi == EPI_GEGLU) && (a.N & 1)) return -22;
    if (a.epi == EPI_RESID && (a.ldres & 3)) return -22;
    if (a.bias2 && a.rows_per_batch < 1) return -22;
    const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    const dim3 grid(tiles), block(256);
    static const bool force_v1 = getenv("EMU_GEMM_V1") != nullptr;
    const bool v2 = !force_v1 && (a.K % BK) == 0;
    if (a.conv.mode != CONV_NONE) {
        const ConvGeom& g = a.conv;
        if ((g.Cin & 63) || a.K != 9 * g.Cin || a.M % (g.Hout * g.Wout)) return -22;
        if (g.mode == CONV_3X3 && (g.Hout != g.Hin || g.Wout != g.Win)) return -22;
        if (g.mode == CONV_3X3_S2 && (g.Hout != (g.Hin + 1) / 2 || g.Wout != (g.Win + 1) / 2)) return -22;
        if (g.mode == CONV_3X3_UP2 && (g.Hout != 2 * g.Hin || g.Wout != 2 * g.Win)) return -22;
        if (v2) {
            switch (a.epi) {
                case EPI_NONE:  launch_v2<EPI_NONE, true>(a, s); break;
                case EPI_RESID: launch_v2<EPI_RESID, true>(a, s); break;
                default: return -22;
            }
            EMU_CHECK_LAUNCH();
            return 0;
        }
        switch (a.epi) {
            case EPI_NONE:  hipLaunchKernelGGL((gemm_nt_kernel<EPI_NONE, true>), grid, block, 0, s, a); break;
            case EPI_RESID: hipLaunchKernelGGL((gemm_nt_kernel<EPI_RESID, true>), grid, block, 0, s, a); break;
            default: return -22;
        }
        EMU_CHECK_LAUNCH();
        return 0;
    }
    if (a.lda & 7) return -22;
    if (v2) {
        switch (a.epi) {
            case EPI_NONE:   launch_v2<EPI_NONE, false>(a, s); break;
            case EPI_RESID:  launch_v2<EPI_RESID, false>(a, s); break;
            case EPI_SWIGLU: launch_v2<EPI_SWIGLU, false>(a, s); break;
            case EPI_SILU:   launch_v2<EPI_SILU, false>(a, s); break;
            case EPI_GELU:   launch_v2<EPI_GELU, false>(a, s); break;
            case EPI_GEGLU:  launch_v2<EPI_GEGLU, false>(a, s); break;
            default: return -22;
        }
        EMU_CHECK_LAUNCH();
        return 0;
    }
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
