// 256(n) x 256(m) x 64(k) bf16 MFMA GEMM tile with a ping-pong phase schedule -- the MFMA-bound shapes of the path
// (LLaMA prefill, ViT blocks, the UNet's large GEMMs and implicit-GEMM convs):  C[m, n] = epilogue(sum_k A[m, k] W[n, k]).
//
// Why this tile: the 128x128 / 256x128 tiles of gemm.hip need 64 / 47 bytes per clock per CU from L2 at MFMA peak,
// which is the whole L2 -> LDS path (~56 B/clk/CU); 256x256 needs 31.  Why this schedule: with one barrier pair per
// k tile every wave of the workgroup loads, then computes, in lock step, and the matrix pipe idles during the loads.
//
// Structure (8 waves = 2 groups of 4, one wave of each group per SIMD; wave (wr, wc) owns a 128(n) x 64(m) output as
// 2x2 "quadrants" of 64(n) x 32(m), each 2 v_mfma_f32_32x32x16_bf16 accumulators):
//   * a k tile is staged as four 16 KiB UNITS -- P0, P1 (weight rows: the first / second 64 rows of every wave row),
//     Q0, Q1 (activation rows: the first / second 32 rows of every wave column) -- by LDS-DMA (global_load_lds, 16 B per
//     lane, 2 instructions per thread per unit), bank-conflict swizzle applied to the per-lane SOURCE address;
//   * a k tile is consumed in four PHASES, one quadrant (8 MFMAs) each:
//         phase 0: read Q0 + P0, compute (P0, Q0)      phase 1: read Q1, compute (P0, Q1)
//         phase 2: read P1,      compute (P1, Q1)      phase 3:          compute (P1, Q0)
//     every phase = { ds_reads, one unit of LDS-DMA for a later k tile, counted vmcnt } barrier { MFMAs } barrier;
//   * group 1 runs one barrier behind group 0, so on every SIMD one wave is in its MFMA segment while its partner
//     reads LDS / issues DMA: the matrix pipe sees back-to-back MFMA segments (s_setprio 1 inside them);
//   * phase p issues unit p + 6 (in the order P0, Q0, Q1, P1 of consecutive k tiles) and then waits vmcnt(8) = "every
//     unit up to p + 2 has landed": a unit is in flight for 4 phases (~1.2k cycles of MFMA work), is waited for one
//     phase (one barrier pair) before its first ds_read, and is overwritten >= 2 phases after its last ds_read -- the
//     ordering rules of LDS-DMA under a staggered barrier (cdna_hip_programming.md, "8-phase template").
// LDS: 2 k tiles x 4 units x 16 KiB = 128 KiB, one workgroup per CU.  Registers: 128 accumulator + 64 fragment.
//
// Replaces the same reference calls as gemm.hip (torch Linear / Conv2d on the ViT, LLaMA-prefill and UNet paths).
#include "gemm_tile.h"

using namespace emu_gemm;

namespace {

constexpr int UNIT = 128 * 128;          // 128 LDS rows of 128 bytes
constexpr int BUFB = 4 * UNIT;           // one k tile
enum { U_P0 = 0, U_Q0 = 1, U_Q1 = 2, U_P1 = 3 };       // unit order inside a k-tile buffer = staging order

template <int V> struct IC { static constexpr int value = V; };

__device__ __forceinline__ void bar() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
}

template <int EPI, bool CONV>
__global__ __launch_bounds__(512) void gemm_pp_kernel(const GemmArgs a) {
    __shared__ __attribute__((aligned(16))) char smem[2 * BUFB];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int l31 = lane & 31, hi = lane >> 5;

    // workgroups [0, full_tiles): whole-K tiles, XCD-aware bijective remap; the rest: K-slices of the remaining tiles
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
    const int tiles_m = (a.M + 255) >> 8;
    const int n0 = (wg / tiles_m) << 8, m0 = (wg % tiles_m) << 8;

    // ---- LDS-DMA sources.  Instruction i (0, 1) of a unit fills LDS rows r = i*64 + srow, srow = wave*8 + lane/8, slot
    // lane%8 <- global chunk slot ^ ((r >> 1) & 7).  P unit s: row r holds weight row n0 + (r >> 6)*128 + s*64 + (r & 63);
    // Q unit s: row r holds activation row m0 + (r >> 5)*64 + s*32 + (r & 31).
    const int srow = wave * 8 + (lane >> 3);
    const int sck = ((lane & 7) ^ (((wave & 1) << 2) | (lane >> 4))) * 8;
    const bf16_t* gP[2][2];
    const bf16_t* gQ[2][2];
    int qpix[2][2];                                 // CONV: output pixel (b << 20 | y << 10 | x) of the row
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int gn = n0 + i * 128 + s * 64 + srow;
            gn = gn < a.N ? gn : a.N - 1;
            gP[s][i] = a.W + (size_t)gn * a.ldw + sck;
            const int r = i * 64 + srow;
            int gm = m0 + (r >> 5) * 64 + s * 32 + (r & 31);
            gm = gm < a.M ? gm : a.M - 1;
            if constexpr (CONV) {
                const int hw = a.conv.Hout * a.conv.Wout;
                const int pb = gm / hw, rr = gm - pb * hw;
                const int py = rr / a.conv.Wout, px = rr - py * a.conv.Wout;
                qpix[s][i] = (pb << 20) | (py << 10) | px;
                gQ[s][i] = a.A + sck;
            } else {
                gQ[s][i] = a.A + (size_t)gm * a.lda + sck;
            }
        }

    const int nk_all = a.K >> 6;
    const int kt0 = (int)((long)ks * nk_all / nsl);
    const int nk = (int)((long)(ks + 1) * nk_all / nsl) - kt0;

    auto stage = [&](auto uc, auto bc, int tau) {
        constexpr int U = decltype(uc)::value, BF = decltype(bc)::value;
        constexpr int S = (U == U_P1 || U == U_Q1) ? 1 : 0;
        const int k0 = (kt0 + tau) << 6;
        char* base = smem + BF * BUFB + U * UNIT + wave * 1024;
        if constexpr (U == U_P0 || U == U_P1) {
#pragma unroll
            for (int i = 0; i < 2; ++i) glds16(gP[S][i] + k0, base + i * 8192);
        } else if constexpr (!CONV) {
#pragma unroll
            for (int i = 0; i < 2; ++i) glds16(gQ[S][i] + k0, base + i * 8192);
        } else {
            // a 64-wide k tile lies inside one filter tap (Cin % 64 == 0)
            const int tap = k0 / a.conv.Cin, ci0 = k0 - tap * a.conv.Cin;
            const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int pb = qpix[S][i] >> 20, py = (qpix[S][i] >> 10) & 1023, px = qpix[S][i] & 1023;
                int yi, xi;
                const bool ok = conv_tap(a.conv, py, px, ky, kx, yi, xi);
                const size_t off = (((size_t)pb * a.conv.Hin + yi) * a.conv.Win + xi) * a.conv.Cin + ci0;
                const bf16_t* src = ok ? gQ[S][i] + off : reinterpret_cast<const bf16_t*>(g_zero16);
                glds16(src, base + i * 8192);
            }
        }
    };

    // ---- fragment reads: lane (l31, hi) reads row base + l31, chunk (2*kk + hi) ^ swizzle(row), swizzle = (l31 >> 1) & 7
    int lp[4], lq[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const int o = l31 * 128 + ((((kk << 1) | hi) ^ ((l31 >> 1) & 7)) << 4);
        lp[kk] = o + wr * 8192;                     // P units: wave row wr owns LDS rows wr*64 .. +63
        lq[kk] = o + wc * 4096;                     // Q units: wave column wc owns LDS rows wc*32 .. +31
    }
    bf16x8_t pf[2][4], q0f[4], q1f[4];
    auto read_p = [&](const char* ub) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) pf[i][kk] = *reinterpret_cast<const bf16x8_t*>(ub + i * 4096 + lp[kk]);
    };
    auto read_q = [&](bf16x8_t (&q)[4], const char* ub) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) q[kk] = *reinterpret_cast<const bf16x8_t*>(ub + lq[kk]);
    };

    f32x16_t acc[2][2][2];                          // [P sub-tile][Q sub-tile][n fragment]
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[x][y][i][r] = 0.f;

    // wave-uniform validity of the quadrants (ragged last tiles): skipped MFMA clusters, clamped loads, predicated stores
    bool pv[2], qv[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        pv[s] = (n0 + wr * 128 + s * 64) < a.N;
        qv[s] = (m0 + wc * 64 + s * 32) < a.M;
    }
    auto mma = [&](f32x16_t (&c)[2], const bf16x8_t (&q)[4], bool valid) {
        __builtin_amdgcn_sched_barrier(0);
        if (valid) {
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int i = 0; i < 2; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf[i][kk], q[kk], c[i], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    // one k tile = 4 phases; tile t lives in buffer BF = t & 1
    auto tile = [&](auto bc, int t) {
        constexpr int BF = decltype(bc)::value;
        const char* sb = smem + BF * BUFB;
        // phase 0
        read_q(q0f, sb + U_Q0 * UNIT);
        read_p(sb + U_P0 * UNIT);
        if (t + 1 < nk) { stage(IC<U_Q1>{}, IC<BF ^ 1>{}, t + 1); wait_vmcnt<8>(); } else wait_vmcnt<2>();
        bar();
        mma(acc[0][0], q0f, pv[0] && qv[0]);
        bar();
        // phase 1
        read_q(q1f, sb + U_Q1 * UNIT);
        if (t + 1 < nk) { stage(IC<U_P1>{}, IC<BF ^ 1>{}, t + 1); wait_vmcnt<8>(); } else wait_vmcnt<0>();
        bar();
        mma(acc[0][1], q1f, pv[0] && qv[1]);
        bar();
        // phase 2
        read_p(sb + U_P1 * UNIT);
        if (t + 2 < nk) { stage(IC<U_P0>{}, IC<BF>{}, t + 2); wait_vmcnt<8>(); } else if (t + 2 == nk) wait_vmcnt<6>();
        bar();
        mma(acc[1][1], q1f, pv[1] && qv[1]);
        bar();
        // phase 3
        if (t + 2 < nk) { stage(IC<U_Q0>{}, IC<BF>{}, t + 2); wait_vmcnt<8>(); } else if (t + 2 == nk) wait_vmcnt<4>();
        bar();
        mma(acc[1][0], q0f, pv[1] && qv[0]);
        bar();
    };

    // prologue: units 0..5 = all of k tile 0, P0 and Q0 of k tile 1
    stage(IC<U_P0>{}, IC<0>{}, 0);
    stage(IC<U_Q0>{}, IC<0>{}, 0);
    stage(IC<U_Q1>{}, IC<0>{}, 0);
    stage(IC<U_P1>{}, IC<0>{}, 0);
    if (nk > 1) {
        stage(IC<U_P0>{}, IC<1>{}, 1);
        stage(IC<U_Q0>{}, IC<1>{}, 1);
        wait_vmcnt<8>();
    } else {
        wait_vmcnt<4>();
    }
    bar();
    if (wr == 1) bar();                             // group 1 runs one barrier behind group 0
    for (int t = 0; t < nk; t += 2) {
        tile(IC<0>{}, t);
        if (t + 1 < nk) tile(IC<1>{}, t + 1);
    }
    if (wr == 0) bar();

    // ---- epilogue: accumulator (x, y, i): rows n = n0 + wr*128 + x*64 + i*32 + 8*g + 4*hi + e, column m = .. + l31
#pragma unroll
    for (int y = 0; y < 2; ++y) {
        const int m = m0 + wc * 64 + y * 32 + l31;
        if (m >= a.M) continue;
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int nb = n0 + wr * 128 + x * 64 + i * 32 + 8 * g + 4 * hi;
                    if (nb >= a.N) continue;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[x][y][i][4 * g + e];
                    if (nsl > 1) {                  // raw fp32 slice tile; splitk_reduce_kernel applies the epilogue
                        float* dst = a.partial + ((size_t)(wg - a.full_tiles) * nsl + ks) * (256 * 256) +
                                     (size_t)(m - m0) * 256 + (nb - n0);
                        *reinterpret_cast<f32x4_t*>(dst) = f32x4_t{v[0], v[1], v[2], v[3]};
                    } else {
                        store_quad<EPI>(a, m, nb, v);
                    }
                }
    }
}

template <int EPI, bool CONV>
void launch_pp(const GemmArgs& a, hipStream_t s, int full_tiles, int ksplit) {
    const int tiles = ((a.M + 255) / 256) * ((a.N + 255) / 256);
    GemmArgs b = a;
    b.full_tiles = full_tiles < 0 ? tiles : full_tiles;
    b.ksplit = ksplit;
    const int tail = tiles - b.full_tiles;
    hipLaunchKernelGGL((gemm_pp_kernel<EPI, CONV>), dim3(b.full_tiles + tail * ksplit), dim3(512), 0, s, b);
    if (tail > 0) hipLaunchKernelGGL((splitk_reduce_kernel<EPI, 256, 256>), dim3(tail, SPLITK_RED_Y), dim3(256), 0, s, b);
}

}  // namespace

// 256x256 ping-pong tile: tiles [0, full_tiles) whole-K, the rest cut into ksplit K-slices (fp32 slices in a.partial,
// summed by a second launch).  full_tiles < 0: no slicing.  Requires K % 64 == 0 (conv: Cin % 64 == 0).
int launch_gemm256(const GemmArgs& a, hipStream_t s, int full_tiles, int ksplit) {
    if (a.K & 63) return -22;
    if (a.conv.mode != CONV_NONE) {
        switch (a.epi) {
            case EPI_NONE:  launch_pp<EPI_NONE, true>(a, s, full_tiles, ksplit); break;
            case EPI_RESID: launch_pp<EPI_RESID, true>(a, s, full_tiles, ksplit); break;
            default: return -22;
        }
    } else {
        switch (a.epi) {
            case EPI_NONE:   launch_pp<EPI_NONE, false>(a, s, full_tiles, ksplit); break;
            case EPI_RESID:  launch_pp<EPI_RESID, false>(a, s, full_tiles, ksplit); break;
            case EPI_SWIGLU: launch_pp<EPI_SWIGLU, false>(a, s, full_tiles, ksplit); break;
            case EPI_SILU:   launch_pp<EPI_SILU, false>(a, s, full_tiles, ksplit); break;
            case EPI_GELU:   launch_pp<EPI_GELU, false>(a, s, full_tiles, ksplit); break;
            case EPI_GEGLU:  launch_pp<EPI_GEGLU, false>(a, s, full_tiles, ksplit); break;
            default: return -22;
        }
    }
    EMU_CHECK_LAUNCH();
    return 0;
}
