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

// VAR 0: the schedule above.  VAR 1: balanced LDS reads -- unit order Q0, P0, Q1, P1; unit s is read in phase s - 1
// (Q0 of the NEXT k tile in phase 3, into a second register set), staged in phase s - 7 (5 phases in flight, vmcnt(10)).
// ABL (timing ablations, results invalid): 1 no LDS-DMA in the loop, 2 no ds_reads in the loop, 4 no MFMAs.
template <int EPI, bool CONV, int VAR = 0, int ABL = 0>
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

    constexpr int SCHED = VAR == 4 ? 2 : VAR;         // VAR 4: schedule 2 with buffer_load ... lds instead of global_load_lds
    constexpr bool BUFL = VAR >= 4 && !CONV;
    // VAR 4: the same sources as 32-bit byte offsets into two buffer descriptors (SGPRs); the k offset rides in soffset
    uint32_t oP[2][2], oQ[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            oP[s][i] = (uint32_t)((const char*)gP[s][i] - (const char*)a.W);
            oQ[s][i] = (uint32_t)((const char*)gQ[s][i] - (const char*)a.A);
        }
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)a.W, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)a.A, 0, 0x7fffffff, 0x00020000);
    auto blds16 = [&](const __amdgpu_buffer_rsrc_t& r, uint32_t voff, int soff, char* lds_wave_base) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, soff, 0, 0);
    };

    const int nk_all = a.K >> 6;
    const int kt0 = (int)((long)ks * nk_all / nsl);
    const int nk = (int)((long)(ks + 1) * nk_all / nsl) - kt0;

    auto stage = [&](auto uc, auto bc, int tau) {
        constexpr int U = decltype(uc)::value, BF = decltype(bc)::value;
        if constexpr (ABL & 1) { if (tau > 1) return; }
        constexpr int S = (U == U_P1 || U == U_Q1) ? 1 : 0;
        const int k0 = (kt0 + tau) << 6;
        char* base = smem + BF * BUFB + U * UNIT + wave * 1024;
        if constexpr (BUFL) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if constexpr (U == U_P0 || U == U_P1) blds16(rW, oP[S][i], k0 * 2, base + i * 8192);
                else blds16(rA, oQ[S][i], k0 * 2, base + i * 8192);
            }
        } else if constexpr (U == U_P0 || U == U_P1) {
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
    bf16x8_t pf[2][4], q0f[4], q1f[4], q0g[4];
    bool abl_first = true;
    auto read_p = [&](const char* ub) {
        if constexpr (ABL & 2) {
            if (!abl_first) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) asm volatile("" : "+v"(pf[i][kk]));
                return;
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) pf[i][kk] = *reinterpret_cast<const bf16x8_t*>(ub + i * 4096 + lp[kk]);
    };
    auto read_q = [&](bf16x8_t (&q)[4], const char* ub) {
        if constexpr (ABL & 2) {
            if (!abl_first) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) asm volatile("" : "+v"(q[kk]));
                return;
            }
        }
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
                for (int i = 0; i < 2; ++i) {
                    if constexpr (ABL & 4) asm volatile("" :: "v"(pf[i][kk]), "v"(q[kk]));
                    else c[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf[i][kk], q[kk], c[i], 0, 0, 0);
                }
            __builtin_amdgcn_s_setprio(0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    if constexpr (SCHED == 0) {
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
            abl_first = false;
            if (t + 1 < nk) tile(IC<1>{}, t + 1);
        }
        if (wr == 0) bar();
    } else if constexpr (SCHED == 2) {
        // VAR 2: two fat phases per k tile (16 MFMAs each, half the barriers).
        //   phase A(t): read P0, P1, Q0 of tile t (20 ds_reads); stage Q1 of tile t+1;      MFMA (P0, Q0), (P1, Q0)
        //   phase B(t): read Q1 of tile t (4 ds_reads);          stage P0, P1, Q0 of t+2;   MFMA (P0, Q1), (P1, Q1)
        // Every phase waits lgkmcnt(0) before its first barrier, so a unit may be overwritten one phase after its
        // ds_reads; vmcnt(8) after the stage leaves the two newest stage groups (6 + 2 instructions) in flight: every
        // unit is in flight for two phases (~2k cycles).
        bf16x8_t pg[2][2][4];                          // [P sub-tile][n fragment][k step]
        auto read_p2 = [&](bf16x8_t (&d)[2][4], const char* ub) {
            if constexpr (ABL & 2) {
                if (!abl_first) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) asm volatile("" : "+v"(d[i][kk]));
                    return;
                }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) d[i][kk] = *reinterpret_cast<const bf16x8_t*>(ub + i * 4096 + lp[kk]);
        };
        auto mma2 = [&](int y, const bf16x8_t (&q)[4]) {
            __builtin_amdgcn_sched_barrier(0);
            if (qv[y] && pv[0]) {                      // pv[1] implies pv[0]; an invalid second half only wastes MFMAs
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int x = 0; x < 2; ++x)
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            if constexpr (ABL & 4) asm volatile("" :: "v"(pg[x][i][kk]), "v"(q[kk]));
                            else acc[x][y][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pg[x][i][kk], q[kk], acc[x][y][i], 0, 0, 0);
                        }
                __builtin_amdgcn_s_setprio(0);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        auto tile2 = [&](auto bc, int t) {
            constexpr int BF = decltype(bc)::value;
            const char* sb = smem + BF * BUFB;
            // phase A
            read_q(q0f, sb + U_Q0 * UNIT);
            read_p2(pg[0], sb + U_P0 * UNIT);
            read_p2(pg[1], sb + U_P1 * UNIT);
            if (t + 1 < nk) { stage(IC<U_Q1>{}, IC<BF ^ 1>{}, t + 1); wait_vmcnt<8>(); } else wait_vmcnt<0>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            bar();
            mma2(0, q0f);
            bar();
            // phase B
            read_q(q1f, sb + U_Q1 * UNIT);
            if (t + 2 < nk) {
                stage(IC<U_P0>{}, IC<BF>{}, t + 2);
                stage(IC<U_P1>{}, IC<BF>{}, t + 2);
                stage(IC<U_Q0>{}, IC<BF>{}, t + 2);
                wait_vmcnt<8>();
            } else if (t + 1 < nk) {
                wait_vmcnt<2>();
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            bar();
            mma2(1, q1f);
            bar();
        };
        stage(IC<U_P0>{}, IC<0>{}, 0);
        stage(IC<U_P1>{}, IC<0>{}, 0);
        stage(IC<U_Q0>{}, IC<0>{}, 0);
        stage(IC<U_Q1>{}, IC<0>{}, 0);
        if (nk > 1) {
            stage(IC<U_P0>{}, IC<1>{}, 1);
            stage(IC<U_P1>{}, IC<1>{}, 1);
            stage(IC<U_Q0>{}, IC<1>{}, 1);
            wait_vmcnt<8>();
        } else {
            wait_vmcnt<2>();
        }
        bar();
        if (wr == 1) bar();
        for (int t = 0; t < nk; t += 2) {
            tile2(IC<0>{}, t);
            abl_first = false;
            if (t + 1 < nk) tile2(IC<1>{}, t + 1);
        }
        if (wr == 0) bar();
    } else if constexpr (SCHED == 3) {
        // VAR 3 = VAR 2 with the LDS-DMA issued INSIDE the MFMA segments (between the k steps of the 16-MFMA cluster,
        // where the wave only waits for the matrix pipe), so the load segment is ds_reads + waits only.
        //   A(t): read P0, P1, Q0 (t); wait Q1(t);            MFMA (P0,Q0), (P1,Q0)  + stage Q1(t+1)
        //   B(t): read Q1 (t);         wait P0,P1,Q0 (t+1);   MFMA (P0,Q1), (P1,Q1)  + stage P0, P1, Q0 (t+2)
        // A group staging in MFMA(q) runs beside the other group's reads of phase q / q+1, so a unit staged in phase q
        // must have been read last in phase <= q - 1: Q1(t-1) was (B(t-1)), P0/P1/Q0(t) were (A(t)).
        bf16x8_t pg[2][2][4];
        auto read_p2 = [&](bf16x8_t (&d)[2][4], const char* ub) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) d[i][kk] = *reinterpret_cast<const bf16x8_t*>(ub + i * 4096 + lp[kk]);
        };
        auto mma_k = [&](int y, const bf16x8_t (&q)[4], int kk) {
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    acc[x][y][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pg[x][i][kk], q[kk], acc[x][y][i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        };
        auto tile3 = [&](auto bc, int t) {
            constexpr int BF = decltype(bc)::value;
            const char* sb = smem + BF * BUFB;
            // ---- phase A
            read_q(q0f, sb + U_Q0 * UNIT);
            read_p2(pg[0], sb + U_P0 * UNIT);
            read_p2(pg[1], sb + U_P1 * UNIT);
            if (t + 1 < nk) wait_vmcnt<6>(); else wait_vmcnt<0>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            bar();
            {
                const bool st = t + 1 < nk, on = qv[0] && pv[0];
                __builtin_amdgcn_s_setprio(1);
                if (on) mma_k(0, q0f, 0);
                if (on) mma_k(0, q0f, 1);
                if (st) stage(IC<U_Q1>{}, IC<BF ^ 1>{}, t + 1);
                __builtin_amdgcn_sched_barrier(0);
                if (on) mma_k(0, q0f, 2);
                if (on) mma_k(0, q0f, 3);
                __builtin_amdgcn_s_setprio(0);
            }
            bar();
            // ---- phase B
            read_q(q1f, sb + U_Q1 * UNIT);
            if (t + 1 < nk) wait_vmcnt<2>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            bar();
            {
                const bool st = t + 2 < nk, on = qv[1] && pv[0];
                __builtin_amdgcn_s_setprio(1);
                if (on) mma_k(1, q1f, 0);
                if (st) stage(IC<U_P0>{}, IC<BF>{}, t + 2);
                __builtin_amdgcn_sched_barrier(0);
                if (on) mma_k(1, q1f, 1);
                if (st) stage(IC<U_P1>{}, IC<BF>{}, t + 2);
                __builtin_amdgcn_sched_barrier(0);
                if (on) mma_k(1, q1f, 2);
                if (st) stage(IC<U_Q0>{}, IC<BF>{}, t + 2);
                __builtin_amdgcn_sched_barrier(0);
                if (on) mma_k(1, q1f, 3);
                __builtin_amdgcn_s_setprio(0);
            }
            bar();
        };
        stage(IC<U_P0>{}, IC<0>{}, 0);
        stage(IC<U_P1>{}, IC<0>{}, 0);
        stage(IC<U_Q0>{}, IC<0>{}, 0);
        stage(IC<U_Q1>{}, IC<0>{}, 0);
        if (nk > 1) {
            stage(IC<U_P0>{}, IC<1>{}, 1);
            stage(IC<U_P1>{}, IC<1>{}, 1);
            stage(IC<U_Q0>{}, IC<1>{}, 1);
            wait_vmcnt<8>();
        } else {
            wait_vmcnt<2>();
        }
        bar();
        if (wr == 1) bar();
        for (int t = 0; t < nk; t += 2) {
            tile3(IC<0>{}, t);
            if (t + 1 < nk) tile3(IC<1>{}, t + 1);
        }
        if (wr == 0) bar();
    } else if constexpr (SCHED == 5) {
        // VAR 5: ONE barrier per phase.  The two wave groups run the same phases in complementary order inside every
        // barrier interval p (phase 2t = A(t), 2t+1 = B(t) as in VAR 2):
        //     group 0:  MFMA(p)  ->  ds_reads of phase p+1  ->  LDS-DMA  -> wait -> barrier
        //     group 1:  ds_reads of phase p  ->  LDS-DMA  ->  MFMA(p)    -> wait -> barrier
        // so on every SIMD one wave computes while its partner loads without a barrier between the halves, and the
        // pipe hand-over inside an interval is arbitrated by the hardware, not by a barrier round trip.
        //   interval A(t): stage Q1(t+1); wait P0,P1,Q0(t+1) (vmcnt 2)      [group 0 reads them in interval B(t)]
        //   interval B(t): stage P0,P1,Q0(t+2); wait Q1(t+1) (vmcnt 6)      [group 0 reads it in interval A(t+1)]
        // A unit is staged in the interval after its last ds_read (group 1's, whose reads are consumed by MFMAs of the
        // same interval) and waited for one interval before its first ds_read (group 0's).
        bf16x8_t pg[2][2][4];
        auto read_p2 = [&](bf16x8_t (&d)[2][4], const char* ub) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) d[i][kk] = *reinterpret_cast<const bf16x8_t*>(ub + i * 4096 + lp[kk]);
        };
        auto mma2 = [&](int y, const bf16x8_t (&q)[4]) {
            __builtin_amdgcn_sched_barrier(0);
            if (qv[y] && pv[0]) {
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int x = 0; x < 2; ++x)
#pragma unroll
                        for (int i = 0; i < 2; ++i)
                            acc[x][y][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pg[x][i][kk], q[kk], acc[x][y][i], 0, 0, 0);
                __builtin_amdgcn_s_setprio(0);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        auto read_a = [&](const char* sb) {            // fragments of phase A
            read_q(q0f, sb + U_Q0 * UNIT);
            read_p2(pg[0], sb + U_P0 * UNIT);
            read_p2(pg[1], sb + U_P1 * UNIT);
            __builtin_amdgcn_sched_barrier(0);
        };
        auto read_b = [&](const char* sb) { read_q(q1f, sb + U_Q1 * UNIT); __builtin_amdgcn_sched_barrier(0); };
        auto stage_a = [&](auto bc, int t) {           // interval A(t), buffer BF = t & 1
            constexpr int BF = decltype(bc)::value;
            if (t + 1 < nk) stage(IC<U_Q1>{}, IC<BF ^ 1>{}, t + 1);
            __builtin_amdgcn_sched_barrier(0);
        };
        auto stage_b = [&](auto bc, int t) {
            constexpr int BF = decltype(bc)::value;
            if (t + 2 < nk) {
                stage(IC<U_P0>{}, IC<BF>{}, t + 2);
                stage(IC<U_P1>{}, IC<BF>{}, t + 2);
                stage(IC<U_Q0>{}, IC<BF>{}, t + 2);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        auto wait_a = [&](int t) { if (t + 1 < nk) wait_vmcnt<2>(); };
        auto wait_b = [&](int t) { if (t + 2 < nk) wait_vmcnt<6>(); else wait_vmcnt<0>(); };

        stage(IC<U_P0>{}, IC<0>{}, 0);
        stage(IC<U_P1>{}, IC<0>{}, 0);
        stage(IC<U_Q0>{}, IC<0>{}, 0);
        stage(IC<U_Q1>{}, IC<0>{}, 0);
        if (nk > 1) {
            stage(IC<U_P0>{}, IC<1>{}, 1);
            stage(IC<U_P1>{}, IC<1>{}, 1);
            stage(IC<U_Q0>{}, IC<1>{}, 1);
            wait_vmcnt<8>();
        } else {
            wait_vmcnt<2>();
        }
        bar();
        if (wr == 0) {
            read_a(smem);                              // "interval -1": group 0 runs one phase of reads ahead
            if (nk > 1) wait_vmcnt<6>(); else wait_vmcnt<0>();
            bar();
            auto tile5 = [&](auto bc, int t) {
                constexpr int BF = decltype(bc)::value;
                const char* sb = smem + BF * BUFB;
                mma2(0, q0f);
                read_b(sb);
                stage_a(bc, t);
                wait_a(t);
                bar();
                mma2(1, q1f);
                if (t + 1 < nk) read_a(smem + (BF ^ 1) * BUFB);
                stage_b(bc, t);
                wait_b(t);
                bar();
            };
            for (int t = 0; t < nk; t += 2) {
                tile5(IC<0>{}, t);
                if (t + 1 < nk) tile5(IC<1>{}, t + 1);
            }
        } else {
            if (nk > 1) wait_vmcnt<6>(); else wait_vmcnt<0>();
            bar();
            auto tile5 = [&](auto bc, int t) {
                constexpr int BF = decltype(bc)::value;
                const char* sb = smem + BF * BUFB;
                read_a(sb);
                stage_a(bc, t);
                mma2(0, q0f);
                wait_a(t);
                bar();
                read_b(sb);
                stage_b(bc, t);
                mma2(1, q1f);
                wait_b(t);
                bar();
            };
            for (int t = 0; t < nk; t += 2) {
                tile5(IC<0>{}, t);
                if (t + 1 < nk) tile5(IC<1>{}, t + 1);
            }
        }
    } else {
        // VAR 1.  Unit s = 4*tile + u, u: 0 = Q0, 1 = P0, 2 = Q1, 3 = P1 (LDS slot of unit u: the U_* position of that
        // operand).  Phase p: read unit p + 1; stage unit p + 7; wait until unit p + 2 has landed.
        const int s_last = 4 * nk - 1;
        auto wait_tail = [&](int k) {                  // k = units that may stay in flight
            if (k >= 4) wait_vmcnt<8>();
            else if (k == 3) wait_vmcnt<6>();
            else if (k == 2) wait_vmcnt<4>();
            else if (k == 1) wait_vmcnt<2>();
            else wait_vmcnt<0>();
        };
        auto feed = [&](auto uc, auto bc, int p) {     // stage unit p + 7 (slot U of buffer BF), then the counted wait
            if (p + 7 <= s_last) { stage(uc, bc, (p + 7) >> 2); wait_vmcnt<10>(); }
            else wait_tail(s_last - p - 2);
        };
        auto tile1 = [&](auto bc, int t, bf16x8_t (&qc)[4], bf16x8_t (&qn)[4]) {
            constexpr int BF = decltype(bc)::value;
            const char* sb = smem + BF * BUFB;
            const int p = 4 * t;
            read_p(sb + U_P0 * UNIT);
            feed(IC<U_P1>{}, IC<BF ^ 1>{}, p);
            bar();
            mma(acc[0][0], qc, pv[0] && qv[0]);
            bar();
            read_q(q1f, sb + U_Q1 * UNIT);
            feed(IC<U_Q0>{}, IC<BF>{}, p + 1);
            bar();
            mma(acc[0][1], q1f, pv[0] && qv[1]);
            bar();
            read_p(sb + U_P1 * UNIT);
            feed(IC<U_P0>{}, IC<BF>{}, p + 2);
            bar();
            mma(acc[1][1], q1f, pv[1] && qv[1]);
            bar();
            if (t + 1 < nk) read_q(qn, smem + (BF ^ 1) * BUFB + U_Q0 * UNIT);
            feed(IC<U_Q1>{}, IC<BF>{}, p + 3);
            bar();
            mma(acc[1][0], qc, pv[1] && qv[0]);
            bar();
        };
        // prologue: units 0..6; units 0 (Q0) and 1 (P0) of k tile 0 landed; Q0 of k tile 0 -> registers
        stage(IC<U_Q0>{}, IC<0>{}, 0);
        stage(IC<U_P0>{}, IC<0>{}, 0);
        stage(IC<U_Q1>{}, IC<0>{}, 0);
        stage(IC<U_P1>{}, IC<0>{}, 0);
        if (nk > 1) {
            stage(IC<U_Q0>{}, IC<1>{}, 1);
            stage(IC<U_P0>{}, IC<1>{}, 1);
            stage(IC<U_Q1>{}, IC<1>{}, 1);
            wait_vmcnt<10>();
        } else {
            wait_vmcnt<4>();
        }
        bar();
        read_q(q0f, smem + U_Q0 * UNIT);
        if (wr == 1) bar();
        for (int t = 0; t < nk; t += 2) {
            tile1(IC<0>{}, t, q0f, q0g);
            abl_first = false;
            if (t + 1 < nk) tile1(IC<1>{}, t + 1, q0g, q0f);
        }
        if (wr == 0) bar();
    }

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

int g_pp_variant = 0;                   // emu_gemm_force_config(cfg | variant << 8): schedule variant / timing ablation

template <int EPI, bool CONV>
void launch_pp(const GemmArgs& a, hipStream_t s, int full_tiles, int ksplit) {
    const int tiles = ((a.M + 255) / 256) * ((a.N + 255) / 256);
    GemmArgs b = a;
    b.full_tiles = full_tiles < 0 ? tiles : full_tiles;
    b.ksplit = ksplit;
    const int tail = tiles - b.full_tiles;
    const dim3 grid(b.full_tiles + tail * ksplit), block(512);
    if constexpr (EPI == EPI_NONE && !CONV) {        // A/B variants are built for the plain GEMM only
        switch (g_pp_variant) {
            case 1:  hipLaunchKernelGGL((gemm_pp_kernel<EPI, CONV, 1, 0>), grid, block, 0, s, b); break;
            case 2:  hipLaunchKernelGGL((gemm_pp_kernel<EPI, CONV, 1, 1>), grid, block, 0, s, b); break;
            case 3:  hipLaunchKernelGGL((gemm_pp_kernel<EPI, CONV, 1, 2>), grid, block, 0, s, b); break;
            case 4:  hipLaunchKernelGGL((gemm_pp_kernel<EPI, CONV, 1, 3>), grid, block, 0, s, b); break;
            case 5:  hipLaunchKernelGGL((gemm_pp_kernel<EPI, CONV, 1, 4>), grid, block, 0, s, b); break;
            case 6:  hipLaunchKernelGGL((gemm_pp_kernel<EPI, CONV, 1, 7>), grid, block, 0, s, b); break;
            case 7:  hipLaunchKernelGGL((gemm_pp_kernel<EPI, CONV, 2, 0>), grid, block, 0, s, b); break;
            case 11: hipLaunchKernelGGL((gemm_pp_kernel<EPI, CONV, 3, 0>), grid, block, 0, s, b); break;
            case 12: hipLaunchKernelGGL((gemm_pp_kernel<EPI, CONV, 4, 0>), grid, block, 0, s, b); break;
            case 13: hipLaunchKernelGGL((gemm_pp_kernel<EPI, CONV, 5, 0>), grid, block, 0, s, b); break;
            case 8:  hipLaunchKernelGGL((gemm_pp_kernel<EPI, CONV, 2, 1>), grid, block, 0, s, b); break;
            case 9:  hipLaunchKernelGGL((gemm_pp_kernel<EPI, CONV, 2, 4>), grid, block, 0, s, b); break;
            case 10: hipLaunchKernelGGL((gemm_pp_kernel<EPI, CONV, 2, 7>), grid, block, 0, s, b); break;
            default: hipLaunchKernelGGL((gemm_pp_kernel<EPI, CONV, 0, 0>), grid, block, 0, s, b); break;
        }
    } else {
        hipLaunchKernelGGL((gemm_pp_kernel<EPI, CONV, 0, 0>), grid, block, 0, s, b);
    }
    if (tail > 0) hipLaunchKernelGGL((splitk_reduce_kernel<EPI, 256, 256>), dim3(tail, SPLITK_RED_Y), dim3(256), 0, s, b);
}

}  // namespace

void emu_gemm256_variant_set(int v) { g_pp_variant = v; }

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
