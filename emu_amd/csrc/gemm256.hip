// 256(n) x 256(m) x 64(k) bf16 MFMA GEMM tile with a ping-pong phase schedule -- the MFMA-bound shapes of the path
// (LLaMA prefill, ViT blocks, the UNet's large GEMMs and implicit-GEMM convs):  C[m, n] = epilogue(sum_k A[m, k] W[n, k]).
//
// Why this tile: the 128x128 / 256x128 tiles of gemm.hip need 64 / 47 bytes per clock per CU from L2 at MFMA peak,
// which is the whole L2 -> LDS path (~56 B/clk/CU); 256x256 needs 31.  Why this schedule: with one barrier pair per
// k tile every wave of a workgroup loads, then computes, in lock step, and the matrix pipe idles during the loads.
//
// Structure (8 waves = 2 groups of 4, one wave of each group per SIMD; wave (wr, wc) owns a 128(n) x 64(m) output as
// 2 x 2 sub-tiles of 64(n) x 32(m), each 2 v_mfma_f32_32x32x16_bf16 accumulators: 128 accumulator registers):
//   * a k tile is staged as four 16 KiB UNITS -- P0, P1 (weight rows: the first / second 64 rows of every wave row),
//     Q0, Q1 (activation rows: the first / second 32 rows of every wave column) -- by LDS-DMA, 2 instructions of 1 KiB
//     per wave per unit, issued as `buffer_load_dwordx4 ... offen lds`: two SGPR descriptors (W, A), a 32-bit per-lane
//     offset and the k offset in soffset, so a piece costs s_mov m0 + the load and no VALU (global_load_lds with 64-bit
//     lane addresses measured 8-19 % slower end to end: profiles/r02_gemm_ab_v6_schedule_variants.log).  The LDS
//     bank-conflict swizzle (16-byte slot ^= (row >> 1) & 7) is applied to the per-lane SOURCE offset, LDS-DMA writes
//     lane-linear.  Conv taps outside the image use an offset beyond the descriptor's range: the hardware returns 0;
//   * a k tile is consumed in two PHASES of 16 MFMAs:
//         A(t): read P0, P1, Q0 (20 ds_read_b128) | stage Q1 of tile t+1        | MFMA (P0,Q0) (P1,Q0)
//         B(t): read Q1 (4)                       | stage P0, P1, Q0 of tile t+2 | MFMA (P0,Q1) (P1,Q1)
//     each phase = { ds_reads, LDS-DMA, counted vmcnt, lgkmcnt(0) } barrier { MFMAs, s_setprio 1 } barrier;
//   * group 1 runs one barrier behind group 0, so on every SIMD one wave is in its MFMA segment while its partner
//     reads LDS / issues DMA;
//   * ordering of LDS-DMA under the staggered barrier (cdna_hip_programming.md, "8-phase template"): a unit is waited
//     for (vmcnt(8): the two newest stage groups, 6 + 2 instructions, stay in flight) one phase before its first
//     ds_read, and overwritten one phase after its last ds_read (which the lgkmcnt(0) ahead of the barrier retired).
//     Every unit is in flight for two phases (~2k cycles).
// Measured on MI355X (profiles/r02_gemm256_*): 71 % matrix-pipe utilisation on the active CUs at the clock the chip
// sustains under MFMA load (1.9 GHz); variants with twice the barriers, LDS-DMA inside the MFMA segment or a single
// barrier per phase measured within +-5 %.
//
// Ragged M ("extension"): prompts are 256*j + a few rows (770 = 3*256 + 2 text+image tokens, the ViT's 1025 = 4*256 + cls),
// and a fourth / fifth row of tiles for 2 rows would cost a quarter of the GEMM.  When 0 < M mod 256 <= 16 the last row
// of tiles carries the remainder itself: one more 1 KiB piece per wave per k tile (unit QX, 64 LDS rows; 16 are read) and
// the remainder rows x this wave's wc-th 32 weight rows on FOUR v_mfma_f32_16x16x32_bf16 per k tile (two 16-row halves x two
// 32-wide k-steps: 64 MFMA cycles against the main tile's 1024, +6.25 %; rounds 2-3 used 32x32x16 on the weight fragments that
// were in registers anyway: +12.5 %, and M = 770 ran 7-10 % behind M = 768).  The 16-row fragments are a different lane layout
// from the 32-row ones, so the weight rows are read a second time (4 ds_read_b128) -- in phase A, the only phase in which the
// P units of the tile are readable (in phase B they are being overwritten with tile t+2) -- together with the remainder rows
// (2 reads), and the four MFMAs close phase A's segment.  QX of tile t+2 rides in B(t)'s stage group with P0, P1, Q0 of that
// tile (two phases in flight like every unit; waited for in B(t+1), read in A(t+2), its buffer last read in A(t)).  fp8
// operands keep the 32x32x64 form on the register-resident fragments (2 more MFMAs per k tile), in phase A as well.  Plain
// GEMM only (conv M is a multiple of 256 on this path).
// LDS: 2 k tiles x 64 KiB + 2 x 8 KiB = 144 KiB, one workgroup per CU.
//
// Replaces the same reference calls as gemm.hip (torch Linear / Conv2d on the ViT, LLaMA-prefill and UNet paths).
#include "gemm_tile.h"

using namespace emu_gemm;

int launch_gemm_w4(const GemmArgs& b, hipStream_t s, int grid, int fx);     // gemm_w4.hip

namespace {

constexpr int UNIT = 128 * 128;          // 128 LDS rows of 128 bytes
constexpr int BUFB = 4 * UNIT;           // one k tile
constexpr int QXB = 64 * 128;            // remainder rows of one k tile (8 pieces of 8 rows; 32 are read)
constexpr int SLICE = 288 * 256;         // fp32 elements of one K-slice of a tile (with its remainder rows)
enum { U_P0 = 0, U_Q0 = 1, U_Q1 = 2, U_P1 = 3 };

template <int V> struct IC { static constexpr int value = V; };

__device__ __forceinline__ void bar() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
}

// rows of tiles and remainder rows carried by the last one (0 = none / a ragged last tile instead)
__host__ __device__ inline int pp_tiles_m(int M, bool allow_ext, int& ext_rows) {
    const int tm = M >> 8, r = M & 255;
    ext_rows = (allow_ext && tm >= 1 && r > 0 && r <= 16) ? r : 0;
    return ext_rows ? tm : (M + 255) >> 8;
}

constexpr uint32_t OOB = 0x80000000u;    // beyond num_records of the descriptors below: the load returns zeros

// F8: both operands are OCP fp8 e4m3 bytes (a k tile is still 128 bytes per row = 128 elements), the MFMA is the gfx950
// block-scaled v_mfma_scale_f32_32x32x64_f8f6f4 with unit block scales (twice the bf16 rate), and the per-row scales of the
// two operands (a.a_scale[m] * a.w_scale[n]) multiply the fp32 sums ahead of the epilogue.
template <int EPI, bool CONV, bool F8, int FX = 0>
__global__ __launch_bounds__(512) void gemm_pp_kernel(const GemmArgs a) {
    __shared__ __attribute__((aligned(16))) char smem[2 * BUFB + 2 * QXB];
    const int tid = threadIdx.x, lane = tid & 63;
    EMU_TRACE_MARK(a.trace, 0);
    const uint32_t pfd = prefetch_lines(a.pf_ptr, a.pf_bytes, blockIdx.x * 512u + tid, gridDim.x * 512u);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int l31 = lane & 31, hi = lane >> 5;

    // workgroups [0, full_tiles): whole-K tiles; the rest: K-slices of the remaining tiles.  Both ranges go through the
    // XCD-aware bijective remap (block b runs on XCD b % 8; every XCD gets a contiguous run of the logical order), and the
    // logical order keeps the tiles_m row tiles of one weight column -- of one K-slice of it -- adjacent, so that the
    // workgroups reading the same weight bytes share one L2.  (Before round 3 the K-slices of a tile were dealt round-robin:
    // the row tiles of a slice sat on different XCDs and FETCH_SIZE showed the S=770 o / down weights fetched ~2.6x.)
    const int b = blockIdx.x;
    int ext_rows;
    constexpr uint32_t ESZ = F8 ? 1u : 2u;                          // bytes per element
    const int tiles_m = pp_tiles_m(a.M, !CONV, ext_rows);
    auto xcd_order = [](int i, int n) {                              // i-th block of n -> its place in the logical order
        const int xcd = i & 7, q8 = n >> 3, r8 = n & 7;
        return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (i >> 3);
    };
    int wg, ks = 0, nsl = 1;
    if (b < a.full_tiles) {
        wg = xcd_order(b, a.full_tiles);
        if (a.sup_m) {
            // 2-D blocks (launch_pp, unsliced plain GEMMs with many row tiles): an XCD's run of the logical order is one sup_m x sup_n
            // block of tiles walked row-first, so the 32 workgroups it runs at a time share sup_m row tiles of A and a few weight
            // tiles instead of one weight tile and 32 different row tiles (8192^3: 4.2 GB fetched for 0.27 GB of operands)
            const int per = a.sup_m * a.sup_n, blk = wg / per, r = wg - blk * per;
            const int bm = tiles_m / a.sup_m, bi = blk % bm, bj = blk / bm;
            wg = (bj * a.sup_n + r / a.sup_m) * tiles_m + bi * a.sup_m + r % a.sup_m;
        }
    } else {
        nsl = a.ksplit;
        const int rest = tiles_m * ((a.N + 255) >> 8) - a.full_tiles;
        const int l = xcd_order(b - a.full_tiles, rest * nsl);       // slice-major: slice ks of every remaining tile, then ks + 1
        ks = l / rest;
        wg = a.full_tiles + (l - ks * rest);
        if (a.slice_rr) {
            const int j = b - a.full_tiles;
            wg = a.full_tiles + j / nsl;
            ks = j - (wg - a.full_tiles) * nsl;
        }
    }
    const int tm = wg % tiles_m;
    const int n0 = (wg / tiles_m) << 8, m0 = tm << 8;
    // this workgroup also owns rows m0 + 256 .. M - 1 (never with the fused epilogues: gemm256_ok refuses that combination)
    const bool ext = !CONV && (FX == 0 || (FX & FX_ROPE) != 0) && ext_rows > 0 && tm == tiles_m - 1;

    // ---- LDS-DMA sources.  Instruction i (0, 1) of a unit fills LDS rows r = i*64 + srow, srow = wave*8 + lane/8, slot
    // lane%8 <- global chunk slot ^ ((r >> 1) & 7).  P unit s: row r holds weight row n0 + (r >> 6)*128 + s*64 + (r & 63);
    // Q unit s: row r holds activation row m0 + (r >> 5)*64 + s*32 + (r & 31); QX: row srow holds m0 + 256 + srow.
    // Each operand needs ONE per-lane byte offset (row srow of the tile, swizzled chunk); the row distance of a piece and
    // the k offset are wave-uniform and ride in soffset.  The descriptors end at the last row, so rows past N - 1 / M - 1
    // of a ragged tile read zeros (no clamping).
    const int srow = wave * 8 + (lane >> 3);
    const uint32_t sck = (uint32_t)(((lane & 7) ^ (((wave & 1) << 2) | (lane >> 4))) * 16);       // bytes
    const uint32_t vP = (uint32_t)(n0 + srow) * (uint32_t)a.ldw * ESZ + sck;
    const uint32_t vQ = (uint32_t)(m0 + (srow >> 5) * 64 + (srow & 31)) * (uint32_t)a.lda * ESZ + sck;     // plain GEMM
    uint32_t qpix[2][2];                             // CONV: tap mask and centre source pixel of every Q piece's row (ConvGeom)
    if constexpr (CONV) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int r = i * 64 + srow;
                int gm = m0 + (r >> 5) * 64 + s * 32 + (r & 31);
                gm = gm < a.M ? gm : a.M - 1;
                const int hw = a.conv.Hout * a.conv.Wout;
                const int pb = gm / hw, rr = gm - pb * hw;
                const int py = rr / a.conv.Wout, px = rr - py * a.conv.Wout;
                {
                    // fast gather (ConvGeom): 9-bit tap mask << 23 | index of the centre source pixel (< 2^23; nearest x2
                    // upsampling: centre (py / 2, px / 2) < 2^21, the parities of py, px in bits 22, 21)
                    const bool up = a.conv.mode == CONV_3X3_UP2;
                    const int cy = up ? py >> 1 : (a.conv.mode == CONV_3X3_S2 ? 2 * py : py);
                    const int cx = up ? px >> 1 : (a.conv.mode == CONV_3X3_S2 ? 2 * px : px);
                    uint32_t mask = 0;
#pragma unroll
                    for (int t = 0; t < 9; ++t) {
                        int yi, xi;
                        mask |= (uint32_t)conv_tap(a.conv, py, px, t / 3, t % 3, yi, xi) << t;
                    }
                    qpix[s][i] = (mask << 23) | (uint32_t)((pb * a.conv.Hin + cy) * a.conv.Win + cx) |
                                 (up ? (uint32_t)(((py & 1) << 22) | ((px & 1) << 21)) : 0u);
                }
            }
    }
    const uint32_t w_bytes = (uint32_t)a.N * (uint32_t)a.ldw * ESZ;
    const uint32_t a_bytes = CONV ? 0x7fffffffu : (uint32_t)a.M * (uint32_t)a.lda * ESZ;
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)a.W, 0, w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)a.A, 0, a_bytes, 0x00020000);
    auto dma = [&](const __amdgpu_buffer_rsrc_t& r, uint32_t voff, int soff, char* lds_wave_base) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, soff, 0, 0);
    };

    const int nk_all = F8 ? a.K >> 7 : a.K >> 6;                      // k tiles of 128 bytes per row
    const int kt0 = (int)((long)ks * nk_all / nsl);
    const int nk = (int)((long)(ks + 1) * nk_all / nsl) - kt0;

    auto stage = [&](auto uc, auto bc, int tau) {
        constexpr int U = decltype(uc)::value, BF = decltype(bc)::value;
        constexpr int S = (U == U_P1 || U == U_Q1) ? 1 : 0;
        char* base = smem + BF * BUFB + U * UNIT + wave * 1024;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if constexpr (U == U_P0 || U == U_P1) {
                dma(rW, vP, ((kt0 + tau) << 7) + (i * 128 + S * 64) * a.ldw * (int)ESZ, base + i * 8192);
            } else if constexpr (!CONV) {
                dma(rA, vQ, ((kt0 + tau) << 7) + (i * 128 + S * 32) * a.lda * (int)ESZ, base + i * 8192);
            } else {
                // implicit-GEMM gather: a 64-wide k tile lies inside one filter tap (Cin % 64 == 0)
                const uint32_t o = qpix[S][i];
                {                                      // (gemm256_ok: only problems the fast gather covers come here)
                    const int ktile = kt0 + tau;
                    const int tap = (ktile * a.conv.cpt_magic) >> 16, ci0 = (ktile - tap * a.conv.cpt) << 6;
                    const int ky = (tap * 11) >> 5, kx = tap - ky * 3;
                    uint32_t src;
                    if (a.conv.mode != CONV_3X3_UP2) {
                        src = (o & 0x7fffffu) + (uint32_t)((ky - 1) * a.conv.Win + (kx - 1));          // wave-uniform step
                    } else {                           // upsampled: the source step of a tap depends on the pixel's parity
                        const int dy = (int)(((o >> 22) & 1u) + ky - 1) >> 1, dx = (int)(((o >> 21) & 1u) + kx - 1) >> 1;
                        src = (o & 0x1fffffu) + (uint32_t)(dy * a.conv.Win + dx);
                    }
                    const uint32_t off = __umul24(src, (uint32_t)a.conv.Cin * 2u) + sck;
                    dma(rA, ((o >> (23 + tap)) & 1u) ? off : OOB, ci0 * 2, base + i * 8192);
                }
            }
        }
    };
    // QX row srow = m0 + 256 + srow = (row of vQ) + 256 - 32 * (wave >> 2): a wave-uniform distance
    auto stage_x = [&](auto bc, int tau) {
        constexpr int BF = decltype(bc)::value;
        if constexpr (!CONV)
            dma(rA, vQ, ((kt0 + tau) << 7) + (256 - 32 * (wave >> 2)) * a.lda * (int)ESZ, smem + 2 * BUFB + BF * QXB + wave * 1024);
    };

    // ---- fragment reads: lane (l31, hi) reads row base + l31, chunk (2*kk + hi) ^ swizzle(row), swizzle = (l31 >> 1) & 7
    // bf16: k step kk (16 wide) = chunk 2*kk + hi.  fp8: k step kk >> 1 (64 wide) = 32 bytes per lane = chunks
    // 4*(kk >> 1) + 2*hi + (kk & 1): pieces 2s, 2s+1 of a fragment form one MFMA operand.  The chunk index enters the byte
    // offset through an XOR (the swizzle), so the offset of k step kk is the offset of k step 0 with bits 4..6 flipped: ONE
    // loop-invariant lane offset (P units: wave row wr owns LDS rows wr*64 .. +63) and three v_xor per phase instead of four
    // registers (the kernel lives at the 256-VGPR edge; round 4's 16x16 remainder-row fragments needed the room)
    const int lp0 = l31 * 128 + ((((F8 ? hi << 1 : hi)) ^ ((l31 >> 1) & 7)) << 4) + wr * 8192;
    auto lpk = [](int base, int kk) { return base ^ (F8 ? (((kk >> 1) << 6) | ((kk & 1) << 4)) : (kk << 5)); };
    bf16x8_t pg[2][2][4], q0f[4], q1f[4], qxf[4];   // [P sub-tile][n fragment][k step]
    auto read_p = [&](bf16x8_t (&d)[2][4], const char* ub, int base) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) d[i][kk] = *reinterpret_cast<const bf16x8_t*>(ub + i * 4096 + lpk(base, kk));
    };
    // Q units: wave column wc owns LDS rows wc*32 .. +31; QX: rows 0 .. 31.  Same lane offsets as P, moved by a
    // wave-uniform distance that is kept out of loop-invariant registers.
    auto read_q = [&](bf16x8_t (&q)[4], const char* ub, int dist, int base) {
        asm volatile("" : "+s"(dist));
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) q[kk] = *reinterpret_cast<const bf16x8_t*>(ub + (lpk(base, kk) + dist));
    };
    const int dq = wc * 4096 - wr * 8192, dx = -wr * 8192;
    // remainder rows on 16x16x32 MFMAs (bf16): lane (l15, q4) reads LDS row base + l15, chunk (4 s + q4) ^ swizzle(row); the
    // swizzle (row >> 1) & 7 of rows R0 + 16 h + l15 (R0 a multiple of 32) is (l15 >> 1) & 7 for both halves h, and k-step s = 1 is
    // the same offset with bit 6 flipped: one lane offset serves the wave's weight rows (P unit wc >> 1, rows wr*64 + (wc&1)*32 ..)
    // and, moved by a wave-uniform distance, the remainder rows (QX rows 0 .. 15)
    const int l15 = lane & 15, q4 = lane >> 4;
    const int lx0 = (wr * 64 + (wc & 1) * 32 + l15) * 128 + ((q4 ^ ((l15 >> 1) & 7)) << 4);
    bf16x8_t ax[2][2], bx[2];                        // [16-row half][k-step], [k-step]
    auto read_x16 = [&](const char* pu, const char* qx) {
        int d0 = 0, d1 = 2048;
        asm volatile("" : "+s"(d0), "+s"(d1));
        ax[0][0] = *reinterpret_cast<const bf16x8_t*>(pu + (lx0 + d0));
        ax[0][1] = *reinterpret_cast<const bf16x8_t*>(pu + ((lx0 ^ 64) + d0));
        ax[1][0] = *reinterpret_cast<const bf16x8_t*>(pu + (lx0 + d1));
        ax[1][1] = *reinterpret_cast<const bf16x8_t*>(pu + ((lx0 ^ 64) + d1));
        bx[0] = *reinterpret_cast<const bf16x8_t*>(qx + lx0);
        bx[1] = *reinterpret_cast<const bf16x8_t*>(qx + (lx0 ^ 64));
    };

    f32x16_t acc[2][2][2], accx;                    // [P sub-tile][Q sub-tile][n fragment]; remainder rows (fp8 operands)
    f32x4_t accx16[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};   // remainder rows x the wave's wc-th weight rows, 16-row halves (bf16)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        accx[r] = 0.f;
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y)
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[x][y][i][r] = 0.f;
    }

    // wave-uniform validity of the sub-tiles (ragged last tiles): skipped MFMA clusters, clamped loads, predicated stores
    bool pv[2], qv[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        pv[s] = (n0 + wr * 128 + s * 64) < a.N;
        qv[s] = (m0 + wc * 64 + s * 32) < a.M;
    }
    auto mma = [&](int y, const bf16x8_t (&q)[4]) {
        if (qv[y] && pv[0]) {                          // pv[1] implies pv[0]; an invalid second half only wastes MFMAs
            if constexpr (!F8) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int x = 0; x < 2; ++x)
#pragma unroll
                        for (int i = 0; i < 2; ++i)
                            acc[x][y][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pg[x][i][kk], q[kk], acc[x][y][i], 0, 0, 0);
            } else {
                typedef int v4i_t __attribute__((ext_vector_type(4)));
                typedef int v8i_t __attribute__((ext_vector_type(8)));
                auto op = [](const bf16x8_t& lo, const bf16x8_t& hi_) {
                    return __builtin_shufflevector(__builtin_bit_cast(v4i_t, lo), __builtin_bit_cast(v4i_t, hi_), 0, 1, 2, 3, 4, 5, 6, 7);
                };
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    const v8i_t qb = op(q[2 * st], q[2 * st + 1]);
#pragma unroll
                    for (int x = 0; x < 2; ++x)
#pragma unroll
                        for (int i = 0; i < 2; ++i)
                            acc[x][y][i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(
                                op(pg[x][i][2 * st], pg[x][i][2 * st + 1]), qb, acc[x][y][i], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                }
            }
        }
    };
    auto mma_x = [&](auto xc, auto ic) {               // remainder rows x this wave's wc-th weight fragment
        constexpr int X = decltype(xc)::value, I = decltype(ic)::value;
        if constexpr (!F8) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) accx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pg[X][I][kk], qxf[kk], accx, 0, 0, 0);
        } else {
            typedef int v4i_t __attribute__((ext_vector_type(4)));
            auto op = [](const bf16x8_t& lo, const bf16x8_t& hi_) {
                return __builtin_shufflevector(__builtin_bit_cast(v4i_t, lo), __builtin_bit_cast(v4i_t, hi_), 0, 1, 2, 3, 4, 5, 6, 7);
            };
#pragma unroll
            for (int st = 0; st < 2; ++st)
                accx = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(op(pg[X][I][2 * st], pg[X][I][2 * st + 1]), op(qxf[2 * st], qxf[2 * st + 1]),
                                                                      accx, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        }
    };

    auto tile = [&](auto bc, int t) {
        constexpr int BF = decltype(bc)::value;
        const char* sb = smem + BF * BUFB;
        // ---- phase A
        int lb = lp0;
        asm volatile("" : "+v"(lb));                   // (keeps the three derived offsets from being hoisted into registers)
        read_q(q0f, sb + U_Q0 * UNIT, dq, lb);
        read_p(pg[0], sb + U_P0 * UNIT, lb);
        read_p(pg[1], sb + U_P1 * UNIT, lb);
        if (ext) {                                     // the remainder rows of this tile (QX: staged two phases ago, waited for in B(t-1))
            if constexpr (!F8) read_x16(sb + (wc >= 2 ? U_P1 : U_P0) * UNIT, smem + 2 * BUFB + BF * QXB - (wr * 64 + (wc & 1) * 32) * 128);
            else read_q(qxf, smem + 2 * BUFB + BF * QXB, dx, lb);
        }
        if (t + 1 < nk) {
            stage(IC<U_Q1>{}, IC<BF ^ 1>{}, t + 1);
            if (ext) wait_vmcnt<9>(); else wait_vmcnt<8>();
        } else wait_vmcnt<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        bar();
        __builtin_amdgcn_s_setprio(1);
        mma(0, q0f);
        if (ext) {
            if constexpr (!F8) {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int h = 0; h < 2; ++h) accx16[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ax[h][s2], bx[s2], accx16[h], 0, 0, 0);
            } else {
                if (wc == 0) mma_x(IC<0>{}, IC<0>{});
                else if (wc == 1) mma_x(IC<0>{}, IC<1>{});
                else if (wc == 2) mma_x(IC<1>{}, IC<0>{});
                else mma_x(IC<1>{}, IC<1>{});
            }
        }
        __builtin_amdgcn_s_setprio(0);
        bar();
        // ---- phase B
        lb = lp0;
        asm volatile("" : "+v"(lb));
        read_q(q1f, sb + U_Q1 * UNIT, dq, lb);
        if (t + 2 < nk) {
            if (ext) stage_x(IC<BF>{}, t + 2);
            stage(IC<U_P0>{}, IC<BF>{}, t + 2);
            stage(IC<U_P1>{}, IC<BF>{}, t + 2);
            stage(IC<U_Q0>{}, IC<BF>{}, t + 2);
            if (ext) wait_vmcnt<9>(); else wait_vmcnt<8>();
        } else if (t + 1 < nk) {
            wait_vmcnt<2>();
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        bar();
        __builtin_amdgcn_s_setprio(1);
        mma(1, q1f);
        __builtin_amdgcn_s_setprio(0);
        bar();
    };

    // prologue: (QX,) P0, P1, Q0 of tile 0 | Q1 of tile 0 | (QX,) P0, P1, Q0 of tile 1
    if (ext) stage_x(IC<0>{}, 0);
    stage(IC<U_P0>{}, IC<0>{}, 0);
    stage(IC<U_P1>{}, IC<0>{}, 0);
    stage(IC<U_Q0>{}, IC<0>{}, 0);
    stage(IC<U_Q1>{}, IC<0>{}, 0);
    if (nk > 1) {
        if (ext) stage_x(IC<1>{}, 1);
        stage(IC<U_P0>{}, IC<1>{}, 1);
        stage(IC<U_P1>{}, IC<1>{}, 1);
        stage(IC<U_Q0>{}, IC<1>{}, 1);
        if (ext) wait_vmcnt<9>(); else wait_vmcnt<8>();
    } else {
        wait_vmcnt<2>();
    }
    bar();
    prefetch_release(pfd);                          // (older than every tile load: complete behind the prologue's counted wait)
    EMU_TRACE_MARK(a.trace, 1);
    if (wr == 1) bar();                             // group 1 runs one barrier behind group 0
    for (int t = 0; t < nk; t += 2) {
        tile(IC<0>{}, t);
        if (t + 1 < nk) tile(IC<1>{}, t + 1);
    }
    if (wr == 0) bar();
    EMU_TRACE_MARK(a.trace, 2);
#ifdef EMU_TRACE
    struct TraceEnd { unsigned long long* t; __device__ ~TraceEnd() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); EMU_TRACE_MARK(t, 3); } } trace_end{a.trace};
#endif

    // ---- epilogue: accumulator (x, y, i): rows n = n0 + wr*128 + x*64 + i*32 + 8*g + 4*hi + e, column m = .. + l31
    auto emit = [&](int m, int nb, float (&v)[4], RowFx& fx, const QuadIn& q) {
        if constexpr (F8) {
            if (nsl == 1) {                            // K-sliced: pp_reduce_kernel scales the summed slices
                const float sa = a.a_scale[m];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= sa * a.w_scale[nb + e < a.N ? nb + e : a.N - 1];
            }
        }
        if (nsl > 1) {                                 // raw fp32 slice tile; pp_reduce_kernel applies the epilogue
            // (slab_rows: slices row-major over the whole output for the row-wise sum + RMSNorm, pp_rows_reduce_norm_kernel)
            float* dst = a.slab_rows ? a.partial + ((size_t)ks * a.M + m) * a.N + nb
                                     : a.partial + ((size_t)(wg - a.full_tiles) * nsl + ks) * SLICE + (size_t)(m - m0) * 256 + (nb - n0);
            *reinterpret_cast<f32x4_t*>(dst) = f32x4_t{v[0], v[1], v[2], v[3]};
        } else {
            store_quad<EPI, FX>(a, m, nb, v, fx, q);
        }
    };
    // ---- LLaMA prefill qkv projection: RoPE + KV append + V^T out of the epilogue (GemmArgs::rope_*; replaces the rope_kv and
    // transpose_v launches).  The tile -- 2 heads of q, k or v for 256 (+ remainder) rows of the one batch element -- is staged
    // row-major in LDS first: a head's two rotation halves d, d + 64 are then 128 bytes apart in a row, whichever accumulator
    // layout (main tile or 16x16 remainder rows) produced them; a lane rotates 8 + 8 elements of one row with rope_kv_kernel's
    // arithmetic (every product and the sum rounded to bf16) and stores 16-byte pieces: q to C, k to the cache rows of its
    // slot; v rows go to the cache as they are and then once more through the transposed staging to vt_out.
    if constexpr ((FX & FX_ROPE) != 0) {
        using StR = EpiStage<256, 256, 512>;
        using StT = EpiStageT<256, 256, 512>;
        const int HDc = a.rope_hl * 128, region = n0 / HDc;      // 0: q, 1: k, 2: v (3 * HDc % 256 == 0: launch_gemm checks)
        __syncthreads();                               // both wave groups are out of the loop
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y) {
                const int row = wc * 64 + y * 32 + l31;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int col = wr * 128 + x * 64 + i * 32 + 8 * g + 4 * hi;
                        u32x2 ov;
                        ov.x = packbf(acc[x][y][i][4 * g], acc[x][y][i][4 * g + 1]);
                        ov.y = packbf(acc[x][y][i][4 * g + 2], acc[x][y][i][4 * g + 3]);
                        *reinterpret_cast<u32x2*>(smem + StR::off(row, col)) = ov;
                    }
            }
        if (ext) {                                     // remainder rows: LDS rows 256 .. 271 behind the tile
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                u32x2 ov;
                ov.x = packbf(accx16[h][0], accx16[h][1]);
                ov.y = packbf(accx16[h][2], accx16[h][3]);
                *reinterpret_cast<u32x2*>(smem + StR::off(256 + l15, wr * 128 + wc * 32 + 16 * h + 4 * q4)) = ov;
            }
        }
        __syncthreads();
        const int nrows = ext ? 272 : 256;
        if (region < 2) {
            const int head0 = (n0 - region * HDc) >> 7;
            for (int idx = tid; idx < nrows * 16; idx += 512) {
                const int row = idx >> 4, hh = (idx >> 3) & 1, c = idx & 7, m = m0 + row;
                if (m >= a.M) continue;
                float x1[8], x2[8], cs[8], sn[8], o1[8], o2[8];
                unpack8(*reinterpret_cast<const u32x4*>(smem + StR::off(row, hh * 128 + c * 8)), x1);
                unpack8(*reinterpret_cast<const u32x4*>(smem + StR::off(row, hh * 128 + 64 + c * 8)), x2);
                const size_t po = (size_t)a.rope_pos[m] * 128 + c * 8;
                unpack8(ld16(a.rope_cos + po), cs);
                unpack8(ld16(a.rope_sin + po), sn);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    o1[j] = bfround(x1[j] * cs[j]) + bfround(-x2[j] * sn[j]);
                    o2[j] = bfround(x2[j] * cs[j]) + bfround(x1[j] * sn[j]);
                }
                bf16_t* dst = region == 0 ? a.C + (size_t)m * a.ldc + n0 + hh * 128 + c * 8
                                          : a.rope_kc + ((size_t)(head0 + hh) * a.rope_smax + a.rope_slot[m]) * 128 + c * 8;
                st16(dst, pack8(o1));
                st16(dst + 64, pack8(o2));
            }
        } else {
            const int head0 = (n0 - 2 * HDc) >> 7;
            for (int idx = tid; idx < nrows * 32; idx += 512) {
                const int row = idx >> 5, ch = idx & 31, m = m0 + row;
                if (m >= a.M) continue;
                st16(a.rope_vc + ((size_t)(head0 + (ch >> 4)) * a.rope_smax + a.rope_slot[m]) * 128 + (ch & 15) * 8,
                     *reinterpret_cast<const u32x4*>(smem + StR::off(row, ch * 8)));
            }
            __syncthreads();                           // the row-major tile has been read: the transposed one takes its place
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y) {
                    const int row = wc * 64 + y * 32 + l31;
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int col = wr * 128 + x * 64 + i * 32 + 8 * g + 4 * hi;
                            const uint32_t p0 = packbf(acc[x][y][i][4 * g], acc[x][y][i][4 * g + 1]);
                            const uint32_t p1 = packbf(acc[x][y][i][4 * g + 2], acc[x][y][i][4 * g + 3]);
                            *reinterpret_cast<bf16_t*>(smem + StT::off(col, row)) = (bf16_t)(p0 & 0xffffu);
                            *reinterpret_cast<bf16_t*>(smem + StT::off(col + 1, row)) = (bf16_t)(p0 >> 16);
                            *reinterpret_cast<bf16_t*>(smem + StT::off(col + 2, row)) = (bf16_t)(p1 & 0xffffu);
                            *reinterpret_cast<bf16_t*>(smem + StT::off(col + 3, row)) = (bf16_t)(p1 >> 16);
                        }
                }
            __syncthreads();
            StT::store(smem, a, m0, n0);               // one batch element: b = 0, key index = row index (launch_gemm checks)
            if (ext) {                                 // remainder rows: a few 2-byte stores per lane
                const int m = m0 + 256 + l15;
                if (m < a.M) {
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int n = n0 + wr * 128 + wc * 32 + 16 * h + 4 * q4 + e;
                            const uint32_t pk = packbf(accx16[h][e], 0.f);
                            a.vt_out[(size_t)(n - a.vt_col0) * a.vt_spad + m] = (bf16_t)(pk & 0xffffu);
                        }
                }
            }
        }
        return;
    }
    // Staged epilogue (gemm_tile.h::EpiStage): the 256 x 256 results leave through the (now dead) k-tile ring as whole rows
    constexpr bool GLU = EPI == EPI_SWIGLU || EPI == EPI_GEGLU;
    using Stage = EpiStage<256, GLU ? 128 : 256, 512>;
    bool staged = (a.stage & 1) != 0 && nsl == 1 && n0 + 256 <= a.N;
    bool vt_tile = false;                              // a whole tile of V columns: transposed staging (gemm_tile.h::EpiStageT)
    if constexpr ((FX & FX_VT) != 0) {
        vt_tile = staged && a.stage_vt && n0 >= a.vt_col0;
        staged = staged && (n0 + 256 <= a.vt_col0 || vt_tile);
    }
    using StageT = EpiStageT<256, 256, 512>;
    if (staged) {
        __syncthreads();                               // both wave groups are out of the loop
        if constexpr (EPI == EPI_RESID) Stage::load(smem, a.res, a.ldres, m0, n0, a.M);
    }
    RowFx rowfx[2];
    if constexpr ((FX & FX_LN) != 0) {                 // fused LayerNorm, consumer side: both rows of this lane in one batch of loads
        if (nsl == 1) {
            int mr[2];
#pragma unroll
            for (int y = 0; y < 2; ++y) { const int m = m0 + wc * 64 + y * 32 + l31; mr[y] = m < a.M ? m : a.M - 1; }
            LnRaw<2> raw;                              // one batch: a single L2 round trip at the head of the epilogue
            ln_rows_load<2>(a, mr, raw);
            ln_rows_finish<2>(a, raw, rowfx);
        }
    }
    if (staged) {
        if constexpr (EPI == EPI_RESID) {
            wait_vmcnt<0>();
            __syncthreads();
        }
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            QuadIn qin[2][4];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) quad_load_cols<EPI, FX>(a, n0 + wr * 128 + x * 64 + i * 32 + 8 * g + 4 * hi, qin[i][g]);
#pragma unroll
            for (int y = 0; y < 2; ++y) {
                const int row = wc * 64 + y * 32 + l31, m = m0 + row;
                if (m >= a.M && !((FX & FX_VT) != 0 && vt_tile)) continue;     // (see gemm.hip: pad keys of a V^T tile stay finite)
                RowFx& fx = rowfx[y];
                float sa = 1.f;
                if constexpr (F8) sa = a.a_scale[m < a.M ? m : a.M - 1];
                if (a.bias2) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int nb = n0 + wr * 128 + x * 64 + i * 32 + 8 * g + 4 * hi;
                            qin[i][g].bias2 = *reinterpret_cast<const u32x2*>(a.bias2 + (size_t)(m / a.rows_per_batch) * a.ld_bias2 + nb);
                        }
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int col = wr * 128 + x * 64 + i * 32 + 8 * g + 4 * hi;
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = acc[x][y][i][4 * g + e];
                        if constexpr (F8) {
                            const f32x4_t ws4 = *reinterpret_cast<const f32x4_t*>(a.w_scale + n0 + col);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] *= sa * ws4[e];
                        }
                        if constexpr (GLU) {
                            const u32x2 ov = quad_value<EPI, FX>(a, v, fx, qin[i][g]);
                            *reinterpret_cast<uint32_t*>(smem + Stage::off(row, col >> 1)) = ov.x;
                        } else if ((FX & FX_VT) != 0 && vt_tile) {
                            const u32x2 ov = quad_value<EPI, FX>(a, v, fx, qin[i][g]);
                            *reinterpret_cast<bf16_t*>(smem + StageT::off(col, row)) = (bf16_t)(ov.x & 0xffffu);
                            *reinterpret_cast<bf16_t*>(smem + StageT::off(col + 1, row)) = (bf16_t)(ov.x >> 16);
                            *reinterpret_cast<bf16_t*>(smem + StageT::off(col + 2, row)) = (bf16_t)(ov.y & 0xffffu);
                            *reinterpret_cast<bf16_t*>(smem + StageT::off(col + 3, row)) = (bf16_t)(ov.y >> 16);
                        } else {
                            u32x2* cell = reinterpret_cast<u32x2*>(smem + Stage::off(row, col));
                            if constexpr (EPI == EPI_RESID) qin[i][g].res = *cell;
                            *cell = quad_value<EPI, FX>(a, v, fx, qin[i][g]);
                        }
                    }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        if ((FX & FX_VT) != 0 && vt_tile) StageT::store(smem, a, m0, n0);
        else Stage::store(smem, a.C, a.ldc, m0, GLU ? n0 >> 1 : n0, a.M);
    } else {
    // Sub-tile x (64 columns) at a time: its column-only operands (bias / fused-LayerNorm vectors) are fetched once for both row
    // blocks y, a row block's residual / per-batch bias for all of its 8 quads before its first store (gemm_tile.h::QuadIn).
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        QuadIn qin[2][4];
        if (nsl == 1) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int nb = n0 + wr * 128 + x * 64 + i * 32 + 8 * g + 4 * hi;
                    if (nb < a.N) quad_load_cols<EPI, FX>(a, nb, qin[i][g]);
                }
        }
#pragma unroll
        for (int y = 0; y < 2; ++y) {
            const int m = m0 + wc * 64 + y * 32 + l31;
            if (m >= a.M) continue;
            RowFx& fx = rowfx[y];                      // statistics accumulate over both sub-tiles x of the row
            if (nsl == 1) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int nb = n0 + wr * 128 + x * 64 + i * 32 + 8 * g + 4 * hi;
                        if (nb < a.N) quad_load_row<EPI>(a, m, nb, qin[i][g]);
                    }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int nb = n0 + wr * 128 + x * 64 + i * 32 + 8 * g + 4 * hi;
                    if (nb >= a.N) continue;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[x][y][i][4 * g + e];
                    emit(m, nb, v, fx, qin[i][g]);
                }
        }
        __builtin_amdgcn_sched_barrier(0);             // keep the next sub-tile's loads from piling up (256-VGPR kernel)
    }
    }
    // fused LayerNorm, producer side: this wave's 128 columns of a row = one statistics slot, halves in lanes l / l + 32
    if constexpr ((FX & FX_STATS) != 0) {
        const int nslot = n0 + wr * 128;
        if (nsl == 1 && nslot < a.N) {
#pragma unroll
            for (int y = 0; y < 2; ++y) {
                const int m = m0 + wc * 64 + y * 32 + l31;
                const float sm = rowfx[y].rs + __shfl_xor(rowfx[y].rs, 32, 64), q = rowfx[y].rq + __shfl_xor(rowfx[y].rq, 32, 64);
                if (hi == 0 && m < a.M)
                    *reinterpret_cast<f32x2_t*>(a.row_stats_out + ((size_t)(nslot / LN_SLOT_COLS) * a.M + m) * 2) = f32x2_t{sm, q};
            }
        }
    }
    if (ext) {
        if constexpr (!F8) {
            // 16x16 results: lane (l15, q4) holds remainder row l15, columns 16 h + 4 q4 .. + 3 of the wave's wc-th 32 weight rows
            const int m = m0 + 256 + l15;
            if (m < a.M) {
                QuadIn qx[2];
                if (nsl == 1) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int nb = n0 + wr * 128 + wc * 32 + 16 * h + 4 * q4;
                        if (nb < a.N) { quad_load_cols<EPI, FX>(a, nb, qx[h]); quad_load_row<EPI>(a, m, nb, qx[h]); }
                    }
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int nb = n0 + wr * 128 + wc * 32 + 16 * h + 4 * q4;
                    if (nb >= a.N) continue;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = accx16[h][e];
                    RowFx fx;                          // remainder rows never carry the fused-LayerNorm features (gemm256_ok)
                    emit(m, nb, v, fx, qx[h]);
                }
            }
        } else {
        const int m = m0 + 256 + l31;
        if (m < a.M) {
            QuadIn qx[4];
            if (nsl == 1) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int nb = n0 + wr * 128 + wc * 32 + 8 * g + 4 * hi;
                    if (nb < a.N) { quad_load_cols<EPI, FX>(a, nb, qx[g]); quad_load_row<EPI>(a, m, nb, qx[g]); }
                }
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nb = n0 + wr * 128 + wc * 32 + 8 * g + 4 * hi;
                if (nb >= a.N) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = accx[4 * g + e];
                RowFx fx;                              // remainder rows never carry the fused-LayerNorm features (gemm256_ok)
                emit(m, nb, v, fx, qx[g]);
            }
        }
        }
    }
}

// second launch of a K-sliced run: sum the slices of every tile (288 x 256 fp32 each: the tile and its remainder rows)
template <int EPI, int FX = 0>
__global__ __launch_bounds__(256) void pp_reduce_kernel(const GemmArgs a) {
    int ext_rows;
    const int tiles_m = pp_tiles_m(a.M, a.conv.mode == CONV_NONE, ext_rows);
    const int wg = a.full_tiles + blockIdx.x;
    const int tm = wg % tiles_m;
    const int n0 = (wg / tiles_m) << 8, m0 = tm << 8;
    const int rows = (ext_rows > 0 && tm == tiles_m - 1) ? 288 : 256;
    const float* base = a.partial + (size_t)blockIdx.x * a.ksplit * SLICE;
    const int per = rows * 64 / SPLITK_RED_Y;
    for (int q = blockIdx.y * per + threadIdx.x; q < (blockIdx.y + 1) * per; q += 256) {      // per % 64 == 0: whole waves
        const int lm = q >> 6, lq = q & 63;
        const int m = m0 + lm, nb = n0 + lq * 4;
        const bool ok = m < a.M && nb < a.N;
        RowFx fx;
        if (ok) {
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            for (int ks = 0; ks < a.ksplit; ++ks) {
                const f32x4_t t = *reinterpret_cast<const f32x4_t*>(base + (size_t)ks * SLICE + (size_t)lm * 256 + lq * 4);
                v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3];
            }
            if (a.a_scale) {
                const float sa = a.a_scale[m];
                for (int e = 0; e < 4; ++e) v[e] *= sa * a.w_scale[nb + e < a.N ? nb + e : a.N - 1];
            }
            QuadIn qi;
            quad_load_cols<EPI, FX>(a, nb, qi);
            quad_load_row<EPI>(a, m, nb, qi);
            if constexpr ((FX & FX_LN) != 0) ln_row_stats(a, m, fx);
            store_quad<EPI, FX>(a, m, nb, v, fx, qi);
        }
        if constexpr ((FX & FX_STATS) != 0) emit_row_stats32(a, m, nb, ok, fx);
    }
}

template <int EPI, bool CONV, bool F8 = false>
void launch_pp(const GemmArgs& a, hipStream_t s, int full_tiles, int ksplit) {
    const int tiles = gemm256_tiles(a);
    GemmArgs b = a;
    b.full_tiles = full_tiles < 0 ? tiles : full_tiles;
    b.ksplit = ksplit;
    b.trace = emu_gemm_trace_get();
    b.stage = stage_ok(b) && !(emu_gemm_tune_get() & 8);
    b.stage_vt = b.stage && stage_vt_ok(b, 256, 256) && !(emu_gemm_tune_get() & (1 << 14));
    if (!(emu_gemm_tune_get() & (1 << 23))) b.stage |= 2;            // four-wave tile: fp32 K-slices leave through LDS too (bit 23: A/B)
    // XCD-aware 2-D tile blocks (gemm.hip::launch_cfg has the rule): unsliced plain GEMMs whose tile count splits evenly over the 8
    // XCDs; the implicit-GEMM convs keep the column-major strips (one weight tile of K = 9 Cin per strip is what their L2 can hold)
    b.sup_m = b.sup_n = 0;
    if (!CONV && b.full_tiles == tiles && tiles % 8 == 0 && !(emu_gemm_tune_get() & (1 << 17)) && (tiles > 256 || !(emu_gemm_tune_get() & (1 << 18)))) {
        int ext_rows;
        const int tm = pp_tiles_m(a.M, true, ext_rows), tn = tiles / tm, per = tiles / 8;
        const int cur_cols = (per + tm - 1) / tm + ((per % tm) ? 1 : 0);
        long best = (long)(per < tm ? per : tm) + (long)(cur_cols < tn ? cur_cols : tn);
        for (int sm = 1; sm <= tm; ++sm) {
            if (tm % sm || per % sm || tn % (per / sm)) continue;
            const long cost = (long)sm + (long)(per / sm);
            if (cost < best) { best = cost; b.sup_m = sm; b.sup_n = per / sm; }
        }
    }
    const int tail = tiles - b.full_tiles;
    const int fx = gemm_fx(b);
    // the main launch: the four-wave tile on its LDS ring (gemm_w4.hip) for bf16 operands; this file's eight-wave ping-pong tile for
    // fp8 operands and, as the A/B twin, under emu_gemm_tune bit 21.  Same tile order, K-slices and slab layout: the reduce launches
    // below serve both.
    // Where the four-wave tile is taken (same-run A/B, profiles/r06_gemm_w4_*.log): plain GEMMs whose tiles all lie inside N (they
    // leave through LDS: bf16 results and fp32 K-slices alike; a tile that reaches past N takes the direct path, whose scattered
    // stores its one wave per SIMD issues at half the ping-pong tile's rate) with a light epilogue: nothing overlaps the dependent
    // VALU chains of an erf on one wave per SIMD, and behind a GELU / GEGLU epilogue (ViT fc1, the UNet's GEGLU with its folded
    // LayerNorm) the faster loop loses in the model (same-run: ViT encode 15.93 vs 15.39 ms, denoise step 26.64 vs 25.11 ms with those
    // launches on it; LLaMA prefill S = 770 49.07 vs 51.35 ms).  K-slices shorter than 16 k tiles do not amortise the ring's prologue
    // (UNet 32^2 attn-out 172 vs 198 TFLOP/s), the implicit-GEMM convs' per-row bias loads cost it more (661 vs 739).
    // emu_gemm_tune bit 22 takes it wherever it is instantiated (tests).
    constexpr bool ACT = EPI == EPI_GELU || EPI == EPI_GEGLU;
    const int tune = emu_gemm_tune_get();
    const bool w4 = !F8 && !(tune & (1 << 21)) &&
                    ((tune & (1 << 22)) || (!CONV && (a.N & 255) == 0 && !ACT && !(gemm_fx(b) & FX_LN) &&
                                            (tail == 0 || (a.K >> 6) / ksplit >= 16)));
    const int grid = b.full_tiles + tail * ksplit;
    if (fx & FX_ROPE) {                                 // launch_gemm: EPI_NONE, unsliced, bf16 (launch_v2 checks the plan)
        if constexpr (!CONV && !F8 && EPI == EPI_NONE) {
            if (w4) launch_gemm_w4(b, s, b.full_tiles, fx);
            else hipLaunchKernelGGL((gemm_pp_kernel<EPI_NONE, false, false, FX_ROPE | FX_VT>), dim3(b.full_tiles), dim3(512), 0, s, b);
        }
        return;
    }
    if (fx) {                                           // gemm256_ok: bf16 plain GEMM; launch_gemm: an instantiated (epi, mask) pair
        if constexpr (!CONV && !F8) {
            gemm_fx_dispatch<EPI>(fx, [&](auto m) {
                constexpr int FXM = decltype(m)::value;
                if (w4) launch_gemm_w4(b, s, grid, fx);
                else hipLaunchKernelGGL((gemm_pp_kernel<EPI, CONV, F8, FXM>), dim3(grid), dim3(512), 0, s, b);
                if (tail > 0) hipLaunchKernelGGL((pp_reduce_kernel<EPI, FXM>), dim3(tail, SPLITK_RED_Y), dim3(256), 0, s, b);
            });
        }
        return;
    }
    if (w4) launch_gemm_w4(b, s, grid, 0);
    else hipLaunchKernelGGL((gemm_pp_kernel<EPI, CONV, F8>), dim3(grid), dim3(512), 0, s, b);
    if (tail > 0 && b.slab_rows) launch_rows_reduce_norm(b, s);     // launch_v2: every tile sliced, bf16, N <= 16384 (gemm_tile.h)
    else if (tail > 0) hipLaunchKernelGGL((pp_reduce_kernel<EPI>), dim3(tail, SPLITK_RED_Y), dim3(256), 0, s, b);
}

}  // namespace

int gemm256_tiles(const GemmArgs& a) {
    int ext_rows;
    return pp_tiles_m(a.M, a.conv.mode == CONV_NONE, ext_rows) * ((a.N + 255) / 256);
}

// operand extents the 32-bit descriptor offsets can address, k tiles of 64
bool gemm256_ok(const GemmArgs& a) {
    if (a.conv.mode != CONV_NONE && !a.conv.cpt_magic) return false;   // the gather here is the fast form only (ConvGeom)
    if (a.a_scale) {                                   // fp8 operands: k tiles of 128 elements, plain GEMM only
        if ((a.K & 127) || a.conv.mode != CONV_NONE || !a.w_scale) return false;
        return (size_t)a.N * a.ldw < 0x7fffffffull && (size_t)a.M * a.lda < 0x7fffffffull;
    }
    if (a.K & 63) return false;
    if (gemm_fx(a)) {                                  // the fused epilogues: bf16 plain GEMM, no remainder-row accumulator
        if (a.a_scale || a.conv.mode != CONV_NONE) return false;
        int ext_rows;
        pp_tiles_m(a.M, a.conv.mode == CONV_NONE, ext_rows);
        if (ext_rows && !a.rope_cos) return false;     // (the RoPE epilogue stages its remainder rows with the tile)
    }
    if (a.conv.mode != CONV_NONE && (a.conv.Hout > 2047 || a.conv.Wout > 2047 || a.M / (a.conv.Hout * a.conv.Wout) > 1023))
        return false;                                  // packed pixel coordinates of the gather
    const size_t wb = (size_t)a.N * a.ldw * 2;
    const size_t ab = a.conv.mode != CONV_NONE
                          ? (size_t)(a.M / (a.conv.Hout * a.conv.Wout)) * a.conv.Hin * a.conv.Win * a.conv.Cin * 2
                          : (size_t)a.M * a.lda * 2;
    return wb < 0x7fffffffull && ab < 0x7fffffffull;
}

// 256x256 ping-pong tile: tiles [0, full_tiles) whole-K, the rest cut into ksplit K-slices (fp32 slices of
// EMU_GEMM256_SLICE_FLOATS each in a.partial, summed by a second launch).  full_tiles < 0: no slicing.
int launch_gemm256(const GemmArgs& a, hipStream_t s, int full_tiles, int ksplit) {
    if (!gemm256_ok(a)) return -22;
    if (a.a_scale) {
        switch (a.epi) {
            case EPI_NONE:   launch_pp<EPI_NONE, false, true>(a, s, full_tiles, ksplit); break;
            case EPI_RESID:  launch_pp<EPI_RESID, false, true>(a, s, full_tiles, ksplit); break;
            case EPI_SWIGLU: launch_pp<EPI_SWIGLU, false, true>(a, s, full_tiles, ksplit); break;
            case EPI_GELU:   launch_pp<EPI_GELU, false, true>(a, s, full_tiles, ksplit); break;
            case EPI_GEGLU:  launch_pp<EPI_GEGLU, false, true>(a, s, full_tiles, ksplit); break;
            default: return -22;
        }
    } else if (a.conv.mode != CONV_NONE) {
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
