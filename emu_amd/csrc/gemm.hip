// bf16 MFMA GEMM for prefill / ViT / UNet shapes:  C[m, n] = epilogue( sum_k A[m, k] * W[n, k] )
// ("NT": both operands K-contiguous, i.e. torch.nn.functional.linear(A, W)), and the implicit-GEMM 3x3 convolution.
//
// CDNA4 design shared by every tile configuration: the WEIGHT tile is the MFMA A operand, so the accumulator registers
// of one lane run along n: 4 consecutive output columns per register quad -> 8-byte stores, bias / activation / SwiGLU
// pairs stay lane-local (gemm_tile.h::store_quad).  Operands go global -> LDS directly with global_load_lds (16 B per
// lane, no VGPR staging, no ds_write pass) into an LDS ring of 64-wide k tiles; the loads of later tiles stay in flight
// ACROSS the per-tile barrier (counted s_waitcnt vmcnt(N), raw s_barrier).  An LDS-DMA instruction writes lane-linear
// (wave base + lane*16), so the bank-conflict swizzle (128-byte rows, 16-byte slot ^= (row >> 1) & 7) is applied to the
// per-lane SOURCE address and undone by the same XOR on the ds_read side.  Workgroup ids are remapped so each XCD walks
// a contiguous range of tiles (weight tile reuse in its L2).  Chunks beyond K (K % 64 != 0) and implicit-GEMM conv taps
// outside the image read a 16-byte zero buffer.
//
// Tile configurations (launch_v2 picks per problem): 256x256 ping-pong (gemm256.hip) for the MFMA-bound shapes,
// 256(n) x 128(m), 128 x 128 (2 workgroups per CU), 128(n) x 64(m) with two k-groups of waves for few-tile problems;
// split-K (fp32 slices + a reduce launch) when a problem has fewer tiles than the chip has CUs.
//
// Replaces the torch Linear calls on the reference hot path: Emu2/emu/eva_vit.py:106,112,198,250 (ViT),
// transformers LlamaAttention/LlamaMLP reached from Emu2/emu/emu.py:133-138,213-229 (prefill),
// project_up/down emu.py:201,147; diffusers Conv2d / Linear of the UNet (Emu2/emu/diffusion.py:136-141).
// Algorithmic FLOPs = 2*M*N*K.
#include "gemm_tile.h"

using namespace emu_gemm;

namespace {

constexpr int BK = 64;

// Tile configuration: WN x WM waves, each owning NF x MF 32x32 accumulators; NSTG-deep LDS ring of 64-wide k tiles.
// KG > 1: KG groups of WN x WM waves split the four 16-wide k-steps of every tile between them (intra-workgroup split-K:
// twice the waves per SIMD for the same tile, partial accumulators summed through LDS in the epilogue).
template <int WN_, int WM_, int NF_, int MF_, int NSTG_, int KG_ = 1>
struct TileCfg {
    static constexpr int WN = WN_, WM = WM_, NF = NF_, MF = MF_, NSTG = NSTG_, KG = KG_;
    static constexpr int THREADS = WN * WM * KG * 64;
    // waves per SIMD the register allocation must leave room for: the 128 x 64 two-k-group tile lives on TWO workgroups per
    // CU (72 KiB of LDS each), i.e. 4 waves per SIMD = at most 128 VGPRs; the others run 2 waves per SIMD
    static constexpr int MINW = (KG == 2 && NSTG_ * (WN_ * NF_ + WM_ * MF_) * 32 * 128 <= 80 * 1024) ? 4 : 2;
    static_assert(KG == 1 || KG == 2, "k-groups: 1 or 2");
    static constexpr int BNv = WN * NF * 32, BMv = WM * MF * 32;
    static constexpr int ROWB = BK * 2;                                       // bytes per LDS row (128)
    static constexpr int W_BYTES = BNv * ROWB, A_BYTES = BMv * ROWB, ST_BYTES = W_BYTES + A_BYTES;
    static constexpr int NLW = (BNv * 8) / THREADS, NLA = (BMv * 8) / THREADS, LPT = NLW + NLA;   // LDS-DMA per thread per tile
    static_assert((BNv * 8) % THREADS == 0 && (BMv * 8) % THREADS == 0, "tile rows must split evenly over the waves");
    static_assert((THREADS / 64) % 2 == 0, "source swizzle assumes an even wave count");
    static_assert(NSTG * ST_BYTES <= 160 * 1024, "LDS ring exceeds 160 KiB");
    static_assert((NSTG - 2) * LPT <= 63, "vmcnt field is 6 bits");
    // conflict-free ds_read_b128 of MFMA fragments: XOR the 16-byte slot with (row >> 1) & 7 (16 consecutive rows then
    // cover all 64 banks exactly once)
    __device__ static __forceinline__ int off(int row, int chunk) { return row * ROWB + ((chunk ^ ((row >> 1) & 7)) << 4); }
};

// F8 (launch_gemm_fp8): both operands are OCP fp8 e4m3 bytes.  A k tile is still 128 bytes per LDS row -- now 128 elements --
// so staging, swizzle and ring are untouched; a k-step is 64 elements (32 bytes per lane: chunks 4 st + 2 hi, + 1) on the
// block-scaled v_mfma_scale_f32_32x32x64_f8f6f4 with unit block scales (twice the bf16 rate: the same MFMA time per k tile for
// twice the K, half the LDS-DMA bytes per FLOP -- the resource that bounds these tiles), and the per-row scales of the two
// operands (a_scale[m] * w_scale[n]) multiply the fp32 sums ahead of the epilogue.  Plain GEMM; of the fused epilogues the two that
// only look at the finished sums -- V^T stores and the cross-attention -- are available, the LayerNorm fold is not (it needs the
// un-normalised rows as the operand, and those quantise badly).
template <int EPI, bool CONV, class T, int FX = 0, bool F8 = false>
__global__ __launch_bounds__(T::THREADS, T::MINW) void gemm2_kernel(const GemmArgs a) {
    static_assert(!F8 || (!CONV && (FX & (FX_LN | FX_STATS)) == 0), "fp8 operands: plain GEMM, V^T / cross-attention epilogues only");
    constexpr int KSH = F8 ? 1 : 0;                     // element offsets -> 2-byte units of the bf16_t pointers below
    constexpr int NW = T::WN * T::WM * T::KG;           // waves per workgroup
    constexpr int RED_BYTES = T::KG > 1 ? T::WN * T::WM * T::NF * T::MF * 16 * 64 * 4 : 0;
    // row-statistics exchange between the two waves of a 128-column slot: behind the k-group sums where those exist (the ring is
    // dead by then), else 8 KiB of its own behind the ring (other waves may still be reading tiles when the first one arrives)
    constexpr int EX_BYTES = (FX & FX_STATS) != 0 ? T::WN * T::WM * T::MF * 64 * 8 : 0;
    // staged epilogue: the tile of results (bf16) sits behind the k-group sums (the residual tile arrives while they are summed)
    // (where both do not fit, the tile takes the k-group sums' place after one more barrier)
    constexpr int STAGE_BYTES = T::BMv * T::BNv * 2;
    constexpr bool STAGE_BEHIND = RED_BYTES + STAGE_BYTES + EX_BYTES <= 160 * 1024;
    constexpr int STAGE_OFF = STAGE_BEHIND ? RED_BYTES : 0;
    constexpr int EX_OFF = STAGE_OFF + STAGE_BYTES > T::NSTG * T::ST_BYTES ? STAGE_OFF + STAGE_BYTES : T::NSTG * T::ST_BYTES;
    constexpr int RING_BYTES = T::NSTG * T::ST_BYTES > STAGE_OFF + STAGE_BYTES ? T::NSTG * T::ST_BYTES : STAGE_OFF + STAGE_BYTES;
    constexpr int SMEM_BYTES = RING_BYTES > EX_OFF + EX_BYTES ? RING_BYTES : EX_OFF + EX_BYTES;
    static_assert(SMEM_BYTES <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(16))) char smem[SMEM_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    EMU_TRACE_MARK(a.trace, 0);
    const uint32_t pfd = prefetch_lines(a.pf_ptr, a.pf_bytes, blockIdx.x * T::THREADS + tid, gridDim.x * T::THREADS);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = wave / (T::WN * T::WM), wtile = wave % (T::WN * T::WM);
    const int wn = wtile / T::WM, wm = wtile % T::WM;
    const int l31 = lane & 31, hi = lane >> 5;

    // Workgroups [0, full_tiles) own whole tiles (XCD-aware bijective remap so the 8 L2s each see a compact set of
    // tiles); the rest are K-slices of the tail tiles (wave-quantisation fix: a 2.13-round problem would otherwise pay
    // for 3 rounds).  full_tiles == number of tiles and ksplit == 1 for an unsplit launch.
    // The slice workgroups take the same remap over a slice-major order (slice ks of every tail tile, then ks + 1), which
    // keeps the row tiles that read one slice of a weight column on one XCD.
    const int b = blockIdx.x;
    const int tiles_m = (a.M + T::BMv - 1) / T::BMv;
    auto xcd_order = [](int i, int n) {                              // i-th block of n -> its place in the logical order
        const int xcd = i & 7, q8 = n >> 3, r8 = n & 7;
        return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (i >> 3);
    };
    int wg, ks = 0, nsl = 1;
    if (b < a.full_tiles) {
        wg = xcd_order(b, a.full_tiles);
        if (a.sup_m) {
            // 2-D blocks: logical index -> (block, position inside it); an XCD's run of the logical order is exactly one block, so
            // its L2 holds sup_m tile rows of A and sup_n tile rows of W instead of (nearly) all of A
            const int per = a.sup_m * a.sup_n, blk = wg / per, r = wg - blk * per;
            const int bm = tiles_m / a.sup_m, bi = blk % bm, bj = blk / bm;
            wg = (bj * a.sup_n + r / a.sup_m) * tiles_m + bi * a.sup_m + r % a.sup_m;
        }
    } else {
        nsl = a.ksplit;
        const int rest = tiles_m * ((a.N + T::BNv - 1) / T::BNv) - a.full_tiles;
        const int l = xcd_order(b - a.full_tiles, rest * nsl);
        ks = l / rest;
        wg = a.full_tiles + (l - ks * rest);
        if (a.slice_rr) {
            const int j = b - a.full_tiles;
            wg = a.full_tiles + j / nsl;
            ks = j - (wg - a.full_tiles) * nsl;
        }
    }
    const int n0 = (wg / tiles_m) * T::BNv, m0 = (wg % tiles_m) * T::BMv;

    // per-lane source of every LDS-DMA instruction: LDS row r = (i*NW + wave)*8 + lane/8, slot p = lane%8 receives global
    // chunk c = p ^ ((r >> 1) & 7); (r >> 1) & 7 = 4*(wave & 1) + lane/16 for every i (NW is even), so one chunk per lane
    const int ck = ((lane & 7) ^ (((wave & 1) << 2) | (lane >> 4))) * 8;
    const bf16_t* gW[T::NLW];
    const bf16_t* gA[T::NLA];
    int pb[T::NLA], py[T::NLA], px[T::NLA];
#pragma unroll
    for (int i = 0; i < T::NLW; ++i) {
        const int r = (i * NW + wave) * 8 + (lane >> 3);
        int gn = n0 + r; gn = gn < a.N ? gn : a.N - 1;
        gW[i] = a.W + (((size_t)gn * a.ldw) >> KSH) + ck;
    }
#pragma unroll
    for (int i = 0; i < T::NLA; ++i) {
        const int r = (i * NW + wave) * 8 + (lane >> 3);
        int gm = m0 + r; gm = gm < a.M ? gm : a.M - 1;
        gA[i] = a.A + (((size_t)gm * a.lda) >> KSH) + ck;
        if constexpr (CONV) {
            const int hw = a.conv.Hout * a.conv.Wout;
            pb[i] = gm / hw;
            const int rr = gm - pb[i] * hw;
            py[i] = rr / a.conv.Wout;
            px[i] = rr - py[i] * a.conv.Wout;
            if (a.conv.cpt_magic) {
                // fast gather (ConvGeom): gA = the centre source pixel's channel 0 (+ this lane's chunk), py = the 9-bit tap mask
                // (nearest x2 upsampling: the centre is (py / 2, px / 2) and the parities of py, px ride in bits 9, 10)
                const bool up = a.conv.mode == CONV_3X3_UP2;
                const int cy = up ? py[i] >> 1 : (a.conv.mode == CONV_3X3_S2 ? 2 * py[i] : py[i]);
                const int cx = up ? px[i] >> 1 : (a.conv.mode == CONV_3X3_S2 ? 2 * px[i] : px[i]);
                int mask = 0;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    int yi, xi;
                    mask |= (int)conv_tap(a.conv, py[i], px[i], t / 3, t % 3, yi, xi) << t;
                }
                gA[i] = a.A + ((size_t)(pb[i] * a.conv.Hin + cy) * a.conv.Win + cx) * a.conv.Cin + ck;
                py[i] = mask | ((py[i] & 1) << 9) | ((px[i] & 1) << 10);
            }
        }
    }
    // split-K: this workgroup owns K-tiles [kt0, kt0 + nk) of slice ks
    const int Kb = a.K >> KSH;                          // K in 2-byte units: k tiles of 128 bytes per row
    const int nk_all = (Kb + BK - 1) / BK;
    const int kt0 = (int)((long)ks * nk_all / nsl);
    const int nk = (int)((long)(ks + 1) * nk_all / nsl) - kt0;
    auto issue = [&](int kt, int stage) {
        kt = kt < nk ? kt : nk - 1;                    // past-the-end tiles re-load the last one (keeps vmcnt counts uniform)
        const int k0 = (kt0 + kt) * BK;
        char* base = smem + stage * T::ST_BYTES + wave * 1024;
        const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero16);
        if constexpr (CONV) {
#pragma unroll
            for (int i = 0; i < T::NLW; ++i) glds16(gW[i] + k0, base + i * NW * 1024);
            // a 64-wide k tile lies inside one filter tap because Cin % 64 == 0
            if (a.conv.cpt_magic) {
                const int ktile = kt0 + kt;
                const int tap = (ktile * a.conv.cpt_magic) >> 16, ci0 = (ktile - tap * a.conv.cpt) << 6;
                const int ky = (tap * 11) >> 5, kx = tap - ky * 3;
                if (a.conv.mode != CONV_3X3_UP2) {
                    const long delta = (long)((ky - 1) * a.conv.Win + (kx - 1)) * a.conv.Cin + ci0;  // wave-uniform
#pragma unroll
                    for (int i = 0; i < T::NLA; ++i)
                        glds16(((py[i] >> tap) & 1) ? gA[i] + delta : zero, base + T::W_BYTES + i * NW * 1024);
                } else {                               // upsampled: the source step of a tap depends on the pixel's parity
#pragma unroll
                    for (int i = 0; i < T::NLA; ++i) {
                        const int dy = (((py[i] >> 9) & 1) + ky - 1) >> 1, dx = (((py[i] >> 10) & 1) + kx - 1) >> 1;
                        const long delta = (long)(dy * a.conv.Win + dx) * a.conv.Cin + ci0;
                        glds16(((py[i] >> tap) & 1) ? gA[i] + delta : zero, base + T::W_BYTES + i * NW * 1024);
                    }
                }
            } else {
                const int tap = k0 / a.conv.Cin, ci0 = k0 - tap * a.conv.Cin;
                const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
                for (int i = 0; i < T::NLA; ++i) {
                    int yi, xi;
                    const bool ok = conv_tap(a.conv, py[i], px[i], ky, kx, yi, xi);
                    const size_t off = (((size_t)pb[i] * a.conv.Hin + yi) * a.conv.Win + xi) * a.conv.Cin + ci0 + ck;
                    glds16(ok ? a.A + off : zero, base + T::W_BYTES + i * NW * 1024);
                }
            }
        } else if (k0 + BK <= Kb) {
#pragma unroll
            for (int i = 0; i < T::NLW; ++i) glds16(gW[i] + k0, base + i * NW * 1024);
#pragma unroll
            for (int i = 0; i < T::NLA; ++i) glds16(gA[i] + k0, base + T::W_BYTES + i * NW * 1024);
        } else {                                       // ragged last k tile (K % 64 != 0): chunks beyond K read zeros
            const bool in = (k0 + ck) < Kb;
#pragma unroll
            for (int i = 0; i < T::NLW; ++i) glds16(in ? gW[i] + k0 : zero, base + i * NW * 1024);
#pragma unroll
            for (int i = 0; i < T::NLA; ++i) glds16(in ? gA[i] + k0 : zero, base + T::W_BYTES + i * NW * 1024);
        }
    };

    f32x16_t acc[T::NF][T::MF];
#pragma unroll
    for (int i = 0; i < T::NF; ++i)
#pragma unroll
        for (int j = 0; j < T::MF; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fused LayerNorm, consumer side: the partial sums of this lane's rows (<= 10 pairs per row) are requested here, ahead of
    // everything else, and only summed in the epilogue: hipcc waits at the first USE of a loaded value, the main loop's waits
    // are our counted asm ones (older loads complete first), so the L2 / fabric round trip (~4 us when it is paid at the end
    // of the kernel: the producer's lines sit in another XCD's L2 or in memory) costs nothing
    LnRaw<T::MF> lnraw;
    const bool ln_on = (FX & FX_LN) != 0 && nsl == 1;
    if constexpr ((FX & FX_LN) != 0) {
        if (ln_on) {
            int mr[T::MF];
#pragma unroll
            for (int j = 0; j < T::MF; ++j) {
                const int m = m0 + (wm * T::MF + j) * 32 + l31;
                mr[j] = m < a.M ? m : a.M - 1;
            }
            ln_rows_load<T::MF>(a, mr, lnraw);
        }
    }
#pragma unroll
    for (int t = 0; t < T::NSTG - 1; ++t) issue(t, t);
    for (int kt = 0; kt < nk; ++kt) {
        wait_vmcnt<(T::NSTG - 2) * T::LPT>();          // this wave's share of tile kt has landed
        __builtin_amdgcn_s_barrier();                  // ... and everyone's; everyone is also done reading tile kt-1
#ifdef EMU_TRACE
        if (kt == 0) EMU_TRACE_MARK(a.trace, 1);
#endif
        issue(kt + T::NSTG - 1, (kt + T::NSTG - 1) % T::NSTG);
        const char* sW = smem + (kt % T::NSTG) * T::ST_BYTES;
        const char* sA = sW + T::W_BYTES;
        if constexpr (F8) {
            typedef int v4i_t __attribute__((ext_vector_type(4)));
            typedef int v8i_t __attribute__((ext_vector_type(8)));
            auto op = [](const bf16x8_t& lo, const bf16x8_t& up) {
                return __builtin_shufflevector(__builtin_bit_cast(v4i_t, lo), __builtin_bit_cast(v4i_t, up), 0, 1, 2, 3, 4, 5, 6, 7);
            };
            bf16x8_t wf8[2][T::NF][2], af8[2][T::MF][2];
            auto frags8 = [&](int st, int buf) {           // k-step st = elements 64 st .. + 63: 32 bytes per lane
                const int ch = st * 4 + hi * 2;
#pragma unroll
                for (int p = 0; p < 2; ++p) {
#pragma unroll
                    for (int i = 0; i < T::NF; ++i)
                        wf8[buf][i][p] = *reinterpret_cast<const bf16x8_t*>(sW + T::off((wn * T::NF + i) * 32 + l31, ch + p));
#pragma unroll
                    for (int j = 0; j < T::MF; ++j)
                        af8[buf][j][p] = *reinterpret_cast<const bf16x8_t*>(sA + T::off((wm * T::MF + j) * 32 + l31, ch + p));
                }
            };
            constexpr int KSTEPS8 = 2 / T::KG;
            const int st0 = kg * KSTEPS8;
            frags8(st0, 0);
#pragma unroll
            for (int st = 0; st < KSTEPS8; ++st) {
                if (st < KSTEPS8 - 1) frags8(st0 + st + 1, (st + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < T::MF; ++j) {
                    const v8i_t bq = op(af8[st & 1][j][0], af8[st & 1][j][1]);
#pragma unroll
                    for (int i = 0; i < T::NF; ++i)
                        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(op(wf8[st & 1][i][0], wf8[st & 1][i][1]), bq, acc[i][j],
                                                                                   0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            continue;
        }
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
        constexpr int KSTEPS = (BK / 16) / T::KG;  // k-steps of this wave's k-group
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
    prefetch_release(pfd);
    EMU_TRACE_MARK(a.trace, 2);
#ifdef EMU_TRACE
    struct TraceEnd { unsigned long long* t; __device__ ~TraceEnd() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); EMU_TRACE_MARK(t, 3); } } trace_end{a.trace};
#endif

    // Staged epilogue (gemm_tile.h::EpiStage): whole-K tiles that lie inside N and carry no V^T / cross-attention output
    constexpr bool GLU = EPI == EPI_SWIGLU || EPI == EPI_GEGLU;
    using Stage = EpiStage<T::BMv, GLU ? T::BNv / 2 : T::BNv, T::THREADS>;
    bool staged = a.stage && nsl == 1 && n0 + T::BNv <= a.N && (FX & FX_CROSS) == 0;
    bool vt_tile = false;                              // a whole tile of V columns: transposed staging
    if constexpr ((FX & FX_VT) != 0) {
        vt_tile = staged && a.stage_vt && n0 >= a.vt_col0;
        staged = staged && (n0 + T::BNv <= a.vt_col0 || vt_tile);
    }
    using StageT = EpiStageT<T::BMv, T::BNv, T::THREADS>;
    char* stage = smem + STAGE_OFF;
    if (staged || T::KG > 1) __syncthreads();          // every wave is done with the ring, every tail DMA has landed
    if constexpr (EPI == EPI_RESID && STAGE_BEHIND) {
        if (staged) Stage::load(stage, a.res, a.ldres, m0, n0, a.M);   // in flight under the k-group sums below
    }

    if constexpr (T::KG > 1) {
        // sum the two k-groups' partial accumulators through LDS (the ring is dead now): group 1 writes, group 0 adds
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
        if (kg == 1 && !staged) {
            if constexpr ((FX & FX_STATS) != 0) {
                if (nsl == 1) __syncthreads();         // the statistics exchange of the epilogue below (workgroup-wide barrier)
            }
            return;
        }
        if (kg == 0) {
#pragma unroll
            for (int i = 0; i < T::NF; ++i)
#pragma unroll
                for (int j = 0; j < T::MF; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] += red[((i * T::MF + j) * 16 + r) * 64 + lane];
        }
    }
    if constexpr (!STAGE_BEHIND) {
        if (staged) {
            __syncthreads();                           // the k-group sums have been read: their place becomes the tile's
            if constexpr (EPI == EPI_RESID) Stage::load(stage, a.res, a.ldres, m0, n0, a.M);
        }
    }

    if constexpr (F8) {
        if (nsl == 1 && kg == 0) {                     // K-sliced: splitk_reduce_kernel scales the summed slices
#pragma unroll
            for (int j = 0; j < T::MF; ++j) {
                const int m = m0 + (wm * T::MF + j) * 32 + l31;
                const float sa = a.a_scale[m < a.M ? m : a.M - 1];
#pragma unroll
                for (int i = 0; i < T::NF; ++i)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int n = n0 + (wn * T::NF + i) * 32 + 8 * g + 4 * hi + e;
                            acc[i][j][4 * g + e] *= sa * a.w_scale[n < a.N ? n : a.N - 1];
                        }
            }
        }
    }
    static_assert(T::NF == 2, "a wave's columns of one row are half a 128-column row-statistics slot");
    RowFx rowfx[T::MF];
    if constexpr ((FX & FX_LN) != 0) {
        if (ln_on) ln_rows_finish<T::MF>(a, lnraw, rowfx);
    }
    // ---- cross-attention epilogue (FX_CROSS, 128 x 64 tile): wave (wn, wm) holds q of ONE head (64 d = acc[0] | acc[1]) for 32
    // queries (lane l31; d = 32 i + (r & 3) + 8 (r >> 2) + 4 hi in register r of acc[i]).  Both MFMAs of the attention take
    // their B operand straight from accumulator registers: a k-step's 8 slots per lane hold whatever contraction indices the
    // lane owns (registers 8 s .. 8 s + 7), and the A operand (K rows / V^T rows from the prompt cache) is gathered with the same
    // permutation -- two 8-byte pieces per fragment -- so no lane exchange and no LDS pass is needed.
    if constexpr ((FX & FX_CROSS) != 0 && T::MF == 1 && T::NF == 2) {
        const int m = m0 + wm * 32 + l31;
        const int ncol = n0 + wn * 64;                 // first column of this wave = head * 64
        if (ncol < a.N) {                              // wave-uniform
            // q, with the LayerNorm correction where it is folded in, rounded to bf16 as the reference's to_q output is
            const RowFx qfx = rowfx[0];
            bf16x8_t qf[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float v[16];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int nb = ncol + i * 32 + 8 * g + 4 * hi;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[4 * g + e] = acc[i][0][4 * g + e];
                    if constexpr ((FX & FX_LN) != 0) {
                        const f32x4_t c = *reinterpret_cast<const f32x4_t*>(a.ln_c + nb);
                        const f32x4_t d = *reinterpret_cast<const f32x4_t*>(a.ln_d + nb);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[4 * g + e] = fmaf(qfx.rstd, v[4 * g + e] - qfx.mean * c[e], d[e]);
                    }
                }
#pragma unroll
                for (int sb = 0; sb < 2; ++sb) {
                    u32x4 w;
                    w.x = packbf(v[8 * sb + 0], v[8 * sb + 1]); w.y = packbf(v[8 * sb + 2], v[8 * sb + 3]);
                    w.z = packbf(v[8 * sb + 4], v[8 * sb + 5]); w.w = packbf(v[8 * sb + 6], v[8 * sb + 7]);
                    qf[i][sb] = __builtin_bit_cast(bf16x8_t, w);
                }
            }
            const int bidx = m0 / a.cross_rows, head = ncol >> 6;
            // slot j of k-step (i, sb) on lane-half hi = contraction index 32 i + 16 sb + 4 hi + (j & 3) + 8 (j >> 2)
            auto gather = [&](const bf16_t* row, int i, int sb) {
                const bf16_t* p = row + 32 * i + 16 * sb + 4 * hi;
                const u32x2 lo = *reinterpret_cast<const u32x2*>(p), up = *reinterpret_cast<const u32x2*>(p + 8);
                return __builtin_bit_cast(bf16x8_t, u32x4{lo.x, lo.y, up.x, up.y});
            };
            // S^T[key, query] = sum_d K[key, d] q[query, d]: key = 32 kb + l31 as the A row
            f32x16_t sacc[2];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[kb][r] = 0.f;
                int key = 32 * kb + l31; key = key < a.cross_n ? key : a.cross_n - 1;
                const bf16_t* krow = a.cross_k + ((size_t)bidx * a.cross_n + key) * a.cross_ldk + head * 64;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int sb = 0; sb < 2; ++sb)
                        sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gather(krow, i, sb), qf[i][sb], sacc[kb], 0, 0, 0);
            }
            // softmax over the keys of this lane's query: 32 scores here, 32 on lane l ^ 32 (exp2 domain, like the flash kernel)
            const float sc = a.cross_scale * 1.4426950408889634f;
            float mloc = -INFINITY;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = 32 * kb + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    const float v = key < a.cross_n ? sacc[kb][r] : -INFINITY;
                    sacc[kb][r] = v;
                    mloc = fmaxf(mloc, v);
                }
            mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64)) * sc;
            const float nm = -mloc;
            float psum = 0.f;
            bf16x8_t pf[2][2];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int sb = 0; sb < 2; ++sb) {
                    float pv[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) { pv[j] = __builtin_amdgcn_exp2f(fmaf(sacc[kb][8 * sb + j], sc, nm)); psum += pv[j]; }
                    u32x4 w;
                    w.x = packbf(pv[0], pv[1]); w.y = packbf(pv[2], pv[3]); w.z = packbf(pv[4], pv[5]); w.w = packbf(pv[6], pv[7]);
                    pf[kb][sb] = __builtin_bit_cast(bf16x8_t, w);
                }
            const float ltot = psum + __shfl_xor(psum, 32, 64);
            const float inv = ltot > 0.f ? 1.f / ltot : 0.f;
            // O^T[d, query] = sum_key V^T[d, key] P[query, key]: d = 32 db + l31 as the A row
            const bf16_t* vbase = a.cross_vt + ((size_t)bidx * (a.N >> 6) + head) * 64 * a.cross_npad;
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                f32x16_t o;
#pragma unroll
                for (int r = 0; r < 16; ++r) o[r] = 0.f;
                const bf16_t* vrow = vbase + (size_t)(32 * db + l31) * a.cross_npad;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int sb = 0; sb < 2; ++sb)
                        o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gather(vrow, kb, sb), pf[kb][sb], o, 0, 0, 0);
                if (m < a.M) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        u32x2 ov;
                        ov.x = packbf(o[4 * g] * inv, o[4 * g + 1] * inv);
                        ov.y = packbf(o[4 * g + 2] * inv, o[4 * g + 3] * inv);
                        *reinterpret_cast<u32x2*>(a.C + (size_t)m * a.ldc + ncol + db * 32 + 8 * g + 4 * hi) = ov;
                    }
                }
            }
        }
        return;
    }
    // Memory operands of the epilogue are fetched ahead of the stores (gemm_tile.h::QuadIn), one 32-column fragment i at a
    // time so the 128 x 64 tile stays inside 128 VGPRs (two workgroups per CU): the column-only operands of its 4 quads once,
    // the per-row ones for all 4 quads of a row before that row's first store.
    RowFx rows[T::MF];
#pragma unroll
    for (int j = 0; j < T::MF; ++j) rows[j] = rowfx[j];
    if (staged) {
        if constexpr (EPI == EPI_RESID) {
            wait_vmcnt<0>();                           // this wave's share of the residual tile ...
            __syncthreads();                           // ... and everyone's
        }
        if (kg == 0) {
#pragma unroll
            for (int i = 0; i < T::NF; ++i) {
                QuadIn qin[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) quad_load_cols<EPI, FX>(a, n0 + (wn * T::NF + i) * 32 + 8 * g + 4 * hi, qin[g]);
#pragma unroll
                for (int j = 0; j < T::MF; ++j) {
                    const int row = (wm * T::MF + j) * 32 + l31, m = m0 + row;
                    // (rows beyond M of a transposed V^T tile are staged too: copies of row M - 1, finite, landing in the pad keys of
                    // the last 8-key group that StageT::store writes -- never whatever the dead k-tile ring held)
                    if (m >= a.M && !((FX & FX_VT) != 0 && vt_tile)) continue;
                    if (a.bias2) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int nb = n0 + (wn * T::NF + i) * 32 + 8 * g + 4 * hi;
                            qin[g].bias2 = *reinterpret_cast<const u32x2*>(a.bias2 + (size_t)(m / a.rows_per_batch) * a.ld_bias2 + nb);
                        }
                    }
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int col = (wn * T::NF + i) * 32 + 8 * g + 4 * hi;       // column inside the tile
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e];
                        if constexpr (GLU) {
                            const u32x2 ov = quad_value<EPI, FX>(a, v, rows[j], qin[g]);
                            *reinterpret_cast<uint32_t*>(stage + Stage::off(row, col >> 1)) = ov.x;
                        } else if ((FX & FX_VT) != 0 && vt_tile) {
                            const u32x2 ov = quad_value<EPI, FX>(a, v, rows[j], qin[g]);
                            *reinterpret_cast<bf16_t*>(stage + StageT::off(col, row)) = (bf16_t)(ov.x & 0xffffu);
                            *reinterpret_cast<bf16_t*>(stage + StageT::off(col + 1, row)) = (bf16_t)(ov.x >> 16);
                            *reinterpret_cast<bf16_t*>(stage + StageT::off(col + 2, row)) = (bf16_t)(ov.y & 0xffffu);
                            *reinterpret_cast<bf16_t*>(stage + StageT::off(col + 3, row)) = (bf16_t)(ov.y >> 16);
                        } else {
                            u32x2* cell = reinterpret_cast<u32x2*>(stage + Stage::off(row, col));
                            if constexpr (EPI == EPI_RESID) qin[g].res = *cell;
                            *cell = quad_value<EPI, FX>(a, v, rows[j], qin[g]);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
        if ((FX & FX_VT) != 0 && vt_tile) StageT::store(stage, a, m0, n0);
        else Stage::store(stage, a.C, a.ldc, m0, GLU ? n0 >> 1 : n0, a.M);
    } else {
#pragma unroll
    for (int i = 0; i < T::NF; ++i) {
        QuadIn qin[4];
        if (nsl == 1) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nb = n0 + (wn * T::NF + i) * 32 + 8 * g + 4 * hi;
                if (nb < a.N) quad_load_cols<EPI, FX>(a, nb, qin[g]);
            }
        }
#pragma unroll
        for (int j = 0; j < T::MF; ++j) {
            const int m = m0 + (wm * T::MF + j) * 32 + l31;
            if (m >= a.M) continue;
            if (nsl == 1) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int nb = n0 + (wn * T::NF + i) * 32 + 8 * g + 4 * hi;
                    if (nb < a.N) quad_load_row<EPI>(a, m, nb, qin[g]);
                }
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nb = n0 + (wn * T::NF + i) * 32 + 8 * g + 4 * hi;
                if (nb >= a.N) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e];
                if (nsl > 1) {                         // raw fp32 slice tile; splitk_reduce_kernel applies the epilogue
                    float* dst = a.slab_rows ? a.partial + ((size_t)ks * a.M + m) * a.N + nb     // row-major slices: rows_reduce_norm_kernel
                                             : a.partial + ((size_t)(wg - a.full_tiles) * nsl + ks) * (T::BMv * T::BNv) +
                                                   (size_t)(m - m0) * T::BNv + (nb - n0);
                    *reinterpret_cast<f32x4_t*>(dst) = f32x4_t{v[0], v[1], v[2], v[3]};
                } else {
                    store_quad<EPI, FX>(a, m, nb, v, rows[j], qin[g]);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    }
    // fused LayerNorm, producer side: a 128-column statistics slot of a row = the 64 columns of wave (wn, wm) + those of wave
    // (wn + 1, wm), each split over lanes l and l + 32.  Odd wn hands its sums to its even neighbour through LDS.
    if constexpr ((FX & FX_STATS) != 0) {
        static_assert(T::WN % 2 == 0, "waves pair up along n");
        if (nsl == 1) {                                // workgroup-uniform
            f32x2_t* ex = reinterpret_cast<f32x2_t*>(smem + EX_OFF);
            if ((wn & 1) && kg == 0) {
#pragma unroll
                for (int j = 0; j < T::MF; ++j) ex[(wtile * T::MF + j) * 64 + lane] = f32x2_t{rows[j].rs, rows[j].rq};
            }
            __syncthreads();
            const int nslot = n0 + wn * 64;
            if (!(wn & 1) && kg == 0 && nslot < a.N) {
#pragma unroll
                for (int j = 0; j < T::MF; ++j) {
                    const int m = m0 + (wm * T::MF + j) * 32 + l31;
                    const f32x2_t o = ex[((wtile + T::WM) * T::MF + j) * 64 + lane];
                    float sm = rows[j].rs + o[0], q = rows[j].rq + o[1];
                    sm += __shfl_xor(sm, 32, 64);
                    q += __shfl_xor(q, 32, 64);
                    if (hi == 0 && m < a.M)
                        *reinterpret_cast<f32x2_t*>(a.row_stats_out + ((size_t)(nslot / LN_SLOT_COLS) * a.M + m) * 2) = f32x2_t{sm, q};
                }
            }
        }
    }
}


using CfgB = TileCfg<2, 2, 2, 2, 2>;     // 128 x 128, 4 waves, 2 stages (64 KiB, 2 workgroups per CU)
using CfgC = TileCfg<4, 2, 2, 2, 3>;     // 256(n) x 128(m), 8 waves, 3 stages (144 KiB)
using CfgK = TileCfg<2, 2, 2, 1, 3, 2>;  // 128(n) x 64(m), 2 k-groups x 4 waves, 3 stages (72 KiB, 2 workgroups per CU)
// Round 4, measured with per-workgroup timelines and not kept (profiles/r04_gemm_trace_configs_DEFG.log): 128 x 128 with two
// k-groups and a FOUR-stage ring (10.9 us of loop on the UNet's 2048 x 1280 x 1280 against 11.6 for CfgB: the loop does not wait
// for the fabric); 256 x 128 with two k-groups = 1024 threads (slower than CfgC everywhere, and its fused-LayerNorm forms spill at
// 128 VGPRs); an extra L2-WARMING wave per workgroup for the 128 x 64 and 128 x 128 tiles (walks the panel ahead of the LDS-DMA
// through its own vmcnt): the first k tile lands after 5.6 us instead of 1.5 and the kernel is 25-60 % slower -- the loops of these
// tiles are bound by what one CU can move into LDS per clock (24-32 KiB per k tile against 256-512 MFMA cycles), not by latency.
// Measured and not kept (profiles/r01_gemm_tilecfg_sweep_*.log): deeper rings of the 128-wide tiles, 128(n) x 256(m),
// 64 x 64, 32-wide k tiles (twice the barriers per FLOP), 4 waves of 128 x 64 / 128 x 128 on the big tiles, and a
// 256 x 256 tile on this lock-step pipeline (-40 %: what the big tile needs is the phase schedule of gemm256.hip).

float* g_splitk_scratch = nullptr;
size_t g_splitk_floats = 0;
unsigned long long* g_trace = nullptr;   // emu_gemm_trace_set
long g_trace_sel = -1, g_trace_count = 0; // emu_gemm_trace_select: only the sel-th GEMM launch since then is traced (-1: all)
int g_force_cfg = 0;                     // emu_gemm_force_config: tests / benches pin one tile configuration
int g_tune = 0;                          // emu_gemm_tune: A/B switches of single dispatch decisions (tools/unet_ab.py)

// full_tiles whole-K workgroups followed by (tiles - full_tiles) * ksplit slice workgroups, one launch (+ the reduce)
template <int EPI, bool CONV, class T, bool F8 = false>
void launch_cfg(const GemmArgs& a, hipStream_t s, int full_tiles = -1, int ksplit = 1) {
    const int tiles = ((a.M + T::BMv - 1) / T::BMv) * ((a.N + T::BNv - 1) / T::BNv);
    GemmArgs b = a;
    b.full_tiles = full_tiles < 0 ? tiles : full_tiles;
    b.ksplit = ksplit;
    b.trace = emu_gemm_trace_get();
    b.stage = stage_ok(b) && !(g_tune & 8);
    b.stage_vt = b.stage && stage_vt_ok(b, T::BMv, T::BNv) && !(g_tune & (1 << 14));
    // XCD-aware 2-D tile blocks (unsplit launches whose tile count splits evenly over the 8 XCDs): block b runs on XCD b % 8 and
    // xcd_order hands every XCD a run of tiles / 8 logical indices; a run of the column-major order covers (nearly) all rows of A
    // when tiles_m is large (2048 x 1280 on 128 x 64 tiles: 32 x 10 tiles, 40 per XCD = all 32 row tiles x 2 weight tiles = 5.2 +
    // 0.65 MB per L2 per GEMM, measured 50 MB fetched for 13.7 MB of operands); as an 8 x 5 block it is 1.3 + 1.6 MB
    if (b.full_tiles == tiles && tiles % 8 == 0 && !(g_tune & 16)) {
        const int tm = (a.M + T::BMv - 1) / T::BMv, tn = tiles / tm, per = tiles / 8;
        const int cur_cols = (per + tm - 1) / tm + ((per % tm) ? 1 : 0);
        long best = (long)(per < tm ? per : tm) * T::BMv + (long)(cur_cols < tn ? cur_cols : tn) * T::BNv;
        for (int sm = 1; sm <= tm; ++sm) {
            if (tm % sm || per % sm || tn % (per / sm)) continue;
            const long cost = (long)sm * T::BMv + (long)(per / sm) * T::BNv;
            if (cost < best) { best = cost; b.sup_m = sm; b.sup_n = per / sm; }
        }
    }
    const int tail = tiles - b.full_tiles;
    const int fx = gemm_fx(b);
    if constexpr (F8) {                                 // launch_gemm_fp8: V^T / cross-attention epilogues only (EPI_NONE, unsliced)
        if constexpr (EPI == EPI_NONE) {
            if (fx == FX_VT && tail == 0) {
                hipLaunchKernelGGL((gemm2_kernel<EPI, false, T, FX_VT, true>), dim3(b.full_tiles), dim3(T::THREADS), 0, s, b);
                return;
            }
            if constexpr (T::MF == 1 && T::NF == 2 && T::KG == 2) {
                if (fx == FX_CROSS && tail == 0) {
                    hipLaunchKernelGGL((gemm2_kernel<EPI, false, T, FX_CROSS, true>), dim3(b.full_tiles), dim3(T::THREADS), 0, s, b);
                    return;
                }
            }
        }
        hipLaunchKernelGGL((gemm2_kernel<EPI, false, T, 0, true>), dim3(b.full_tiles + tail * ksplit), dim3(T::THREADS), 0, s, b);
        if (tail > 0)
            hipLaunchKernelGGL((splitk_reduce_kernel<EPI, T::BMv, T::BNv>), dim3(tail, SPLITK_RED_Y), dim3(256), 0, s, b);
        return;
    }
    if (fx) {                                           // launch_gemm has checked gemm_fx_ok(epi, fx)
        if constexpr (!CONV) {
            gemm_fx_dispatch<EPI>(fx, [&](auto m) {
                constexpr int FXM = decltype(m)::value;
                hipLaunchKernelGGL((gemm2_kernel<EPI, CONV, T, FXM>), dim3(b.full_tiles + tail * ksplit), dim3(T::THREADS), 0, s, b);
                if (tail > 0)
                    hipLaunchKernelGGL((splitk_reduce_kernel<EPI, T::BMv, T::BNv, FXM>), dim3(tail, SPLITK_RED_Y), dim3(256), 0, s, b);
            });
        }
        return;
    }
    hipLaunchKernelGGL((gemm2_kernel<EPI, CONV, T>), dim3(b.full_tiles + tail * ksplit), dim3(T::THREADS), 0, s, b);
    if (tail > 0 && b.slab_rows) launch_rows_reduce_norm(b, s);        // launch_v2: every tile sliced, slices row-major
    else if (tail > 0)
        hipLaunchKernelGGL((splitk_reduce_kernel<EPI, T::BMv, T::BNv>), dim3(tail, SPLITK_RED_Y), dim3(256), 0, s, b);
}

inline int tiles_of(const GemmArgs& a, int bn, int bm) { return ((a.M + bm - 1) / bm) * ((a.N + bn - 1) / bn); }
// k tiles (128 bytes per operand row) of a problem: 64 bf16 or 128 fp8 elements each
inline int ktiles_of(const GemmArgs& a) { return a.a_scale ? a.K / 128 : a.K / BK; }

// K-slices for a problem of `tiles` tiles (fewer than the 256 CUs) so that every CU gets about one workgroup; 0 = do
// not split.  Needs N % 4 == 0 and a non-GLU epilogue (slices are raw fp32 quads), >= min_k k tiles per slice, and a
// scratch that holds tiles * ksplit fp32 tiles.
template <int EPI>
int pick_ksplit(const GemmArgs& a, int tiles, int tile_elems, int min_k) {
    if (EPI == EPI_SWIGLU || EPI == EPI_GEGLU || (a.N & 3) || !a.partial) return 0;
    int ksplit = 256 / tiles;
    if (ksplit > 8) ksplit = 8;
    const int nk = ktiles_of(a);
    while (ksplit > 1 && nk / ksplit < min_k) --ksplit;
    if (ksplit < 2 || (size_t)tiles * ksplit * tile_elems > a.partial_floats) return 0;
    return ksplit;
}

// How the 256x256 ping-pong tile (gemm256.hip) is spread over the 256 CUs: `full` tiles by one workgroup each, every
// later tile cut into `ks` K-slices (fp32 slabs + a reduce launch).  A cost model picks ks for the tiles beyond the last
// whole round (all of them when there are fewer tiles than CUs): rounds of workgroups x the time of one slice, plus what
// the slabs cost (288 KiB written and read per slice at ~4 TB/s) and the second launch.  Checked against tools/gemm_ab.py
// (profiles/r02_gemm_ab_v10_tail_slices.log): S=770 o / down 3 slices (+75 % / +100 % over one workgroup per tile),
// S=1544 down 3 slices (+13 %), S=770 gate/up none (420 tiles: 2 rounds either way), S=1544 gate/up 2-3 slices (+3 %).
struct PpPlan { bool use; int full_tiles, ksplit; double us; };
inline PpPlan plan_pp(const GemmArgs& a) {
    const int tp = gemm256_tiles(a), nk = ktiles_of(a), CU = 256;
    // us: one CU at ~55 % of its share of the MFMA peak (fp8 operands: twice the rate at a somewhat lower fraction)
    const double t_tile = 256.0 * 256.0 * a.K * 2.0 / 5.4e6 * (a.a_scale ? 0.6 : 1.0);
    const int rem = tp % CU, full = tp - rem;
    PpPlan best{true, tp, 1, (double)((tp + CU - 1) / CU) * t_tile};
    if (rem == 0 || (a.N & 3) || !a.partial) return best;
    for (int ks = 2; ks <= 8; ++ks) {
        if (nk / ks < 8) break;
        if ((size_t)rem * ks * EMU_GEMM256_SLICE_FLOATS > a.partial_floats) break;
        const double us = (full / CU + (double)((rem * ks + CU - 1) / CU) / ks) * t_tile + rem * ks * 0.144 + 8.0;
        if (us < best.us * 0.95) best = PpPlan{true, full, ks, us};
    }
    return best;
}

// Does the 256x256 ping-pong tile take this problem?  From tools/gemm_ab.py on MI355X (profiles/r02_gemm_ab_*): it wins
// wherever a workgroup gets enough k tiles to amortise its pipeline fill and 64K-element epilogue -- LLaMA prefill
// (+30-60 % over the 128x128 / 256x128 tiles), every S >= 1024 shape, ViT fc1 / fc2, the UNet's 64^2 / 128^2 levels and
// convs -- and loses on short-K few-tile GEMMs (ViT qkv / proj, the UNet's K = 1280 projections), on a badly quantised
// round of short-K tiles (320 tiles, K = 1280) and on thin prompts (M < 192).
inline PpPlan pick_pp(const GemmArgs& a) {
    PpPlan no{false, 0, 1, 0.0};
    if (!gemm256_ok(a) || a.M < 192) return no;
    const int tp = gemm256_tiles(a), nk = ktiles_of(a), CU = 256;
    const PpPlan p = plan_pp(a);
    if (tp >= 180) {
        if (nk < 8) return no;
        const int rounds = (tp + CU - 1) / CU;
        return (p.ksplit > 1 || tp * 10 >= rounds * CU * 7) ? p : no;
    }
    // under a quarter round (UNet ff-out / 32^2 convs: 40 tiles, ViT fc2: 28) the K-sliced 256x128 tile has twice the slices to
    // spread and smaller slabs: same-run A/B of the denoise step 28.3 -> 27.8 ms (the micro-benchmark had them level)
    if (tp < 64) return no;
    if (p.ksplit > 1 && nk / p.ksplit >= 16 && tp * p.ksplit >= 150) return p;
    return (tp >= 150 && nk >= 32) ? PpPlan{true, tp, 1, 0.0} : no;
}

// Whole rounds on the 256x256 tile + the remaining weight rows on the lock-step tiles.  A problem of r * 256 + f tiles with a
// small f (the UNet's GEGLU: 8 x 40 = 320 tiles; S=1544 gate/up: 6 x 140 = 840) pays a whole extra round for the f tiles,
// and K-slicing them costs fp32 slabs.  Instead the first n1 tile columns (tiles_m * n1 ~ r * 256: every CU exactly r
// tiles) go to the ping-pong kernel and columns [n1 * 256, N) are a second, independent GEMM on the same A (pointer offsets
// only: W rows, bias, residual / output columns -- half of them for the interleaved GLU pairs).  n1 = 0: not applicable.
// The heuristic takes it for f <= 3/8 of a round when the cost model would not K-slice the tail (short K): measured
// (profiles/r02_gemm_ab_v12_hybrid.log) UNet GEGLU 608 (256x128) / 701 (256x256, two rounds) -> 747 TFLOP/s, denoise step
// 29.1 -> 28.3 ms (same-run A/B); with long K the sliced tail stays ahead (S=1544 gate/up 1124 vs 1066) and at f = 1/2 the plain rounds do.
inline int plan_hybrid(const GemmArgs& a, bool forced) {
    if (!gemm256_ok(a) || a.M < 192 || a.K / BK < 8 || a.conv.mode != CONV_NONE || a.vt_out) return 0;
    const int CU = 256, tn = (a.N + 255) / 256, tp = gemm256_tiles(a), tm = tp / tn;
    const int r = tp / CU, rem = tp % CU;
    if (r < 1 || rem == 0 || rem > CU / 2) return 0;
    if (!forced && (rem * 8 > CU * 3 || plan_pp(a).ksplit > 1)) return 0;
    const int n1 = r * CU / tm;
    if (n1 < 1 || n1 >= tn || tm * n1 * 10 < r * CU * 9) return 0;
    return n1;
}

// The lock-step tile of a problem the 256x256 ping-pong tile does not take ('S' = K-sliced 256 x 128), bf16 and fp8 alike
// (the rules count tiles and k tiles of 128 bytes per row).
template <int EPI, bool CONV>
int pick_lockstep(const GemmArgs& a) {
    const bool k64 = a.a_scale ? (a.K & 127) == 0 : (a.K & 63) == 0;
    const int tc = tiles_of(a, 256, 128);
    // (>= 24 k tiles per slice: with fewer the reduce launch costs more than the slices save -- the ViT's proj at 16 per
    // slice ran 284 TFLOP/s sliced, 381 on the 128 x 64 tile; profiles/r03_gemm_ilv_ab.log)
    if (k64 && tc < 256 && pick_ksplit<EPI>(a, tc, 256 * 128, 24)) return 'S';
    // 128x128 tiles are L1/TA-bandwidth-bound (64 FLOP/B needs ~64 B/clk/CU), so the largest problems take the
    // 256(n) x 128(m) tile; mid-size GEMMs 128x128 with two workgroups per CU; few-tile / long-K problems (UNet 32x32
    // level, implicit-GEMM convs, skinny ViT fc2) take 128 x 64 tiles with two k-groups of waves (intra-workgroup
    // split-K) and two workgroups per CU.  Thresholds from tools/kbench.py sweeps (profiles/r01_gemm_tilecfg_*).
    // Round 4, with the staged epilogue (its tail was 6.5-9 us on the 128 x 128 tile, now 3): one to one and a half rounds of
    // 128 x 128 tiles (the UNet's 8192 x 640 outputs: 320 tiles; the 2048-column remainder of the GEGLU: 256) beat the 128 x 64
    // and 256 x 128 tiles by 10-15 % also with COLD weights (profiles/r04_gemm_ab_staged_unet_cold_weights.log); below 200
    // tiles (2048 x 1280: 160 tiles, one workgroup per CU and a two-stage ring) the 128 x 128 tile wins by 15 % on weights
    // that sit in the cache and loses in the model, where every launch finds them in HBM (same-box kernel stats: 24.7 vs 22.3
    // us); not for ragged M (the ViT's 1025 rows: a ninth row of tiles for one row)
    if (!CONV && tiles_of(a, 128, 128) >= 200 && tiles_of(a, 128, 128) < 400 &&
             ((a.M + 127) / 128) * 128 <= a.M + a.M / 16 && !(g_tune & 32)) return 'B';
    if (tc >= 1024) return 'C';
    if ((EPI == EPI_GEGLU || EPI == EPI_SWIGLU) && tc >= 512) return 'C';   // in situ (UNet step): 256x128 29.4, 128x128 29.4, 256x256 29.8 ms
    if (!CONV && tc >= 180 && tc < 400) return 'C';           // ~one 256x128 tile per CU: ViT qkv
    if (!CONV && tiles_of(a, 128, 128) >= 400) return 'B';
    return 'K';
}

template <int EPI, bool CONV>
int launch_v2(const GemmArgs& a0, hipStream_t s) {
    GemmArgs a = a0;
    if (!a.partial) { a.partial = g_splitk_scratch; a.partial_floats = g_splitk_floats; }
    a.slice_rr = (g_tune >> 1) & 1;
    if (g_tune & (1 << 16)) { a.pf_ptr = nullptr; a.pf_bytes = 0; }      // A/B: no successor prefetch
    if (a.cross_k) {                                   // the cross-attention epilogue lives on the 128 x 64 tile
        if constexpr (EPI == EPI_NONE && !CONV) { launch_cfg<EPI, CONV, CfgK>(a, s); EMU_CHECK_LAUNCH(); return 0; }
        return -22;
    }
    if (a.rope_cos) {
        // only the 256x256 tile has this epilogue, and only whole-K tiles: -95 tells the caller to run the unfused sequence
        // (qkv GEMM, rope_kv, transpose_v) instead -- thin prompts, K-sliced tail rounds
        if constexpr (EPI == EPI_NONE && !CONV) {
            const PpPlan pp = pick_pp(a);
            if (!gemm256_ok(a) || !pp.use || pp.ksplit > 1) return -95;
            return launch_gemm256(a, s, -1, 1);
        }
        return -22;
    }
    if (a.norm_w) {
        // slice sum + the following RMSNorm in one row-wise launch: only where the 256x256 tile K-slices EVERY tile of the problem
        // (fewer tiles than CUs: S = 770 o_proj / down_proj) and the row-major slices fit the scratch; -95 = run GEMM and rmsnorm apart
        if constexpr ((EPI == EPI_NONE || EPI == EPI_RESID) && !CONV) {
            if ((a.N & 7) || a.N > 16384) return -95;
            const PpPlan pp = pick_pp(a);
            if (gemm256_ok(a) && pp.use) {
                if (pp.ksplit < 2 || pp.full_tiles != 0 || (size_t)pp.ksplit * a.M * a.N > a.partial_floats) return -95;
                a.slab_rows = 1;
                return launch_gemm256(a, s, 0, pp.ksplit);
            }
            // the K-sliced 256 x 128 lock-step tile (ViT fc2): all of its tiles are slices
            if (pick_lockstep<EPI, CONV>(a) != 'S') return -95;
            const int tc = tiles_of(a, 256, 128);
            const int ksplit = pick_ksplit<EPI>(a, tc, 256 * 128, 24);
            if (ksplit < 2 || (size_t)ksplit * a.M * a.N > a.partial_floats) return -95;
            a.slab_rows = 1;
            launch_cfg<EPI, CONV, CfgC>(a, s, 0, ksplit);
            EMU_CHECK_LAUNCH();
            return 0;
        }
        return -22;
    }
    int cfg = g_force_cfg;
    const bool k64 = (a.K & 63) == 0;
    if (cfg == 'S' && !k64) cfg = 0;
    if ((cfg == 'P' || cfg == 'Q') && !gemm256_ok(a)) cfg = 0;
    if (!cfg || cfg == 'H') {
        int n1 = CONV ? 0 : plan_hybrid(a, cfg == 'H');
        // A/B switch 1: a GLU problem the hybrid would split in two launches runs as ONE launch of 128 x 128 tiles instead
        constexpr bool GLU_EPI = EPI == EPI_SWIGLU || EPI == EPI_GEGLU;
        if (GLU_EPI && n1 > 0 && !cfg && (g_tune & 1) && tiles_of(a, 128, 128) >= 1024) { n1 = 0; cfg = 'B'; }
        if (n1 > 0) {
            constexpr bool GLU = EPI == EPI_SWIGLU || EPI == EPI_GEGLU;
            GemmArgs head = a, rest = a;
            head.N = n1 * 256;
            rest.N = a.N - head.N;
            rest.W = a.W + (size_t)head.N * a.ldw;
            if (a.bias) rest.bias = a.bias + head.N;
            if (a.res) rest.res = a.res + head.N;
            if (a.bias2) rest.bias2 = a.bias2 + head.N;
            if (a.ln_c) { rest.ln_c = a.ln_c + head.N; rest.ln_d = a.ln_d + head.N; }
            if (a.row_stats_out) rest.row_stats_out = a.row_stats_out + (size_t)(head.N / LN_SLOT_COLS) * a.M * 2;   // slots of 128 columns
            rest.C = a.C + (GLU ? head.N / 2 : head.N);
            rest.pf_ptr = nullptr; rest.pf_bytes = 0;              // the head launch prefetches for the successor
            int st = launch_gemm256(head, s, -1, 1);
            if (st != 0) return st;
            const int keep = g_force_cfg;
            g_force_cfg = 0;                                       // the remainder takes whatever the heuristic says
            st = launch_v2<EPI, CONV>(rest, s);
            g_force_cfg = keep;
            return st;
        }
        if (cfg == 'H') cfg = 0;
    }
    if (!cfg) {
        const PpPlan pp = pick_pp(a);
        if (pp.use) return pp.ksplit > 1 ? launch_gemm256(a, s, pp.full_tiles, pp.ksplit) : launch_gemm256(a, s, -1, 1);
        // The 256x128 tile moves the fewest bytes per FLOP through L2 of the lock-step tiles (the binding resource of
        // these kernels) but runs one workgroup per CU: a 2048 x 1280 output is only 80 tiles (0.31 round).  Problems
        // with fewer tiles than CUs are cut into K-slices so they fill the CUs once: fp32 slice tiles land in a
        // scratch, a second launch sums them in order and applies the epilogue.  (Slicing only the tail round of a
        // multi-round problem measured flat: a thin last round simply runs faster.)
        cfg = pick_lockstep<EPI, CONV>(a);
    }
    switch (cfg) {
        case 'Q': return launch_gemm256(a, s, -1, 1);  // 256x256 ping-pong, never K-sliced (A/B)
        case 'P': {                                   // 256x256 ping-pong with the planned K-slices, whatever the shape
            const PpPlan pp = plan_pp(a);
            return pp.ksplit > 1 ? launch_gemm256(a, s, pp.full_tiles, pp.ksplit) : launch_gemm256(a, s, -1, 1);
        }
        case 'S': {
            const int tc = tiles_of(a, 256, 128);
            const int ksplit = tc < 256 ? pick_ksplit<EPI>(a, tc, 256 * 128, g_force_cfg ? 8 : 24) : 0;
            if (ksplit) launch_cfg<EPI, CONV, CfgC>(a, s, 0, ksplit);
            else launch_cfg<EPI, CONV, CfgC>(a, s);
            break;
        }
        case 'C': launch_cfg<EPI, CONV, CfgC>(a, s); break;
        case 'K': launch_cfg<EPI, CONV, CfgK>(a, s); break;
        default:  launch_cfg<EPI, CONV, CfgB>(a, s); break;
    }
    EMU_CHECK_LAUNCH();
    return 0;
}

}  // namespace

void emu_gemm_set_splitk_scratch(float* ptr, size_t floats) { g_splitk_scratch = ptr; g_splitk_floats = floats; }
void emu_gemm_force_config_set(int cfg) { g_force_cfg = cfg & 255; }
void emu_gemm_tune_set(int mask) { g_tune = mask; }
void emu_gemm_trace_set(unsigned long long* buf) { g_trace = buf; }
void emu_gemm_trace_select(long n) { g_trace_sel = n; g_trace_count = 0; }
// the buffer of THIS launch (every GEMM launch asks once): all launches, or only the selected one
unsigned long long* emu_gemm_trace_get() {
    if (!g_trace) return nullptr;
    if (g_trace_sel < 0) return g_trace;
    return g_trace_count++ == g_trace_sel ? g_trace : nullptr;
}
int emu_gemm_tune_get() { return g_tune; }

// fp8 x fp8 -> bf16: the 256x256 ping-pong tile where the bf16 rules would take it, else the lock-step tiles' fp8 form
template <int EPI>
static int launch_fp8_v2(const GemmArgs& a, hipStream_t s) {
    if (a.cross_k) {                                   // the cross-attention epilogue lives on the 128 x 64 tile
        if constexpr (EPI == EPI_NONE) { launch_cfg<EPI, false, CfgK, true>(a, s); EMU_CHECK_LAUNCH(); return 0; }
        return -22;
    }
    if (a.vt_out) {                                    // V^T epilogue: the lock-step tiles, unsliced
        if constexpr (EPI == EPI_NONE) {
            int c = g_force_cfg ? g_force_cfg : pick_lockstep<EPI, false>(a);
            if (c == 'C' && !g_force_cfg) c = 'B';
            if (c == 'K') launch_cfg<EPI, false, CfgK, true>(a, s);
            else if (c == 'C') launch_cfg<EPI, false, CfgC, true>(a, s);
            else launch_cfg<EPI, false, CfgB, true>(a, s);
            EMU_CHECK_LAUNCH();
            return 0;
        }
        return -22;
    }
    int cfg = g_force_cfg;
    if (cfg == 'H') cfg = 0;
    if ((cfg == 'P' || cfg == 'Q') && !gemm256_ok(a)) cfg = 0;
    if (!cfg) {
        const PpPlan pp = pick_pp(a);
        if (pp.use) return pp.ksplit > 1 ? launch_gemm256(a, s, pp.full_tiles, pp.ksplit) : launch_gemm256(a, s, -1, 1);
        cfg = pick_lockstep<EPI, false>(a);
        // with half the LDS-DMA bytes per FLOP the 128 x 128 tile (two workgroups per CU) beats the 256 x 128 tile wherever the
        // bf16 rules pick the latter (profiles/r04_fp8_gemm_time_*.log: UNet qkv 17.7 vs 18.6 us, GEGLU 50.8 vs 53.2, ViT qkv 17.5 vs 20.1)
        if (cfg == 'C') cfg = 'B';
    }
    switch (cfg) {
        case 'Q': return launch_gemm256(a, s, -1, 1);
        case 'P': {
            const PpPlan pp = plan_pp(a);
            return pp.ksplit > 1 ? launch_gemm256(a, s, pp.full_tiles, pp.ksplit) : launch_gemm256(a, s, -1, 1);
        }
        case 'S': {
            const int tc = tiles_of(a, 256, 128);
            const int ksplit = tc < 256 ? pick_ksplit<EPI>(a, tc, 256 * 128, g_force_cfg ? 4 : 24) : 0;
            if (ksplit) launch_cfg<EPI, false, CfgC, true>(a, s, 0, ksplit);
            else launch_cfg<EPI, false, CfgC, true>(a, s);
            break;
        }
        case 'C': launch_cfg<EPI, false, CfgC, true>(a, s); break;
        case 'K': launch_cfg<EPI, false, CfgK, true>(a, s); break;
        default:  launch_cfg<EPI, false, CfgB, true>(a, s); break;
    }
    EMU_CHECK_LAUNCH();
    return 0;
}

static int launch_gemm_fp8_impl(const GemmArgs& a0, hipStream_t s) {
    GemmArgs a = a0;
    if (a.M < 1 || a.N < 1 || !a.a_scale || !a.w_scale || (a.K & 127) || (a.lda & 15) || (a.ldw & 15) || a.conv.mode != CONV_NONE) return -22;
    if ((a.epi == EPI_SWIGLU || a.epi == EPI_GEGLU) && ((a.N & 1) || (a.ldc & 1))) return -22;
    const int fx = gemm_fx(a);
    if ((fx != 0 && fx != FX_VT && fx != FX_CROSS) || a.bias2) return -22;
    if (a.vt_out && (a.epi != EPI_NONE || (a.vt_col0 & 63) || ((a.N - a.vt_col0) & 63) || a.vt_col0 < 0 || a.vt_col0 >= a.N || a.vt_s < 1 ||
                     a.M % a.vt_s || a.vt_spad < a.vt_s || (a.ldc & 3))) return -22;
    if (a.cross_k && (!a.cross_vt || a.epi != EPI_NONE || a.bias || (a.N & 63) || (a.ldc & 3) || a.cross_n < 1 || a.cross_n > 64 ||
                      a.cross_npad < 64 || a.cross_rows < 64 || (a.cross_rows & 63) || a.M % a.cross_rows || (a.cross_ldk & 3) ||
                      (a.cross_npad & 3))) return -22;
    if (!a.partial) { a.partial = g_splitk_scratch; a.partial_floats = g_splitk_floats; }
    a.slice_rr = (g_tune >> 1) & 1;
    switch (a.epi) {
        case EPI_NONE:   return launch_fp8_v2<EPI_NONE>(a, s);
        case EPI_RESID:  return launch_fp8_v2<EPI_RESID>(a, s);
        case EPI_SWIGLU: return launch_fp8_v2<EPI_SWIGLU>(a, s);
        case EPI_GELU:   return launch_fp8_v2<EPI_GELU>(a, s);
        case EPI_GEGLU:  return launch_fp8_v2<EPI_GEGLU>(a, s);
        default: return -22;
    }
}
int launch_gemm_fp8(const GemmArgs& a, hipStream_t s) {
    if (!emu_prof_on()) return launch_gemm_fp8_impl(a, s);
    emu_prof_begin(s);
    const int st = launch_gemm_fp8_impl(a, s);
    if (st == 0) emu_prof_end(s, "gemm_fp8", a.M, a.N, a.K, a.epi, 2.0 * a.M * a.N * a.K);
    else emu_prof_drop();
    return st;
}

static int launch_gemm_impl(const GemmArgs& a, hipStream_t s);
int launch_gemm(const GemmArgs& a, hipStream_t s) {
    if (!emu_prof_on()) return launch_gemm_impl(a, s);
    emu_prof_begin(s);
    const int st = launch_gemm_impl(a, s);
    if (st == 0) emu_prof_end(s, a.conv.mode != CONV_NONE ? "conv" : "gemm", a.M, a.N, a.K, a.epi | (gemm_fx(a) << 8), 2.0 * a.M * a.N * a.K);
    else emu_prof_drop();                                  // (a -95 probe of a fused form, a refused shape: nothing was launched)
    return st;
}
static int launch_gemm_impl(const GemmArgs& a, hipStream_t s) {
    if (a.M < 1 || a.N < 1 || (a.K & 7) || (a.ldw & 7)) return -22;
    if ((a.epi == EPI_SWIGLU || a.epi == EPI_GEGLU) && ((a.N & 1) || (a.ldc & 1))) return -22;
    if (a.bias2 && a.rows_per_batch < 1) return -22;
    // fused LayerNorm / V^T epilogues: whole quads only, statistics slots of LN_SLOT_COLS = 128 columns
    if (!gemm_fx_ok(a.epi, gemm_fx(a))) return -22;
    if (a.ln_c && (!a.ln_d || !a.ln_stats || a.ln_slots < 1 || a.ln_slots > LN_MAX_SLOTS || a.bias || (a.N & 3) || (a.ldc & 3) || a.conv.mode != CONV_NONE)) return -22;
    if (a.row_stats_out && ((a.N & 127) || (a.ldc & 3) || (a.epi != EPI_NONE && a.epi != EPI_RESID) ||
                            (a.epi == EPI_RESID && (a.ldres & 3)))) return -22;
    if (a.norm_w && (!a.norm_out || (a.norm_ld & 7) || a.norm_ld < a.N || a.bias2 || gemm_fx(a) || a.conv.mode != CONV_NONE ||
                     (a.epi != EPI_NONE && a.epi != EPI_RESID) || (a.C && (a.ldc & 7)) || (a.epi == EPI_RESID && (a.ldres & 7)) ||
                     (!a.C && !a.norm_b) || (a.norm_res && (!a.norm_b || (a.norm_ldres & 7)))))
        return -22;
    if (a.rope_cos) {                                  // RoPE + KV append + V^T epilogue of the LLaMA prefill's qkv projection
        const int hd = a.rope_hl * 128;
        if (!a.rope_sin || !a.rope_pos || !a.rope_slot || !a.rope_kc || !a.rope_vc || !a.vt_out || a.rope_hl < 1 || a.N != 3 * hd ||
            (hd & 255) || a.vt_col0 != 2 * hd || a.vt_s != a.M || a.rope_smax < a.M || a.epi != EPI_NONE || a.bias || a.bias2 ||
            a.ln_c || a.row_stats_out || a.cross_k || a.conv.mode != CONV_NONE || (a.ldc & 7) || ((uintptr_t)a.C & 15) ||
            (a.vt_spad & 7) || ((uintptr_t)a.vt_out & 15) || a.vt_spad < a.M)
            return -22;
    }
    if (a.vt_out && (a.epi != EPI_NONE || a.conv.mode != CONV_NONE || (a.vt_col0 & 63) || ((a.N - a.vt_col0) & 63) || a.vt_col0 < 0 ||
                     a.vt_col0 >= a.N || a.vt_s < 1 || a.M % a.vt_s || a.vt_spad < a.vt_s || (a.ldc & 3))) return -22;
    if (a.cross_k && (!a.cross_vt || a.epi != EPI_NONE || a.conv.mode != CONV_NONE || a.bias || (a.N & 63) || (a.ldc & 3) || a.cross_n < 1 ||
                      a.cross_n > 64 || a.cross_npad < 64 || a.cross_rows < 64 || (a.cross_rows & 63) || a.M % a.cross_rows ||
                      (a.cross_ldk & 3) || (a.cross_npad & 3) || a.row_stats_out || a.vt_out)) return -22;
    if (a.conv.mode != CONV_NONE) {
        const ConvGeom& g = a.conv;
        if ((g.Cin & 63) || a.K != 9 * g.Cin || a.M % (g.Hout * g.Wout)) return -22;
        if (g.mode == CONV_3X3 && (g.Hout != g.Hin || g.Wout != g.Win)) return -22;
        if (g.mode == CONV_3X3_S2 && (g.Hout != (g.Hin + 1) / 2 || g.Wout != (g.Win + 1) / 2)) return -22;
        if (g.mode == CONV_3X3_UP2 && (g.Hout != 2 * g.Hin || g.Wout != 2 * g.Win)) return -22;
        GemmArgs b = a;
        const int cpt = g.Cin / 64;
        const long src_pixels = (long)(a.M / (g.Hout * g.Wout)) * g.Hin * g.Win;
        if (cpt <= 64 && src_pixels < (g.mode == CONV_3X3_UP2 ? 1L << 21 : 1L << 23) && src_pixels * g.Cin * 2 < (1L << 31)) {
            b.conv.cpt = cpt;
            b.conv.cpt_magic = (65536 + cpt - 1) / cpt;
        }
        switch (a.epi) {
            case EPI_NONE:  return launch_v2<EPI_NONE, true>(b, s);
            case EPI_RESID: return launch_v2<EPI_RESID, true>(b, s);
            default: return -22;
        }
    }
    if (a.lda & 7) return -22;
    switch (a.epi) {
        case EPI_NONE:   return launch_v2<EPI_NONE, false>(a, s);
        case EPI_RESID:  return launch_v2<EPI_RESID, false>(a, s);
        case EPI_SWIGLU: return launch_v2<EPI_SWIGLU, false>(a, s);
        case EPI_SILU:   return launch_v2<EPI_SILU, false>(a, s);
        case EPI_GELU:   return launch_v2<EPI_GELU, false>(a, s);
        case EPI_GEGLU:  return launch_v2<EPI_GEGLU, false>(a, s);
        default: return -22;
    }
}
