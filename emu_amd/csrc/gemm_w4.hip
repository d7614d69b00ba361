// 256(n) x 256(m) x 64(k) bf16 MFMA GEMM tile on FOUR waves -- one per SIMD, 256 fp32 accumulators each in AGPRs -- fed by a ring of
// five 32 KiB LDS slots.  Round 6's main loop for the large, light-epilogue GEMMs of the path (LLaMA prefill: qkv with the RoPE / KV /
// V^T epilogue, gate/up, the K-sliced o_proj / down_proj); gemm256.hip's 8-wave ping-pong tile keeps fp8 operands, the GELU / GEGLU /
// folded-LayerNorm epilogues, convs and tiles that reach past N (launch_pp has the rule and the measurements), and is the A/B twin
// everywhere (emu_gemm_tune bit 21; bit 22 takes this tile wherever it is instantiated).  C[m, n] = epilogue(sum_k A[m, k] W[n, k]).
//
// Why four waves: a wave that owns 128(n) x 128(m) of the tile reads (128 + 128) x 128 B = 32 KiB of fragments per k tile, 128 KiB
// per CU; the eight waves of the ping-pong tile (128 x 64 each) read 192 KiB, and on this chip the matrix pipe is power-limited: at
// 8192^3 a bare MFMA stream on random operands holds 1.73 GHz (1600 TFLOP/s), every LDS byte moved beside it costs clock
// (profiles/r06_gemm_w4_probe_ablations_cycles_clock.log).  One wave per SIMD overlaps its own ds_reads / LDS-DMA with its own MFMAs.
//
// Accumulators.  The MFMAs are asm statements with the register CLASSES in their constraints -- accumulators "a", fragments "v".
// Behind the MFMA builtins, whose operands may live in either file, hipcc parks fragments in AGPRs and shuffles the 256 accumulator
// registers through v_accvgpr moves and scratch (84 .. 1020 spills); with literal register names (a[0:15] ...) it does not know they
// are live and spills INTO them in the epilogue (both tried: tools/probe/gemm_w4_probe.hip, tools/w4_audit.py).  With the classes
// fixed it allocates 16 x 16 AGPRs once and never moves them.  An asm statement is opaque to the scheduler, so the order of the loop
// is pinned instruction by instruction with sched_barrier(0): per k step 16 MFMAs, the 8 ds_read_b128 of the NEXT k step's fragments
// behind MFMAs 0, 2, .. 14 (two fragment sets, 64 VGPRs) and 4 LDS-DMA pieces behind MFMAs 1, 5, 9, 13.
//
// The ring.  Half-tile h = 2 t + o (o = 0: the 256 weight rows, 1: the 256 activation rows of k tile t; 128 B per row, the 16-byte
// slot ^= (row >> 1) & 7 swizzle on the DMA source as in gemm256.hip) lives in slot h % 5; the tile loop is unrolled over the five
// ring positions, so every LDS address is a constant (one s_mov into m0 per piece, one running s_add for the source offset).  A wave
// issues 8 of a half-tile's 32 pieces (piece p = wave + 4 j: rows 8 p .. 8 p + 7), FOUR PER K STEP ALL THE TIME:
//     k step 3 of tile t-1, k step 0 of tile t : activations of tile t+1   (slot of W(t-1), free since the barrier of tile t-1)
//     k steps 1, 2 of tile t                   : weights of tile t+2       (slot of A(t-1))
// A first version with two 64 KiB buffers issued a tile's 16 pieces per wave in the k step behind the barrier: every wave bursts at
// once, the CU's single texture-address path serialises 64 instructions, and that k step takes twice its MFMA time (1225 TFLOP/s at
// 8192^3; this ring 1338 with the probe's direct epilogue, 1408-1427 here; the ping-pong tile 1239-1326).  ONE barrier per k tile,
// ahead of k step 3: by then every wave has read tile t's last fragments (lgkmcnt(0)) and waited for its own pieces of tile t+1
// (vmcnt(8): the newest eight -- weights of t+2 -- stay in flight).  Beyond the last k tile the pieces carry an out-of-range offset
// (no fetch) and the fragment reads of "tile nk" are never used, so every tile runs the same instruction stream.  Measured in the
// loop: 2250 shader cycles per k tile against the MFMA floor of 2048 (barrier ~60, reads ~90, DMA ~100), at 1.585 GHz.
// Scalar work belongs IN the MFMA gaps: left to itself hipcc emits the address arithmetic of a tile's sixteen pieces (~60 SALU) ahead
// of the tile's first fenced region, where no MFMA covers it (10 % of the loop); running offsets pass through an empty asm per gap.
//
// Ragged M ("extension", as gemm256.hip): when 0 < M mod 256 <= 16 the last row of tiles carries the remainder rows itself.  There is
// no LDS left for them, and they need none: every wave loads the rows' 16x16x32 fragments straight from global memory (2 KiB per k
// tile, the same for all waves: L1 hits), two k tiles ahead, with asm loads counted by hand beside the LDS-DMA stream, and multiplies
// them with its 64 weight rows (re-read in the 16-row layout: 8 ds_read_b128 and 8 v_mfma_f32_16x16x32_bf16 per k tile, +6 %).
//
// Epilogue: gemm256.hip's arithmetic and rounding points (bit-identical results: tests/test_gpu_gemm_cfgs.py config "W"), re-cut for
// one wave per SIMD, where nothing hides a dependent round trip: the column operands of the wave's 128 columns and the row statistics
// of its rows are fetched once, branch-free (a per-quad `if (nb < N)` made hipcc wait for every load inside its own branch: sixteen
// serialised L2 round trips, 10 of a 15 us epilogue); a run-time loop walks the four 32-row blocks (the body exists once: two unrolled
// halves were 180-320 KB of code and ran at the speed of their instruction fetch), loop-invariant inputs passing through empty asm
// statements so that LICM does not hoist several hundred registers' worth of derived values; bf16 results AND fp32 K-slices leave
// through the dead ring as whole rows.  What stays slower than on the ping-pong tile: VALU-heavy epilogues (GELU / GEGLU: no second
// wave on the SIMD to overlap the erf's dependent chains) and the direct (unstaged) store path.
// Replaces the same reference calls as gemm.hip (torch Linear on the LLaMA-prefill path: emu.py:213-229).
#include <utility>

#include "gemm_tile.h"

using namespace emu_gemm;

namespace {

constexpr int SLOT = 32768;              // one half-tile: 256 rows of 128 bytes
constexpr int SLICE = 288 * 256;         // fp32 elements of one K-slice of a tile (gemm256.hip's slab layout: pp_reduce_kernel reads it)
constexpr uint32_t OOB = 0x80000000u;    // beyond num_records of the descriptors: the load returns zeros and fetches nothing

template <int V> struct IC { static constexpr int value = V; };
template <int... Is, class F>
__device__ __forceinline__ void static_for_seq(std::integer_sequence<int, Is...>, F&& f) { (f(IC<Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_seq(std::make_integer_sequence<int, N>{}, f); }

__device__ __forceinline__ void bar() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
}

// accumulator block acc (16 AGPRs) += p x q.  An asm statement with the register CLASSES in its constraints -- accumulators "a",
// fragments "v" -- and nothing else: behind the MFMA builtins, whose operands may live in either file, hipcc's allocator parks
// fragments in AGPRs and shuffles the 256 accumulator registers through v_accvgpr moves and scratch (84 .. 1020 spills); with literal
// register names (a[0:15] ...) it does not know the accumulators are live and uses "free" AGPRs for its own spills in the epilogue
// (tools/w4_audit.py).  With the classes fixed it allocates 16 x 16 AGPRs once and never moves them.
__device__ __forceinline__ void mfma_a(f32x16_t& acc, const bf16x8_t& p, const bf16x8_t& q) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(p), "v"(q));
}
// acc (VGPRs) += p x q on the 16x16x32 shape (remainder rows)
__device__ __forceinline__ void mfma16_v(f32x4_t& acc, const bf16x8_t& p, const u32x4& q) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(p), "v"(q));
}

// rows of tiles and remainder rows carried by the last one (gemm256.hip::pp_tiles_m)
__host__ __device__ inline int w4_tiles_m(int M, bool allow_ext, int& ext_rows) {
    const int tm = M >> 8, r = M & 255;
    ext_rows = (allow_ext && tm >= 1 && r > 0 && r <= 16) ? r : 0;
    return ext_rows ? tm : (M + 255) >> 8;
}

template <int EPI, bool CONV, int FX = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_w4_kernel(const GemmArgs a) {
    __shared__ __attribute__((aligned(16))) char smem[5 * SLOT];
    const int tid = threadIdx.x, lane = tid & 63;
    EMU_TRACE_MARK(a.trace, 0);
    const uint32_t pfd = prefetch_lines(a.pf_ptr, a.pf_bytes, blockIdx.x * 256u + tid, gridDim.x * 256u);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;

    // ---- tile of this workgroup: gemm256.hip's order (whole-K tiles, then K-slices; XCD-aware bijective remap; 2-D tile blocks)
    const int b = blockIdx.x;
    int ext_rows;
    const int tiles_m = w4_tiles_m(a.M, !CONV, ext_rows);
    auto xcd_order = [](int i, int n) {
        const int xcd = i & 7, q8 = n >> 3, r8 = n & 7;
        return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (i >> 3);
    };
    int wg, ks = 0, nsl = 1;
    if (b < a.full_tiles) {
        wg = xcd_order(b, a.full_tiles);
        if (a.sup_m) {
            const int per = a.sup_m * a.sup_n, blk = wg / per, r = wg - blk * per;
            const int bm = tiles_m / a.sup_m, bi = blk % bm, bj = blk / bm;
            wg = (bj * a.sup_n + r / a.sup_m) * tiles_m + bi * a.sup_m + r % a.sup_m;
        }
    } else {
        nsl = a.ksplit;
        const int rest = tiles_m * ((a.N + 255) >> 8) - a.full_tiles;
        const int l = xcd_order(b - a.full_tiles, rest * nsl);
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
    const bool ext = !CONV && (FX == 0 || (FX & FX_ROPE) != 0) && ext_rows > 0 && tm == tiles_m - 1;

    // ---- LDS-DMA sources.  Piece j (0 .. 7) of a wave fills rows prow + 32 j of a half-tile, prow = wave*8 + lane/8, 16-byte slot
    // lane % 8 <- source chunk slot ^ ((row >> 1) & 7) = slot ^ ((wave & 1) << 2 | lane >> 4): one per-lane offset per operand, the
    // row distance of a piece and the k offset ride in soffset.  Rows past N - 1 / M - 1 lie beyond the descriptors: zeros.
    const int prow = wave * 8 + (lane >> 3);
    const uint32_t sck = (uint32_t)(((lane & 7) ^ (((wave & 1) << 2) | (lane >> 4))) * 16);
    const uint32_t vW = (uint32_t)(n0 + prow) * (uint32_t)a.ldw * 2u + sck;
    const uint32_t vA = CONV ? 0u : (uint32_t)(m0 + prow) * (uint32_t)a.lda * 2u + sck;     // plain GEMM (conv: only "not OOB")
    uint32_t qpix[8];                                 // CONV: tap mask and centre source pixel of the row of every activation piece
    if constexpr (CONV) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int gm = m0 + prow + 32 * j;
            gm = gm < a.M ? gm : a.M - 1;
            const int hw = a.conv.Hout * a.conv.Wout;
            const int pb = gm / hw, rr = gm - pb * hw;
            const int py = rr / a.conv.Wout, px = rr - py * a.conv.Wout;
            const bool up = a.conv.mode == CONV_3X3_UP2;
            const int cy = up ? py >> 1 : (a.conv.mode == CONV_3X3_S2 ? 2 * py : py);
            const int cx = up ? px >> 1 : (a.conv.mode == CONV_3X3_S2 ? 2 * px : px);
            uint32_t mask = 0;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                int yi, xi;
                mask |= (uint32_t)conv_tap(a.conv, py, px, t / 3, t % 3, yi, xi) << t;
            }
            qpix[j] = (mask << 23) | (uint32_t)((pb * a.conv.Hin + cy) * a.conv.Win + cx) |
                      (up ? (uint32_t)(((py & 1) << 22) | ((px & 1) << 21)) : 0u);
        }
    }
    const uint32_t w_bytes = (uint32_t)a.N * (uint32_t)a.ldw * 2u;
    const uint32_t a_bytes = CONV ? 0x7fffffffu : (uint32_t)a.M * (uint32_t)a.lda * 2u;
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)a.W, 0, w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)a.A, 0, a_bytes, 0x00020000);
    auto dma = [&](const __amdgpu_buffer_rsrc_t& r, uint32_t voff, int soff, char* lds_wave_base) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, soff, 0, 0);
    };
    const int nk_all = a.K >> 6;
    const int kt0 = (int)((long)ks * nk_all / nsl);
    const int nk = (int)((long)(ks + 1) * nk_all / nsl) - kt0;
    const int rsW = 32 * a.ldw * 2, rsA = 32 * a.lda * 2;
    const int wl = wave * 1024;
    // Scalar operands of a piece are computed IN the MFMA gap that issues it: every base passes through an empty asm there.  (Left to
    // itself hipcc emits the address arithmetic of all sixteen pieces of a tile -- ~60 SALU instructions -- ahead of the tile's first
    // fenced region, where no MFMA covers it: 10 % of the loop.)
    auto lau = [](int x) { asm volatile("" : "+s"(x)); return x; };
    // CONV: filter tap of a k tile (wave-uniform): tap index, channel offset in bytes, (ky, kx); computed once per window of four pieces
    struct TapInfo { int tap, cob, ky, kx; };
    auto tap_of = [&](int tau) {
        const int ktile = kt0 + lau(tau);
        const int tap = (ktile * a.conv.cpt_magic) >> 16, ky = (tap * 11) >> 5;
        return TapInfo{tap, ((ktile - tap * a.conv.cpt) << 6) * 2, ky, tap - ky * 3};
    };
    // One piece: `lds` = LDS offset of the slot + 4096 j (a compile-time constant: the tile loop is unrolled over the five slots), so the
    // destination costs one s_add into m0; the source offset `so` is a RUNNING scalar of the window (k offset of the tile + 32 j rows),
    // advanced by the caller: one s_add per piece.  v = the operand's lane offset, or OOB for a tile beyond the last (set per window).
    auto pieceW = [&](int lds, int so, uint32_t v) { dma(rW, v, so, smem + wl + lds); };
    auto pieceA = [&](int lds, int so, uint32_t v, int j, const TapInfo& ti) {
        char* dst = smem + wl + lds;
        if constexpr (!CONV) {
            dma(rA, v, so, dst);
        } else {
            // implicit-GEMM gather (gemm256.hip): a 64-wide k tile lies inside one filter tap (Cin % 64 == 0); v = 0: a tile beyond the last
            const uint32_t o = qpix[j];
            uint32_t src;
            if (a.conv.mode != CONV_3X3_UP2) {
                src = (o & 0x7fffffu) + (uint32_t)((ti.ky - 1) * a.conv.Win + (ti.kx - 1));
            } else {
                const int dy = (int)(((o >> 22) & 1u) + ti.ky - 1) >> 1, dx = (int)(((o >> 21) & 1u) + ti.kx - 1) >> 1;
                src = (o & 0x1fffffu) + (uint32_t)(dy * a.conv.Win + dx);
            }
            const uint32_t off = __umul24(src, (uint32_t)a.conv.Cin * 2u) + sck;
            dma(rA, (v != OOB && ((o >> (23 + ti.tap)) & 1u)) ? off : OOB, ti.cob, dst);
        }
    };

    // ---- fragments: lane (l31, hi) reads row base + l31, chunk (2 kk + hi) ^ ((l31 >> 1) & 7); k step kk = bits 5, 6 of the offset
    const int lp0 = l31 * 128 + ((hi ^ ((l31 >> 1) & 7)) << 4);
    const int pb = lp0 + wr * 16384, qb = lp0 + wc * 16384;
    bf16x8_t f0p[4], f0q[4], f1p[4], f1q[4];
    f32x16_t acc[16];                                // block b = 4 j + i: weight rows wr*128 + 32 i .., activation rows wc*128 + 32 j ..
#pragma unroll
    for (int bb = 0; bb < 16; ++bb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[bb][r] = 0.f;
    auto rd1 = [&](bf16x8_t (&p)[4], bf16x8_t (&q)[4], int pa, int qa, int i) {
        if (i < 4) p[i] = *reinterpret_cast<const bf16x8_t*>(smem + (pa + i * 4096));
        else q[i - 4] = *reinterpret_cast<const bf16x8_t*>(smem + (qa + (i - 4) * 4096));
    };

    // ---- remainder rows (ext): fragments straight from global memory.  Lane (l15, q4) holds row M0x + l15, k = 32 s + 8 q4 .. + 7 of
    // the k tile (k-step s = 0, 1): two 16-byte loads per k tile, issued two tiles ahead into a ring of three register pairs.  They
    // are asm loads (hipcc would wait vmcnt(0) for a register load beside LDS-DMA in flight and drain the ring); their place in the
    // wave's VMEM order is fixed -- behind the fourth piece of k step 0 -- so the counted waits below know them.
    const int l15 = lane & 15, q4 = lane >> 4;
    u32x4 xf[5][2];                                  // a ring of five register pairs: tile t in pair t % 5 (the loop is unrolled over five)
#pragma unroll
    for (int i = 0; i < 5; ++i) xf[i][0] = xf[i][1] = u32x4{0u, 0u, 0u, 0u};
    f32x4_t accx[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    uint32_t vX = OOB;
    if (ext) {
        const int mx = m0 + 256 + l15;
        vX = mx < a.M ? (uint32_t)mx * (uint32_t)a.lda * 2u + (uint32_t)q4 * 16u : OOB;
    }
    // (the descriptor of the activations once more, as four dwords an asm operand can name: base, stride 0, num_records, raw dword format)
    const u32x4 dA = {(uint32_t)(uintptr_t)a.A, (uint32_t)((uintptr_t)a.A >> 32) & 0xffffu, a_bytes, 0x00020000u};
    auto xload = [&](u32x4 (&d)[2], int tau) {
        const uint32_t v = tau < nk ? vX : OOB;
        const int so = __builtin_amdgcn_readfirstlane((kt0 + tau) << 7);      // (an "s" operand must be provably wave-uniform)
        asm volatile("buffer_load_dwordx4 %0, %2, %3, %4 offen\n\tbuffer_load_dwordx4 %1, %2, %3, %4 offen offset:64"
                     : "=&v"(d[0]), "=&v"(d[1]) : "v"(v), "s"(dA), "s"(so) : "memory");
    };
    // this wave's 64 weight rows in the 16-row layout: rows wr*128 + wc*64 + 16 h + l15, chunk (4 s + q4) ^ ((row >> 1) & 7)
    const int lx0 = (wr * 128 + wc * 64 + l15) * 128 + ((q4 ^ ((l15 >> 1) & 7)) << 4);

    // one k step: 16 MFMAs on (p, q); behind MFMAs 0, 2, .. 14 the fragment reads of the next k step (np, nq <- pa, qa); behind MFMAs
    // 1, 5, 9, 13 one LDS-DMA piece each (dmaf(0 .. 3)); xf(idx) may add work behind MFMA idx (remainder rows)
    auto kstep = [&](const bf16x8_t (&p)[4], const bf16x8_t (&q)[4], bf16x8_t (&np)[4], bf16x8_t (&nq)[4], int pa, int qa, auto&& dmaf, auto&& xfn) {
        __builtin_amdgcn_sched_barrier(0);
        static_for<16>([&](auto ic) {
            constexpr int idx = decltype(ic)::value, i = idx & 3, j = idx >> 2;
            mfma_a(acc[j * 4 + i], p[i], q[j]);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr ((idx & 3) == 1) dmaf(IC<(idx >> 2)>{});
            if constexpr ((idx & 1) == 0) rd1(np, nq, pa, qa, idx >> 1);
            xfn(ic);
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    auto none = [](auto) {};

    // ---- prologue: W(0) -> slot 0, A(0) -> slot 1, W(1) -> slot 2, first half of A(1) -> slot 3 (remainder rows of tiles 0, 1 ahead of
    // them, so that the counted waits see the steady-state order)
    if (ext) { xload(xf[0], 0); xload(xf[1], 1); }
    TapInfo ti{0, 0, 0, 0};                            // CONV: tap of the activation window in flight
    {
        const uint32_t v1w = nk > 1 ? vW : OOB, v1a = nk > 1 ? vA : OOB;
        if constexpr (CONV) ti = tap_of(0);
#pragma unroll
        for (int j = 0; j < 8; ++j) pieceW(0 * SLOT + j * 4096, (kt0 << 7) + j * rsW, vW);
#pragma unroll
        for (int j = 0; j < 8; ++j) pieceA(1 * SLOT + j * 4096, (kt0 << 7) + j * rsA, vA, j, ti);
        if constexpr (CONV) ti = tap_of(1);
#pragma unroll
        for (int j = 0; j < 8; ++j) pieceW(2 * SLOT + j * 4096, ((kt0 + 1) << 7) + j * rsW, v1w);
#pragma unroll
        for (int j = 0; j < 4; ++j) pieceA(3 * SLOT + j * 4096, ((kt0 + 1) << 7) + j * rsA, v1a, j, ti);
    }
    // running state of the two issue windows: source offset of the next piece, lane offset (or OOB) of the window's operand
    int soA = ((kt0 + 1) << 7) + 4 * rsA, soW = (kt0 + 2) << 7;
    uint32_t vwA = nk > 1 ? vA : OOB, vwW = OOB;
    const int nxtA = 128 - 7 * rsA, nxtW = 128 - 7 * rsW;       // from a window's last piece to the next k tile's first
    wait_vmcnt<12>();
    bar();
    prefetch_release(pfd);
    EMU_TRACE_MARK(a.trace, 1);
#pragma unroll
    for (int i = 0; i < 8; ++i) rd1(f0p, f0q, pb, qb + SLOT, i);

    // One k tile.  SW = slot of W(t) (t % 5 -> 0, 2, 4, 1, 3); A(t) sits in SW + 1, W(t+1) in SW + 2, A(t+1) in SW + 3, W(t+2) goes
    // to SW + 4 and A(t+2) to SW (all mod 5): compile-time constants, the loop below is unrolled over the five positions of the ring.
    auto tile = [&](int t, auto swc, auto xc) {
        constexpr int SW = decltype(swc)::value, S_A = (SW + 1) % 5, S_W1 = (SW + 2) % 5, S_A1 = (SW + 3) % 5, S_W2 = (SW + 4) % 5, S_A2 = SW;
        constexpr int XS = decltype(xc)::value;        // register pair of this tile's remainder rows (= the ring position); -1: no remainder rows
        const int pa = pb + SW * SLOT, qa = qb + S_A * SLOT;
        auto dA1 = [&](auto jc) {                      // k step 0: activation pieces 4 .. 7 of tile t + 1
            constexpr int j = 4 + decltype(jc)::value;
            pieceA(S_A1 * SLOT + j * 4096, soA, vwA, j, ti);
            soA = lau(soA) + (j == 7 ? nxtA : rsA);
        };
        auto dW2a = [&](auto jc) {                     // k step 1: weight pieces 0 .. 3 of tile t + 2
            constexpr int j = decltype(jc)::value;
            if constexpr (j == 0) vwW = t + 2 < nk ? vW : OOB;
            pieceW(S_W2 * SLOT + j * 4096, soW, vwW);
            soW = lau(soW) + rsW;
        };
        auto dW2b = [&](auto jc) {                     // k step 2: weight pieces 4 .. 7 of tile t + 2
            constexpr int j = 4 + decltype(jc)::value;
            pieceW(S_W2 * SLOT + j * 4096, soW, vwW);
            soW = lau(soW) + (j == 7 ? nxtW : rsW);
        };
        auto dA2 = [&](auto jc) {                      // k step 3: activation pieces 0 .. 3 of tile t + 2
            constexpr int j = decltype(jc)::value;
            if constexpr (j == 0) {
                vwA = t + 2 < nk ? vA : OOB;
                if constexpr (CONV) ti = tap_of(t + 2);
            }
            pieceA(S_A2 * SLOT + j * 4096, soA, vwA, j, ti);
            soA = lau(soA) + rsA;
        };
        if constexpr (XS < 0) {
            kstep(f0p, f0q, f1p, f1q, pa ^ 32, qa ^ 32, dA1, none);
            kstep(f1p, f1q, f0p, f0q, pa ^ 64, qa ^ 64, dW2a, none);
            kstep(f0p, f0q, f1p, f1q, pa ^ 96, qa ^ 96, dW2b, none);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            wait_vmcnt<8>();
        } else {
            // remainder rows: the loads of tile t + 2 behind the last piece of k step 0; this wave's 64 weight rows of tile t in the
            // 16-row layout, read in k step 1 (8 ds_read_b128 behind the odd MFMAs), multiplied in k step 2 behind MFMAs 1, 3, .. 15
            // (a v_mfma_f32_16x16x32_bf16 holds the pipe for 16 of the 32 cycles of the gap it sits in)
            bf16x8_t wx[4][2];
            constexpr int XN = (XS + 2) % 5;
            kstep(f0p, f0q, f1p, f1q, pa ^ 32, qa ^ 32, dA1,
                  [&](auto ic) { if constexpr (decltype(ic)::value == 13) xload(xf[XN], t + 2); });
            const int wxa = lx0 + SW * SLOT;
            kstep(f1p, f1q, f0p, f0q, pa ^ 64, qa ^ 64, dW2a,
                  [&](auto ic) {
                      constexpr int idx = decltype(ic)::value;
                      if constexpr ((idx & 1) == 1) {
                          constexpr int h = idx >> 2, s2 = (idx >> 1) & 1;
                          wx[h][s2] = *reinterpret_cast<const bf16x8_t*>(smem + ((wxa ^ (s2 << 6)) + h * 2048));
                      }
                  });
            // xf[XS] (tile t) was issued two tiles ago and is older than every piece of tile t, which the counted wait ahead of the
            // previous barrier (or the prologue's) has seen land: no wait here
            asm volatile("" : "+v"(xf[XS][0]), "+v"(xf[XS][1]));
            kstep(f0p, f0q, f1p, f1q, pa ^ 96, qa ^ 96, dW2b,
                  [&](auto ic) {
                      constexpr int idx = decltype(ic)::value;
                      if constexpr ((idx & 1) == 1) {
                          constexpr int h = idx >> 2, s2 = (idx >> 1) & 1;
                          // (an asm statement with the accumulator in VGPRs: behind the builtin hipcc keeps accx in a[0:15] -- it does not
                          // know the named accumulators -- and block 0 is gone)
                          mfma16_v(accx[h], wx[h][s2], xf[XS][s2]);
                      }
                  });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // newest: the 8 weight pieces of t+2.  The remainder pair of t+2 (issued in k step 0, older than those) is waited for as well:
            // hipcc counts an asm load's destination as written at the statement and may copy it at the loop's back edge
            wait_vmcnt<8>();
        }
        bar();
        kstep(f1p, f1q, f0p, f0q, pb + S_W1 * SLOT, qb + S_A1 * SLOT, dA2, none);
    };
    if (!ext) {
        for (int t = 0; t < nk; t += 5) {
            tile(t, IC<0>{}, IC<-1>{});
            if (t + 1 < nk) tile(t + 1, IC<2>{}, IC<-1>{});
            if (t + 2 < nk) tile(t + 2, IC<4>{}, IC<-1>{});
            if (t + 3 < nk) tile(t + 3, IC<1>{}, IC<-1>{});
            if (t + 4 < nk) tile(t + 4, IC<3>{}, IC<-1>{});
        }
    } else {
        for (int t = 0; t < nk; t += 5) {
            tile(t, IC<0>{}, IC<0>{});
            if (t + 1 < nk) tile(t + 1, IC<2>{}, IC<1>{});
            if (t + 2 < nk) tile(t + 2, IC<4>{}, IC<2>{});
            if (t + 3 < nk) tile(t + 3, IC<1>{}, IC<3>{});
            if (t + 4 < nk) tile(t + 4, IC<3>{}, IC<4>{});
        }
    }
    EMU_TRACE_MARK(a.trace, 2);
#ifdef EMU_TRACE
    struct TraceEnd { unsigned long long* t; __device__ ~TraceEnd() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); EMU_TRACE_MARK(t, 3); } } trace_end{a.trace};
#endif
    // the last MFMAs' results -> v_accvgpr_read (12 wait states for an 8-pass MFMA; the asm statements carry no hazard information),
    // and the stale pieces / remainder loads of "tiles nk, nk + 1" are gone before the ring is reused as the epilogue's stage
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_waitcnt vmcnt(0)" ::: "memory");
    // the remainder-row registers stay allocated until their last loads (of "tiles nk, nk + 1": zeros, never used) have landed --
    // hipcc counts an asm load's destination as written at the statement, finds the value dead and hands the register to an
    // address while the load is still in flight (memory access fault: measured)
#pragma unroll
    for (int i = 0; i < 5; ++i) asm volatile("" ::"v"(xf[i][0]), "v"(xf[i][1]));

    // ================= epilogue =================
    // accumulator block 4 j + i' holds rows n = n0 + wr*128 + i'*32 + 8 g + 4 hi + e (register 4 g + e), column m = m0 + wc*128 + j*32 + l31.
    // load_half (the RoPE epilogue's form): half mh as gemm256.hip's (x, y, i) = block (2 mh + y) * 4 + 2 x + i, wq = 2 wc + mh
    auto load_half = [&](auto mhc, f32x16_t (&h)[2][2][2]) {
        constexpr int mh = decltype(mhc)::value;
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    h[x][y][i] = acc[(2 * mh + y) * 4 + 2 * x + i];
                    asm volatile("" : "+v"(h[x][y][i]));
                }
    };
    auto emit = [&](int m, int nb, float (&v)[4], RowFx& fx, const QuadIn& q) {
        if (nsl > 1) {                                 // raw fp32 slice tile; pp_reduce_kernel applies the epilogue
            float* dst = a.slab_rows ? a.partial + ((size_t)ks * a.M + m) * a.N + nb
                                     : a.partial + ((size_t)(wg - a.full_tiles) * nsl + ks) * SLICE + (size_t)(m - m0) * 256 + (nb - n0);
            *reinterpret_cast<f32x4_t*>(dst) = f32x4_t{v[0], v[1], v[2], v[3]};
        } else {
            store_quad<EPI, FX>(a, m, nb, v, fx, q);
        }
    };
    // remainder rows: lane (l15, q4) holds row m0 + 256 + l15, columns 16 h + 4 q4 .. + 3 of this wave's 64 weight rows wr*128 + wc*64 ..
    const int xcol0 = wr * 128 + wc * 64;

    // ---- LLaMA prefill qkv projection: RoPE + KV append + V^T out of the epilogue (gemm256.hip has the story)
    if constexpr ((FX & FX_ROPE) != 0) {
        using StR = EpiStage<256, 256, 256>;
        using StT = EpiStageT<256, 256, 256>;
        const int HDc = a.rope_hl * 128, region = n0 / HDc;      // 0: q, 1: k, 2: v
        __syncthreads();
        static_for<2>([&](auto mhc) {
            constexpr int mh = decltype(mhc)::value;
            f32x16_t hacc[2][2][2];
            load_half(mhc, hacc);
            const int wq = 2 * wc + mh;
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y) {
                    const int row = wq * 64 + y * 32 + l31;
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int col = wr * 128 + x * 64 + i * 32 + 8 * g + 4 * hi;
                            u32x2 ov;
                            ov.x = packbf(hacc[x][y][i][4 * g], hacc[x][y][i][4 * g + 1]);
                            ov.y = packbf(hacc[x][y][i][4 * g + 2], hacc[x][y][i][4 * g + 3]);
                            *reinterpret_cast<u32x2*>(smem + StR::off(row, col)) = ov;
                        }
                }
        });
        if (ext) {                                     // remainder rows: LDS rows 256 .. 271 behind the tile
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                u32x2 ov;
                ov.x = packbf(accx[h][0], accx[h][1]);
                ov.y = packbf(accx[h][2], accx[h][3]);
                *reinterpret_cast<u32x2*>(smem + StR::off(256 + l15, xcol0 + 16 * h + 4 * q4)) = ov;
            }
        }
        __syncthreads();
        const int nrows = ext ? 272 : 256;
        if (region < 2) {
            const int head0 = (n0 - region * HDc) >> 7;
            for (int idx = tid; idx < nrows * 16; idx += 256) {
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
            for (int idx = tid; idx < nrows * 32; idx += 256) {
                const int row = idx >> 5, ch = idx & 31, m = m0 + row;
                if (m >= a.M) continue;
                st16(a.rope_vc + ((size_t)(head0 + (ch >> 4)) * a.rope_smax + a.rope_slot[m]) * 128 + (ch & 15) * 8,
                     *reinterpret_cast<const u32x4*>(smem + StR::off(row, ch * 8)));
            }
            __syncthreads();                           // the row-major tile has been read: the transposed one takes its place
            static_for<2>([&](auto mhc) {
                constexpr int mh = decltype(mhc)::value;
                f32x16_t hacc[2][2][2];
                load_half(mhc, hacc);
                const int wq = 2 * wc + mh;
#pragma unroll
                for (int x = 0; x < 2; ++x)
#pragma unroll
                    for (int y = 0; y < 2; ++y) {
                        const int row = wq * 64 + y * 32 + l31;
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                const int col = wr * 128 + x * 64 + i * 32 + 8 * g + 4 * hi;
                                const uint32_t p0 = packbf(hacc[x][y][i][4 * g], hacc[x][y][i][4 * g + 1]);
                                const uint32_t p1 = packbf(hacc[x][y][i][4 * g + 2], hacc[x][y][i][4 * g + 3]);
                                *reinterpret_cast<bf16_t*>(smem + StT::off(col, row)) = (bf16_t)(p0 & 0xffffu);
                                *reinterpret_cast<bf16_t*>(smem + StT::off(col + 1, row)) = (bf16_t)(p0 >> 16);
                                *reinterpret_cast<bf16_t*>(smem + StT::off(col + 2, row)) = (bf16_t)(p1 & 0xffffu);
                                *reinterpret_cast<bf16_t*>(smem + StT::off(col + 3, row)) = (bf16_t)(p1 >> 16);
                            }
                    }
            });
            __syncthreads();
            StT::store(smem, a, m0, n0);               // one batch element: b = 0, key index = row index (launch_gemm checks)
            if (ext) {                                 // remainder rows: a few 2-byte stores per lane
                const int m = m0 + 256 + l15;
                if (m < a.M) {
#pragma unroll
                    for (int h = 0; h < 4; ++h)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int n = n0 + xcol0 + 16 * h + 4 * q4 + e;
                            const uint32_t pk = packbf(accx[h][e], 0.f);
                            a.vt_out[(size_t)(n - a.vt_col0) * a.vt_spad + m] = (bf16_t)(pk & 0xffffu);
                        }
                }
            }
        }
        return;
    }

    // ---- K-slice of a tile that lies inside N: the raw fp32 tile leaves through LDS as well, 128 columns at a time (256 rows x 512 B
    // = 128 KiB).  The waves that own the half (wr) drop their accumulator quads as 16-byte LDS writes, the 16-byte slot of a row XORed
    // with the row (32 lanes = 32 rows would otherwise share a bank); then all four waves store whole 512-byte row segments.  Against
    // 16 bytes per lane to 32 different rows straight from the accumulators: a quarter of the store instructions, every one of them
    // coalesced (S = 770 o_proj, same-run: profiles/r06_gemm_w4_sliced_ab.log).
    const bool slab_stage = (a.stage & 2) != 0 && nsl > 1 && n0 + 256 <= a.N;
    if (slab_stage) {
        float* sbase = a.slab_rows ? a.partial + ((size_t)ks * a.M + m0) * a.N + n0
                                   : a.partial + ((size_t)(wg - a.full_tiles) * nsl + ks) * SLICE;
        const size_t sld = a.slab_rows ? (size_t)a.N : 256;
        __syncthreads();                               // every wave is out of the loop
        static_for<2>([&](auto hc) {
            constexpr int H = decltype(hc)::value;
            if (wr == H) {
                static_for<4>([&](auto jc) {
                    constexpr int J = decltype(jc)::value;
                    const int row = wc * 128 + J * 32 + l31;
#pragma unroll
                    for (int x = 0; x < 2; ++x)
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            f32x16_t t = acc[J * 4 + 2 * x + i];
                            asm volatile("" : "+v"(t));
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                const int slot = x * 16 + i * 8 + 2 * g + hi;
                                *reinterpret_cast<f32x4_t*>(smem + row * 512 + ((slot ^ (row & 31)) << 4)) =
                                    f32x4_t{t[4 * g], t[4 * g + 1], t[4 * g + 2], t[4 * g + 3]};
                            }
                        }
                });
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 32; ++r) {
                const int idx = r * 256 + tid, row = idx >> 5, ls = (idx & 31) ^ (row & 31);
                const f32x4_t v = *reinterpret_cast<const f32x4_t*>(smem + idx * 16);
                if (m0 + row < a.M) *reinterpret_cast<f32x4_t*>(sbase + (size_t)row * sld + H * 128 + ls * 4) = v;
            }
            __syncthreads();
        });
    }

    // ---- staged epilogue (gemm_tile.h::EpiStage): the 256 x 256 results leave through the (now dead) ring as whole rows
    constexpr bool GLU = EPI == EPI_SWIGLU || EPI == EPI_GEGLU;
    using Stage = EpiStage<256, GLU ? 128 : 256, 256>;
    bool staged = (a.stage & 1) != 0 && nsl == 1 && n0 + 256 <= a.N;
    bool vt_tile = false;                              // a whole tile of V columns: transposed staging (gemm_tile.h::EpiStageT)
    if constexpr ((FX & FX_VT) != 0) {
        vt_tile = staged && a.stage_vt && n0 >= a.vt_col0;
        staged = staged && (n0 + 256 <= a.vt_col0 || vt_tile);
    }
    using StageT = EpiStageT<256, 256, 256>;
    if (staged) {
        __syncthreads();                               // every wave is out of the loop
        if constexpr (EPI == EPI_RESID) Stage::load(smem, a.res, a.ldres, m0, n0, a.M);
    }
    // One wave per SIMD: nothing hides a dependent global round trip, so the epilogue makes as few as it can.  Column-only operands
    // (bias / fused-LayerNorm vectors) of the wave's 128 columns and the row statistics of its 128 rows are fetched ONCE, up front
    // (one round trip); then a run-time loop walks the four 32-row blocks j of the wave's tile -- 64 accumulator values per lane at a
    // time, the body exists once (two unrolled halves made the kernels 180-320 KB of code, twice the ping-pong tile's: a straight-line
    // epilogue that long runs at the speed of its instruction fetch).
    QuadIn qin[2][2][4];                               // [x][i][g]: columns n0 + wr*128 + x*64 + i*32 + 8 g + 4 hi ..
    // (branch-free: ONE wave-uniform condition around all sixteen loads and a clamped column for quads that reach past N -- their
    // values are never used, store_quad's scalar path re-reads what it needs.  With a per-quad `if (nb < N)` hipcc branches around every
    // load and waits for it inside its branch: sixteen serialised L2 round trips, 10 us of a 15 us epilogue)
    if (nsl == 1) {
        int nq[2][2][4];
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int nb = n0 + wr * 128 + x * 64 + i * 32 + 8 * g + 4 * hi;
                    nq[x][i][g] = nb + 3 < a.N ? nb : 0;
                }
        if constexpr ((FX & FX_LN) != 0) {
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        qin[x][i][g].c = *reinterpret_cast<const f32x4_t*>(a.ln_c + nq[x][i][g]);
                        qin[x][i][g].d = *reinterpret_cast<const f32x4_t*>(a.ln_d + nq[x][i][g]);
                    }
        } else if (a.bias) {
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int g = 0; g < 4; ++g) qin[x][i][g].bias = *reinterpret_cast<const u32x2*>(a.bias + nq[x][i][g]);
        }
    }
    RowFx rowfx[4];
    if constexpr ((FX & FX_LN) != 0) {                 // fused LayerNorm, consumer side: the four rows of this lane in one batch of loads
        if (nsl == 1) {
            int mr[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { const int m = m0 + wc * 128 + j * 32 + l31; mr[j] = m < a.M ? m : a.M - 1; }
            LnRaw<4> raw;
            ln_rows_load<4>(a, mr, raw);
            ln_rows_finish<4>(a, raw, rowfx);
        }
    }
    if (staged) {
        if constexpr (EPI == EPI_RESID) {
            wait_vmcnt<0>();
            __syncthreads();
        }
    }
    f32x16_t h[2][2];                                  // [x][i] of row block j = accumulator block 4 j + 2 x + i
#pragma clang loop unroll(disable)
    for (int j = slab_stage ? 4 : 0; j < 4; ++j) {
        RowFx fx;
        auto pick = [&](auto jc) {
            constexpr int J = decltype(jc)::value;
            // (a statement with a side effect: without it hipcc turns the four branches into selects, reads all 256 accumulators into
            // VGPRs at once and spills several hundred registers)
            asm volatile("; row block %c0" ::"n"(J));
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    h[x][i] = acc[J * 4 + 2 * x + i];
                    asm volatile("" : "+v"(h[x][i]));      // the copy lives in VGPRs (left free, hipcc keeps it in AGPRs and spills accumulators)
                }
            fx = rowfx[J];
        };
        if (j == 0) pick(IC<0>{}); else if (j == 1) pick(IC<1>{}); else if (j == 2) pick(IC<2>{}); else pick(IC<3>{});
        // everything the body derives from loop-invariant values (column indices, masks, the bias quads unpacked to floats: several
        // hundred registers for the sixteen quads) would be hoisted out of this loop and spilled; the lane's column base and the packed
        // bias words pass through empty asm statements here, so what depends on them is computed where it is used
        int hl = hi;
        asm volatile("" : "+v"(hl));
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) asm volatile("" : "+v"(qin[x][i][g].bias.x), "+v"(qin[x][i][g].bias.y));
        const int row = wc * 128 + j * 32 + l31, m = m0 + row;
        if (staged) {
            if (m < a.M || ((FX & FX_VT) != 0 && vt_tile)) {     // (see gemm.hip: pad keys of a V^T tile stay finite)
                if (a.bias2) {
#pragma unroll
                    for (int x = 0; x < 2; ++x)
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                const int nb = n0 + wr * 128 + x * 64 + i * 32 + 8 * g + 4 * hl;
                                qin[x][i][g].bias2 = *reinterpret_cast<const u32x2*>(a.bias2 + (size_t)(m / a.rows_per_batch) * a.ld_bias2 + nb);
                            }
                }
#pragma unroll
                for (int x = 0; x < 2; ++x)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int col = wr * 128 + x * 64 + i * 32 + 8 * g + 4 * hl;
                            float v[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = h[x][i][4 * g + e];
                            if constexpr (GLU) {
                                const u32x2 ov = quad_value<EPI, FX>(a, v, fx, qin[x][i][g]);
                                *reinterpret_cast<uint32_t*>(smem + Stage::off(row, col >> 1)) = ov.x;
                            } else if ((FX & FX_VT) != 0 && vt_tile) {
                                const u32x2 ov = quad_value<EPI, FX>(a, v, fx, qin[x][i][g]);
                                *reinterpret_cast<bf16_t*>(smem + StageT::off(col, row)) = (bf16_t)(ov.x & 0xffffu);
                                *reinterpret_cast<bf16_t*>(smem + StageT::off(col + 1, row)) = (bf16_t)(ov.x >> 16);
                                *reinterpret_cast<bf16_t*>(smem + StageT::off(col + 2, row)) = (bf16_t)(ov.y & 0xffffu);
                                *reinterpret_cast<bf16_t*>(smem + StageT::off(col + 3, row)) = (bf16_t)(ov.y >> 16);
                            } else {
                                u32x2* cell = reinterpret_cast<u32x2*>(smem + Stage::off(row, col));
                                if constexpr (EPI == EPI_RESID) qin[x][i][g].res = *cell;
                                *cell = quad_value<EPI, FX>(a, v, fx, qin[x][i][g]);
                            }
                        }
            }
        } else if (m < a.M) {
            // direct stores: the row's residual / per-batch bias for all of its 16 quads before its first store (gemm_tile.h::QuadIn)
            if (nsl == 1 && (a.bias2 || EPI == EPI_RESID)) {
                const bool al = EPI != EPI_RESID || ((a.ldc | a.ldres) & 3) == 0;      // quad_full's alignment condition of the 8-byte accesses
#pragma unroll
                for (int x = 0; x < 2; ++x)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int nb = n0 + wr * 128 + x * 64 + i * 32 + 8 * g + 4 * hl;
                            const int nc = (nb + 3 < a.N && al) ? nb : 0;           // clamped: a quad that is not full never uses these values
                            if (a.bias2) qin[x][i][g].bias2 = *reinterpret_cast<const u32x2*>(a.bias2 + (size_t)(m / a.rows_per_batch) * a.ld_bias2 + nc);
                            if constexpr (EPI == EPI_RESID) qin[x][i][g].res = *reinterpret_cast<const u32x2*>(a.res + (size_t)m * a.ldres + (al ? nc : 0));
                        }
            }
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int nb = n0 + wr * 128 + x * 64 + i * 32 + 8 * g + 4 * hl;
                        if (nb >= a.N) continue;
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = h[x][i][4 * g + e];
                        emit(m, nb, v, fx, qin[x][i][g]);
                    }
        }
        // fused LayerNorm, producer side: this wave's 128 columns of a row = one statistics slot, halves in lanes l / l + 32
        if constexpr ((FX & FX_STATS) != 0) {
            const int nslot = n0 + wr * 128;
            if (nsl == 1 && nslot < a.N) {
                const float sm = fx.rs + __shfl_xor(fx.rs, 32, 64), q = fx.rq + __shfl_xor(fx.rq, 32, 64);
                if (hi == 0 && m < a.M)
                    *reinterpret_cast<f32x2_t*>(a.row_stats_out + ((size_t)(nslot / LN_SLOT_COLS) * a.M + m) * 2) = f32x2_t{sm, q};
            }
        }
    }
    if (staged) {
        __syncthreads();
        if ((FX & FX_VT) != 0 && vt_tile) StageT::store(smem, a, m0, n0);
        else Stage::store(smem, a.C, a.ldc, m0, GLU ? n0 >> 1 : n0, a.M);
    }
    if (ext) {
        const int m = m0 + 256 + l15;
        if (m < a.M) {
            QuadIn qx[4];
            if (nsl == 1) {                            // (branch-free, as above; remainder rows: no fused LayerNorm, FX == 0 or the RoPE form)
                const bool al = EPI != EPI_RESID || ((a.ldc | a.ldres) & 3) == 0;
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    const int nb = n0 + xcol0 + 16 * h + 4 * q4;
                    const int nc = (nb + 3 < a.N && al) ? nb : 0;
                    if (a.bias) qx[h].bias = *reinterpret_cast<const u32x2*>(a.bias + nc);
                    if (a.bias2) qx[h].bias2 = *reinterpret_cast<const u32x2*>(a.bias2 + (size_t)(m / a.rows_per_batch) * a.ld_bias2 + nc);
                    if constexpr (EPI == EPI_RESID) qx[h].res = *reinterpret_cast<const u32x2*>(a.res + (size_t)m * a.ldres + nc);
                }
            }
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const int nb = n0 + xcol0 + 16 * h + 4 * q4;
                if (nb >= a.N) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = accx[h][e];
                RowFx fx;                              // remainder rows never carry the fused-LayerNorm features (gemm256_ok)
                emit(m, nb, v, fx, qx[h]);
            }
        }
    }
}

template <int EPI, bool CONV>
int launch_w4_epi(const GemmArgs& b, hipStream_t s, int grid, int fx) {
    if (fx & FX_ROPE) {
        if constexpr (!CONV && EPI == EPI_NONE) {
            hipLaunchKernelGGL((gemm_w4_kernel<EPI_NONE, false, FX_ROPE | FX_VT>), dim3(grid), dim3(256), 0, s, b);
            return 0;
        }
        return -22;
    }
    if (fx) {
        if constexpr (!CONV) {
            const bool ok = gemm_fx_dispatch<EPI>(fx, [&](auto m) {
                constexpr int FXM = decltype(m)::value;
                if constexpr ((FXM & FX_CROSS) == 0) hipLaunchKernelGGL((gemm_w4_kernel<EPI, CONV, FXM>), dim3(grid), dim3(256), 0, s, b);
            });
            return ok && !(fx & FX_CROSS) ? 0 : -22;
        }
        return -22;
    }
    hipLaunchKernelGGL((gemm_w4_kernel<EPI, CONV>), dim3(grid), dim3(256), 0, s, b);
    return 0;
}

}  // namespace

// The main launch of a 256 x 256 run on the four-wave tile.  `b` is the fully prepared argument block of gemm256.hip::launch_pp
// (full_tiles / ksplit / stage flags / tile blocks set); the caller launches the K-slice reduce that follows.  -22: not instantiated.
int launch_gemm_w4(const GemmArgs& b, hipStream_t s, int grid, int fx) {
    const bool conv = b.conv.mode != CONV_NONE;
    if (conv) {
        switch (b.epi) {
            case EPI_NONE:  return launch_w4_epi<EPI_NONE, true>(b, s, grid, fx);
            case EPI_RESID: return launch_w4_epi<EPI_RESID, true>(b, s, grid, fx);
            default: return -22;
        }
    }
    switch (b.epi) {
        case EPI_NONE:   return launch_w4_epi<EPI_NONE, false>(b, s, grid, fx);
        case EPI_RESID:  return launch_w4_epi<EPI_RESID, false>(b, s, grid, fx);
        case EPI_SWIGLU: return launch_w4_epi<EPI_SWIGLU, false>(b, s, grid, fx);
        case EPI_SILU:   return launch_w4_epi<EPI_SILU, false>(b, s, grid, fx);
        case EPI_GELU:   return launch_w4_epi<EPI_GELU, false>(b, s, grid, fx);
        case EPI_GEGLU:  return launch_w4_epi<EPI_GEGLU, false>(b, s, grid, fx);
        default: return -22;
    }
}
