// Device code shared by the MFMA GEMM kernels (gemm.hip: 128x128 / 256x128 / 128x64 tiles; gemm256.hip: the 256x256
// ping-pong tile): the fused epilogue of one accumulator quad, LDS-DMA and counted-wait helpers, the implicit-GEMM 3x3
// gather and the split-K reduce launch.
#pragma once
#include <type_traits>

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

// Per-workgroup timeline of a GEMM kernel (only in a library built with -DEMU_TRACE: `python -m emu_amd.build --trace`,
// tools/gemm_trace.py).  Record = 8 x u64 at trace[blockIdx.x * 8]: [0] entry, [1] first k tile landed (after the first
// barrier of the main loop), [2] main loop done, [3] epilogue stores complete -- all s_memrealtime (100 MHz, chip-wide) --,
// [4] XCC_ID | HW_ID << 8, [5] / [6] s_memtime (shader clock) at entry / loop end, [7] logical tile.
#ifdef EMU_TRACE
__device__ __forceinline__ void trace_mark(unsigned long long* t, int slot) {
    if (t && threadIdx.x == 0) {
        t[(size_t)blockIdx.x * 8 + slot] = wall_clock64();
        if (slot == 0) {
            t[(size_t)blockIdx.x * 8 + 4] = (unsigned long long)__builtin_amdgcn_s_getreg(6164) |      // XCC_ID[3:0]
                                            ((unsigned long long)__builtin_amdgcn_s_getreg(63492) << 8);   // HW_ID
            t[(size_t)blockIdx.x * 8 + 5] = clock64();
        }
        if (slot == 2) t[(size_t)blockIdx.x * 8 + 6] = clock64();
    }
}
#define EMU_TRACE_MARK(t, slot) trace_mark((t), (slot))
#else
#define EMU_TRACE_MARK(t, slot) do {} while (0)
#endif

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

// Fused-LayerNorm state of one output row inside an epilogue (GemmArgs::ln_* / row_stats_out): the statistics of row m
// of A for the consumer-side correction, and the running (sum, sum of squares) of the bf16 outputs this lane has written
// into the current 64-column slot for the producer side.
// Compile-time feature mask of the fused epilogues (template parameter FX of the kernels and of store_quad): every
// combination the UNet launches has its own instantiation, so a producer that only emits row statistics carries none of the
// consumer's registers (its 20 statistics pairs, the fp32 ln_c / ln_d quads) and vice versa.
constexpr int FX_LN = 1;            // GemmArgs::ln_*: LayerNorm of A folded into this GEMM
constexpr int FX_STATS = 2;         // GemmArgs::row_stats_out: emit per-row partial sums of C
// successor prefetch (GemmArgs::pf_*): this thread's (at most two) lines, requested and never waited for; the value returned keeps
// the destination register allocated until the caller has passed its first counted wait (loads return in order: the register is
// written for the last time before any later load of the wave lands)
__device__ __forceinline__ uint32_t prefetch_lines(const void* p, size_t bytes, uint32_t gid, uint32_t total) {
    uint32_t d = 0;
    if (p) {
        const size_t lines = bytes >> 7;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const size_t l = (size_t)gid + (size_t)i * total;
            if (l < lines) {
                // "+v": the destination stays LIVE between the two requests -- as a plain output the compiler reuses the register of a
                // load still in flight as scratch for the second address, and the late dword then lands in an address (a wild
                // global_load, MEMORY_APERTURE_VIOLATION: measured) -- and, being an input too, can never share the address pair
                asm volatile("global_load_dword %0, %1, off" : "+v"(d) : "v"(reinterpret_cast<const char*>(p) + (l << 7)) : "memory");
            }
        }
    }
    return d;
}
__device__ __forceinline__ void prefetch_release(uint32_t d) { asm volatile("" :: "v"(d)); }

constexpr int FX_VT = 4;            // GemmArgs::vt_out: V heads stored key-contiguous
constexpr int FX_CROSS = 8;         // GemmArgs::cross_*: cross-attention over <= 64 cached keys in the epilogue (128 x 64 tile)
constexpr int FX_ROPE = 16;         // GemmArgs::rope_*: RoPE + KV append (+ FX_VT) in the epilogue of the LLaMA prefill's qkv projection (256x256 tile)

struct RowFx {
    float mean = 0.f, rstd = 1.f;
    float rs = 0.f, rq = 0.f;
};

// mean / rstd of NR rows of A from the producer's per-slot partial sums, in two halves so the loads can be issued long
// before they are needed: ln_rows_load fetches ALL slots of the rows unconditionally (slot index clamped: no branch around a
// load, every lane the same instruction stream), ln_rows_finish sums them in slot order (deterministic).  The lock-step
// kernels issue the loads at the top of the kernel and finish in the epilogue: hipcc inserts its wait at the first USE of a
// loaded value, so the main loop (whose only vector-memory waits are our counted asm ones; older loads complete first) never
// stalls on them and the L2 round trip costs nothing.  (A plain `for slot: s += p[slot]` loop made every consumer GEMM 4-6 us
// slower: hipcc waits for each load before issuing the next.)  Rows past M are clamped by the caller.
constexpr int LN_SLOT_COLS = 128;                      // columns per row-statistics slot
constexpr int LN_MAX_SLOTS = 10;                       // K / 128 for K <= 1280 (the widest UNet level)
template <int NR>
struct LnRaw { f32x2_t v[NR][LN_MAX_SLOTS]; };

template <int NR>
__device__ __forceinline__ void ln_rows_load(const GemmArgs& a, const int (&m)[NR], LnRaw<NR>& raw) {
    const f32x2_t* p = reinterpret_cast<const f32x2_t*>(a.ln_stats);
    const int last = a.ln_slots - 1;
#pragma unroll
    for (int t = 0; t < LN_MAX_SLOTS; ++t) {
        const int tt = t < last ? t : last;
#pragma unroll
        for (int r = 0; r < NR; ++r) raw.v[r][t] = p[(size_t)tt * a.M + m[r]];
    }
}
template <int NR>
__device__ __forceinline__ void ln_rows_finish(const GemmArgs& a, const LnRaw<NR>& raw, RowFx (&fx)[NR]) {
    const float inv = 1.0f / (float)a.K;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int t = 0; t < LN_MAX_SLOTS; ++t) {
            const bool in = t < a.ln_slots;
            s += in ? raw.v[r][t][0] : 0.f;
            q += in ? raw.v[r][t][1] : 0.f;
        }
        fx[r].mean = s * inv;
        const float var = fmaxf(q * inv - fx[r].mean * fx[r].mean, 0.f);
        fx[r].rstd = rsqrtf(var + a.ln_eps);
    }
}
__device__ __forceinline__ void ln_row_stats(const GemmArgs& a, int m, RowFx& fx) {
    const int mm[1] = {m};
    LnRaw<1> raw;
    RowFx f[1];
    ln_rows_load<1>(a, mm, raw);
    ln_rows_finish<1>(a, raw, f);
    fx.mean = f[0].mean; fx.rstd = f[0].rstd;
}

// Memory operands of one accumulator quad.  They are fetched by quad_load_cols / quad_load_row for ALL quads of a lane
// before the first store of the epilogue: written as `load bias; compute; store; load next bias; ...` hipcc must assume a
// store may alias the next load and waits for every load on its own -- 8 to 32 serialised L2 round trips per lane in every
// epilogue with a bias or a residual (measured: +2.8 us on the 128x64 tile, ~10 us on the 256x256 tile).
struct QuadIn {
    u32x2 bias = {0u, 0u};                             // a.bias[nb .. nb+3]
    u32x2 bias2 = {0u, 0u};                            // a.bias2[batch of m][nb .. nb+3]
    u32x2 res = {0u, 0u};                              // a.res[m][nb .. nb+3]
    f32x4_t c = {0.f, 0.f, 0.f, 0.f}, d = {0.f, 0.f, 0.f, 0.f};      // fused LayerNorm: ln_c / ln_d [nb .. nb+3]
};

// 8-byte accesses need every row start 8-byte aligned: ldc (and ldres) % 4 == 0; other strides (a [M, 32274] logits buffer)
// take the scalar path of store_quad (GLU epilogues store one 4-byte pair per quad: an even ldc is enough)
template <int EPI>
__device__ __forceinline__ bool quad_full(const GemmArgs& a, int nb) {
    constexpr bool GLU = EPI == EPI_SWIGLU || EPI == EPI_GEGLU;
    return (nb + 3) < a.N && (GLU || ((a.ldc | (EPI == EPI_RESID ? a.ldres : 0)) & 3) == 0);
}
// operands that depend on the column only: one fetch serves every row of the lane
template <int EPI, int FX>
__device__ __forceinline__ void quad_load_cols(const GemmArgs& a, int nb, QuadIn& q) {
    if (!quad_full<EPI>(a, nb)) return;
    if constexpr ((FX & FX_LN) != 0) {
        q.c = *reinterpret_cast<const f32x4_t*>(a.ln_c + nb);
        q.d = *reinterpret_cast<const f32x4_t*>(a.ln_d + nb);
    } else if (a.bias) {
        q.bias = *reinterpret_cast<const u32x2*>(a.bias + nb);
    }
}
// operands of row m
template <int EPI>
__device__ __forceinline__ void quad_load_row(const GemmArgs& a, int m, int nb, QuadIn& q) {
    if (!quad_full<EPI>(a, nb)) return;
    if (a.bias2) q.bias2 = *reinterpret_cast<const u32x2*>(a.bias2 + (size_t)(m / a.rows_per_batch) * a.ld_bias2 + nb);
    if constexpr (EPI == EPI_RESID) q.res = *reinterpret_cast<const u32x2*>(a.res + (size_t)m * a.ldres + nb);
}

// The value part of the epilogue of one FULL accumulator quad (quad_full): bias / fused-LayerNorm correction / per-batch bias /
// activation / residual with the reference's bf16 rounding points, packed to bf16.  Non-GLU: 4 columns in (x, y); GLU: the 2
// output columns in x.  Row statistics (FX_STATS) accumulate in fx.  res_in = the quad's residual values (EPI_RESID).
template <int EPI, int FX>
__device__ __forceinline__ u32x2 quad_value(const GemmArgs& a, float (&v)[4], RowFx& fx, const QuadIn& q) {
    if constexpr ((FX & FX_LN) != 0) {             // LayerNorm folded into this GEMM (launch_gemm: N % 4 == 0, no bias)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaf(fx.rstd, v[e] - fx.mean * q.c[e], q.d[e]);
    } else if (a.bias) {
        v[0] += bflo(q.bias.x); v[1] += bfhi(q.bias.x); v[2] += bflo(q.bias.y); v[3] += bfhi(q.bias.y);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = bfround(v[e]);
    if (a.bias2) {
        v[0] = bfround(v[0] + bflo(q.bias2.x)); v[1] = bfround(v[1] + bfhi(q.bias2.x));
        v[2] = bfround(v[2] + bflo(q.bias2.y)); v[3] = bfround(v[3] + bfhi(q.bias2.y));
    }
    u32x2 ov{0u, 0u};
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
        ov.x = packbf(o0, o1);
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
            v[0] += bflo(q.res.x); v[1] += bfhi(q.res.x); v[2] += bflo(q.res.y); v[3] += bfhi(q.res.y);
        }
        ov.x = packbf(v[0], v[1]);
        ov.y = packbf(v[2], v[3]);
        if constexpr ((FX & FX_STATS) != 0) {      // statistics of what the next LayerNorm will read: the bf16 values
            const float r0 = bflo(ov.x), r1 = bfhi(ov.x), r2 = bflo(ov.y), r3 = bfhi(ov.y);
            fx.rs += (r0 + r1) + (r2 + r3);
            fx.rq += fmaf(r0, r0, r1 * r1) + fmaf(r2, r2, r3 * r3);
        }
    }
    return ov;
}

// Epilogue for one accumulator quad: lane-local 4 consecutive output columns nb..nb+3 of row m; q = its memory operands.
// FX = 0 compiles every fused-LayerNorm / V^T feature out (the instantiations all other callers use are unchanged).
template <int EPI, int FX = 0>
__device__ __forceinline__ void store_quad(const GemmArgs& a, int m, int nb, float (&v)[4], RowFx& fx, const QuadIn& q) {
    if (quad_full<EPI>(a, nb)) {
        const u32x2 ov = quad_value<EPI, FX>(a, v, fx, q);
        if constexpr (EPI == EPI_SWIGLU || EPI == EPI_GEGLU) {
            *reinterpret_cast<uint32_t*>(a.C + (size_t)m * a.ldc + (nb >> 1)) = ov.x;
        } else {
            if ((FX & FX_VT) != 0 && nb >= a.vt_col0) {   // V heads: key-contiguous store for the P.V MFMA (wave-uniform branch)
                const int b = m / a.vt_s, sidx = m - b * a.vt_s;
                bf16_t* dst = a.vt_out + ((size_t)b * (a.N - a.vt_col0) + (nb - a.vt_col0)) * a.vt_spad + sidx;
                dst[0] = (bf16_t)(ov.x & 0xffffu);
                dst[a.vt_spad] = (bf16_t)(ov.x >> 16);
                dst[2 * (size_t)a.vt_spad] = (bf16_t)(ov.y & 0xffffu);
                dst[3 * (size_t)a.vt_spad] = (bf16_t)(ov.y >> 16);
            } else {
                *reinterpret_cast<u32x2*>(a.C + (size_t)m * a.ldc + nb) = ov;
            }
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

// ---- Staged epilogue.  In the accumulator layout a lane owns 4 consecutive columns of ONE row, so a direct store instruction
// writes 8 bytes to each of 32 different rows: 32 cache lines touched per instruction, and the residual loads likewise.  Traced
// per workgroup (profiles/r04_gemm_trace_baseline_*.log) that tail is 4-6 us on the 128 x 64 tile and 10-19 us on the 256 x 256
// tile -- a third to a half of a K = 640 / 1280 GEMM.  Here the workgroup's tile of results goes through LDS (the k-tile ring is
// dead by then): the residual tile comes in by LDS-DMA, 16 bytes per lane and row-contiguous; every lane reads its quads' residual
// from LDS, computes and writes the packed results back IN PLACE (each element has exactly one owner lane); then the tile leaves
// with 16-byte stores, 64 lanes covering whole rows.  16-byte slots of a row are XOR-swizzled with the row so that the 8-byte
// quad accesses of 16 consecutive rows hit distinct banks; the same XOR is applied to the global column of the DMA source / the
// store, which only permutes 16-byte pieces inside a row.
// BMv rows x NOUT output columns (bf16) per tile; the tile must lie inside N (rows beyond M are clamped on the way in and skipped
// on the way out).
template <int BMv, int NOUT, int THREADS>
struct EpiStage {
    static constexpr int ROWB = NOUT * 2, SLOTS = ROWB / 16, KEYM = SLOTS >= 16 ? 15 : SLOTS - 1;
    static constexpr int BYTES = BMv * ROWB;
    static constexpr int ROUNDS = (BMv * SLOTS) / THREADS;
    static_assert((BMv * SLOTS) % THREADS == 0 && (SLOTS & (SLOTS - 1)) == 0, "tile splits into whole rounds of 16-byte pieces");
    // byte offset of output column col (element index inside the tile row; the access must not cross a 16-byte slot)
    __device__ static __forceinline__ int off(int row, int col) {
        return row * ROWB + ((((col >> 3) ^ row) & KEYM) << 4 | ((col >> 3) & ~KEYM) << 4) + ((col & 7) << 1);
    }
    // residual tile [m0 .. m0 + BMv) x [c0 .. c0 + NOUT) -> LDS (LDS-DMA; caller waits vmcnt(0) + barrier)
    __device__ static __forceinline__ void load(char* lds, const bf16_t* src, int ld, int m0, int c0, int M) {
        const int tid = threadIdx.x;
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const int idx = r * THREADS + tid, row = idx / SLOTS, ps = idx % SLOTS;
            const int ls = (ps & ~KEYM) | ((ps ^ row) & KEYM);           // logical slot held at physical slot ps of this row
            int m = m0 + row; m = m < M ? m : M - 1;
            glds16(src + (size_t)m * ld + c0 + ls * 8, lds + (r * THREADS + wave * 64) * 16);
        }
    }
    // LDS tile -> C, 16 bytes per lane (after a barrier that follows the last in-place write)
    __device__ static __forceinline__ void store(const char* lds, bf16_t* dst, int ld, int m0, int c0, int M) {
        const int tid = threadIdx.x;
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const int idx = r * THREADS + tid, row = idx / SLOTS, ps = idx % SLOTS;
            const int ls = (ps & ~KEYM) | ((ps ^ row) & KEYM);
            const u32x4 v = *reinterpret_cast<const u32x4*>(lds + idx * 16);
            if (m0 + row < M) *reinterpret_cast<u32x4*>(dst + (size_t)(m0 + row) * ld + c0 + ls * 8) = v;
        }
    }
};
// The transposed twin for the V^T tiles of a fused qkv projection (GemmArgs::vt_out: output column n of row m = b * vt_s + s lands
// at vt_out[(b * (N - vt_col0) + n - vt_col0) * vt_spad + s], key-contiguous): the tile sits in LDS as [n][m], every lane drops its
// 4 values as 2-byte writes (32 lanes = 32 consecutive m of one n: 64 contiguous bytes), and it leaves as 16-byte stores of 8
// consecutive keys -- against 64 two-byte global stores per lane straight from the accumulators (traced: those tiles' epilogue
// 6.7-10 us where the others take 3-4).  Needs whole tiles inside one batch element (vt_s % BMv == 0) or a single batch element.
template <int BMv, int BNv, int THREADS>
struct EpiStageT {
    static constexpr int ROWB = BMv * 2, SLOTS = ROWB / 16, KEYM = SLOTS >= 16 ? 15 : SLOTS - 1;
    static constexpr int ROUNDS = (BNv * SLOTS) / THREADS;
    static_assert((BNv * SLOTS) % THREADS == 0 && (SLOTS & (SLOTS - 1)) == 0, "tile splits into whole rounds of 16-byte pieces");
    // byte offset of (column n of the tile, row r of the tile)
    __device__ static __forceinline__ int off(int n, int r) {
        const int slot = r >> 3;
        return n * ROWB + (((slot & ~KEYM) | ((slot ^ n) & KEYM)) << 4) + ((r & 7) << 1);
    }
    __device__ static __forceinline__ void store(const char* lds, const GemmArgs& a, int m0, int n0) {
        const int tid = threadIdx.x;
        const int b = m0 / a.vt_s, s0 = m0 - b * a.vt_s;
        bf16_t* base = a.vt_out + ((size_t)b * (a.N - a.vt_col0) + (n0 - a.vt_col0)) * a.vt_spad + s0;
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const int idx = r * THREADS + tid, n = idx / SLOTS, ps = idx % SLOTS;
            const int ls = (ps & ~KEYM) | ((ps ^ n) & KEYM);
            const u32x4 v = *reinterpret_cast<const u32x4*>(lds + idx * 16);
            if (m0 + ls * 8 < a.M && n0 + n < a.N) *reinterpret_cast<u32x4*>(base + (size_t)n * a.vt_spad + ls * 8) = v;
        }
    }
};
// may the V^T tiles of this launch take the transposed staging? (host side)
inline bool stage_vt_ok(const GemmArgs& a, int bm, int bn) {
    // whole tiles inside one batch element -- or ONE batch element (vt_s == M: the ViT's 1025 tokens), whose ragged last tile
    // writes a few pad keys (< 8, inside vt_spad) along with its last 8-key group
    const bool one = a.vt_s == a.M && a.vt_spad >= (a.M + 7) / 8 * 8;
    return a.vt_out && (one || (a.vt_s % bm == 0 && a.M % bm == 0)) && (a.vt_col0 % bn) == 0 && !(a.vt_spad & 7) && !((uintptr_t)a.vt_out & 15);
}

// host side: may this launch take the staged epilogue?  16-byte alignment of C (and the residual) rows; the kernels add the
// per-tile conditions (tile inside N, whole-K tile, no V^T columns, no remainder-row accumulator)
inline bool stage_ok(const GemmArgs& a) {
    const bool glu = a.epi == EPI_SWIGLU || a.epi == EPI_GEGLU;
    if ((a.ldc & 7) || ((uintptr_t)a.C & 15) || (a.N & (glu ? 15 : 7))) return false;
    if (a.epi == EPI_RESID && ((a.ldres & 7) || ((uintptr_t)a.res & 15))) return false;
    return true;
}

// Reduce kernels: consecutive lanes walk consecutive quads of a row, so an aligned 32-lane half-wave holds 128 columns = one
// statistics slot of one row: 16-lane DPP sums, then the two 16-lane groups meet.  Every lane of the wave must call this.
__device__ __forceinline__ void emit_row_stats32(const GemmArgs& a, int m, int nb, bool ok, RowFx& fx) {
    float s = row16_sum(ok ? fx.rs : 0.f), q = row16_sum(ok ? fx.rq : 0.f);
    s += __shfl_xor(s, 16, 64);
    q += __shfl_xor(q, 16, 64);
    if (ok && (threadIdx.x & 31) == 0)
        *reinterpret_cast<f32x2_t*>(a.row_stats_out + ((size_t)(nb / LN_SLOT_COLS) * a.M + m) * 2) = f32x2_t{s, q};
}

// second launch of a split-K GEMM: sum the K-slices of every sliced tile in slice order (deterministic) and apply the
// fused epilogue.  SPLITK_RED_Y workgroups per tile (a handful of tiles must still fill the chip).
constexpr int SPLITK_RED_Y = 16;
template <int EPI, int BMv, int BNv, int FX = 0>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmArgs a) {
    const int wg = a.full_tiles + blockIdx.x;
    const int tiles_m = (a.M + BMv - 1) / BMv;
    const int n0 = (wg / tiles_m) * BNv, m0 = (wg % tiles_m) * BMv;
    const float* base = a.partial + (size_t)blockIdx.x * a.ksplit * (BMv * BNv);
    constexpr int QN = BNv / 4;
    constexpr int PER = BMv * QN / SPLITK_RED_Y;                      // quads per workgroup (grid.y chunks of a tile)
    static_assert(PER % 64 == 0 && QN % 32 == 0, "whole waves per iteration, 32-lane halves inside one 128-column slot");
    for (int q = blockIdx.y * PER + threadIdx.x; q < (blockIdx.y + 1) * PER; q += 256) {
        const int lm = q / QN, lq = q - lm * QN;
        const int m = m0 + lm, nb = n0 + lq * 4;
        const bool ok = m < a.M && nb < a.N;
        RowFx fx;
        if (ok) {
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            for (int ks = 0; ks < a.ksplit; ++ks) {
                const f32x4_t t = *reinterpret_cast<const f32x4_t*>(base + (size_t)ks * (BMv * BNv) + (size_t)lm * BNv + lq * 4);
                v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3];
            }
            if (a.a_scale) {                           // fp8 operands: the per-row scales of both multiply the summed slices
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

// Second launch of a run whose EVERY tile was K-sliced into row-major slices (GemmArgs::slab_rows): one workgroup per output
// row sums the slices in slice order, applies the EPI_NONE / EPI_RESID epilogue (bias, bf16 rounding, residual: the reduce
// kernels' rounding points), keeps the row's bf16 values in registers and writes the norm of the row: RMSNorm with
// rmsnorm_kernel's arithmetic (norm_b null), or LayerNorm (+ residual) with layernorm_kernel's -- same summation order (thread t
// owns the 16-byte vectors t, t + 256, ...), so C and norm_out are bit-identical to the two launches this replaces (LLaMA
// prefill: o_proj / down_proj + RMSNorm; ViT: fc2 + LayerNorm + residual).
template <int NV>
__global__ __launch_bounds__(256) void rows_reduce_norm_kernel(const GemmArgs a) {
    __shared__ float scratch[4];
    const int m = blockIdx.x, nv = a.N >> 3, tid = threadIdx.x;
    float f[NV][8];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int vi = tid + i * 256;
#pragma unroll
        for (int j = 0; j < 8; ++j) f[i][j] = 0.f;
        if (vi < nv) {
            float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int ks = 0; ks < a.ksplit; ++ks) {
                const float* src = a.partial + ((size_t)ks * a.M + m) * a.N + vi * 8;
                const f32x4_t t0 = *reinterpret_cast<const f32x4_t*>(src), t1 = *reinterpret_cast<const f32x4_t*>(src + 4);
                v[0] += t0[0]; v[1] += t0[1]; v[2] += t0[2]; v[3] += t0[3];
                v[4] += t1[0]; v[5] += t1[1]; v[6] += t1[2]; v[7] += t1[3];
            }
            if (a.bias) {
                float bb[8];
                unpack8(ld16(a.bias + vi * 8), bb);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] += bb[j];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = bfround(v[j]);
            if (a.epi == EPI_RESID) {
                float r[8];
                unpack8(ld16(a.res + (size_t)m * a.ldres + vi * 8), r);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] += r[j];
            }
            const u32x4 hv = pack8(v);
            if (a.C) st16(a.C + (size_t)m * a.ldc + vi * 8, hv);
            unpack8(hv, f[i]);
            if (a.norm_b) {
#pragma unroll
                for (int j = 0; j < 8; ++j) ss += f[i][j];
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) ss += f[i][j] * f[i][j];
            }
        }
    }
    if (!a.norm_b) {                                   // RMSNorm (rmsnorm_kernel)
        const float rinv = rsqrtf(block_sum<4>(ss, scratch) / (float)a.N + a.norm_eps);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int vi = tid + i * 256;
            if (vi < nv) {
                float g[8], o[8];
                unpack8(ld16(a.norm_w + vi * 8), g);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = g[j] * bfround(f[i][j] * rinv);
                st16(a.norm_out + (size_t)m * a.norm_ld + vi * 8, pack8(o));
            }
        }
        return;
    }
    // LayerNorm (+ residual) (layernorm_kernel)
    const float mean = block_sum<4>(ss, scratch) / (float)a.N;
    float var = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int vi = tid + i * 256;
        if (vi < nv) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = f[i][j] - mean; var += d * d; }
        }
    }
    const float rstd = rsqrtf(block_sum<4>(var, scratch) / (float)a.N + a.norm_eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int vi = tid + i * 256;
        if (vi < nv) {
            float g[8], bb[8], o[8];
            unpack8(ld16(a.norm_w + vi * 8), g);
            unpack8(ld16(a.norm_b + vi * 8), bb);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (f[i][j] - mean) * rstd * g[j] + bb[j];
            if (a.norm_res) {
                float r[8];
                unpack8(ld16(a.norm_res + (size_t)m * a.norm_ldres + vi * 8), r);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = r[j] + bfround(o[j]);
            }
            st16(a.norm_out + (size_t)m * a.norm_ld + vi * 8, pack8(o));
        }
    }
}
inline void launch_rows_reduce_norm(const GemmArgs& b, hipStream_t s) {
    if (b.N <= 2048) hipLaunchKernelGGL((rows_reduce_norm_kernel<1>), dim3(b.M), dim3(256), 0, s, b);
    else if (b.N <= 8192) hipLaunchKernelGGL((rows_reduce_norm_kernel<4>), dim3(b.M), dim3(256), 0, s, b);
    else hipLaunchKernelGGL((rows_reduce_norm_kernel<8>), dim3(b.M), dim3(256), 0, s, b);
}

// the feature mask of a launch (host side: picks the FX instantiation), and the (epilogue, mask) pairs that are instantiated:
// what the UNet transformer blocks launch -- proj_in / to_q producers and consumers, the qkv projection (+ V^T), the residual
// out-projections and ff-out as producers, the GEGLU projection as a consumer
inline int gemm_fx(const GemmArgs& a) {
    return (a.ln_c ? FX_LN : 0) | (a.row_stats_out ? FX_STATS : 0) | (a.vt_out ? FX_VT : 0) | (a.cross_k ? FX_CROSS : 0) | (a.rope_cos ? FX_ROPE : 0);
}
constexpr bool gemm_fx_ok(int epi, int fx) {
    return fx == 0 || (epi == EPI_NONE && (fx == FX_LN || fx == FX_STATS || fx == FX_VT || fx == (FX_LN | FX_VT) || fx == FX_CROSS ||
                                           fx == (FX_LN | FX_CROSS) || fx == (FX_ROPE | FX_VT))) ||
           (epi == EPI_RESID && fx == FX_STATS) || (epi == EPI_GEGLU && fx == FX_LN);
}
// calls f(std::integral_constant<int, FX>) for the instantiated mask of this epilogue; false when the pair does not exist
template <int EPI, class F>
inline bool gemm_fx_dispatch(int fx, F&& f) {
    if constexpr (EPI == EPI_NONE) {
        switch (fx) {
            case FX_LN: f(std::integral_constant<int, FX_LN>{}); return true;
            case FX_STATS: f(std::integral_constant<int, FX_STATS>{}); return true;
            case FX_VT: f(std::integral_constant<int, FX_VT>{}); return true;
            case FX_LN | FX_VT: f(std::integral_constant<int, FX_LN | FX_VT>{}); return true;
            case FX_CROSS: f(std::integral_constant<int, FX_CROSS>{}); return true;
            case FX_LN | FX_CROSS: f(std::integral_constant<int, FX_LN | FX_CROSS>{}); return true;
        }
    } else if constexpr (EPI == EPI_RESID) {
        if (fx == FX_STATS) { f(std::integral_constant<int, FX_STATS>{}); return true; }
    } else if constexpr (EPI == EPI_GEGLU) {
        if (fx == FX_LN) { f(std::integral_constant<int, FX_LN>{}); return true; }
    }
    return false;
}

}  // namespace emu_gemm
