// Common device helpers for the Emu2 MI355X (gfx950 / CDNA4) kernels.
// wave = 64 lanes; MFMA fragments follow the gfx950 32x32x16 bf16 layout:
//   A/B operand: lane l holds row/col (l & 31), k = 8*(l >> 5) + j, j = 0..7  (8 bf16 = 4 VGPRs)
//   C/D        : lane l holds col (l & 31), row = (r & 3) + 8*(r >> 2) + 4*(l >> 5), r = 0..15
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;                                             // raw bfloat16 bits
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));         // one MFMA A/B fragment
typedef float f32x16_t __attribute__((ext_vector_type(16)));         // one 32x32 accumulator
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));           // 16-byte load/store unit (8 bf16)
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

#define EMU_WAVE 64

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// float -> bfloat16, round-to-nearest-even: gfx950 has the conversion in hardware (v_cvt_pk_bf16_f32, one VALU op for
// two values) -- same results as torch's cast for every finite input (NaNs come back quiet)
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ float bfround(float f) { return bf2f(f2bf(f)); }

// low / high bf16 of a packed dword as float
__device__ __forceinline__ float bflo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bfhi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ uint32_t packbf(float lo, float hi) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t));
}

__device__ __forceinline__ void unpack8(const u32x4& v, float* f) {
    f[0] = bflo(v.x); f[1] = bfhi(v.x); f[2] = bflo(v.y); f[3] = bfhi(v.y);
    f[4] = bflo(v.z); f[5] = bfhi(v.z); f[6] = bflo(v.w); f[7] = bfhi(v.w);
}
__device__ __forceinline__ u32x4 pack8(const float* f) {
    u32x4 v;
    v.x = packbf(f[0], f[1]); v.y = packbf(f[2], f[3]); v.z = packbf(f[4], f[5]); v.w = packbf(f[6], f[7]);
    return v;
}

// Cross-lane reductions on the DPP path (one VALU op per step) instead of ds_bpermute round trips through the LDS
// crossbar.  quad_perm xor-1 / xor-2, then row_half_mirror and row_mirror (each half / row is uniform by then), leave the
// sum of every aligned 16-lane row in all of its lanes; the four rows are combined through v_readlane (uniform).
// All 64 lanes must be active at the call.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float quad_sum(float v) {
    v += dpp_mov<0xB1>(v);                         // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);                         // quad_perm [2,3,0,1]
    return v;
}
__device__ __forceinline__ float row8_sum(float v) { v = quad_sum(v); v += dpp_mov<0x141>(v); return v; }   // row_half_mirror
__device__ __forceinline__ float row16_sum(float v) { v = row8_sum(v); v += dpp_mov<0x140>(v); return v; }  // row_mirror
__device__ __forceinline__ float quad_max(float v) {
    v = fmaxf(v, dpp_mov<0xB1>(v));
    v = fmaxf(v, dpp_mov<0x4E>(v));
    return v;
}
__device__ __forceinline__ float row16_max(float v) {
    v = quad_max(v);
    v = fmaxf(v, dpp_mov<0x141>(v));
    v = fmaxf(v, dpp_mov<0x140>(v));
    return v;
}
__device__ __forceinline__ float readlane_f(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ float wave_sum(float v) {
    v = row16_sum(v);
    return (readlane_f(v, 0) + readlane_f(v, 16)) + (readlane_f(v, 32) + readlane_f(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
    v = row16_max(v);
    return fmaxf(fmaxf(readlane_f(v, 0), readlane_f(v, 16)), fmaxf(readlane_f(v, 32), readlane_f(v, 48)));
}

// block-wide sum for blocks of NW waves; scratch must hold NW floats; all threads get the result
template <int NW>
__device__ __forceinline__ float block_sum(float v, float* scratch) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __syncthreads();
    if (l == 0) scratch[w] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NW; ++i) t += scratch[i];
    return t;
}

// acc + w.lo * x.lo + w.hi * x.hi over packed bf16 pairs (v_dot2c_f32_bf16, fp32 accumulate)
__device__ __forceinline__ float bf16_dot2(unsigned int w, unsigned int x, float acc) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, w), __builtin_bit_cast(bf16x2_t, x), acc, false);
}

// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, far inside the bf16 rounding that follows every use here):
// 1 rcp + 1 exp2 + a 5-term Horner instead of libm's ~40-instruction erff -- the GELU / GEGLU epilogues are otherwise a
// visible share of their GEMMs (one erf per output element).
__device__ __forceinline__ float erf_as(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    p *= t;
    const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);
    return copysignf(1.0f - p * e, x);
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752440f)); }
// x * sigmoid(x) on the hardware exp2 / rcp (1 ulp each; every use is followed by a bf16 rounding)
__device__ __forceinline__ float silu(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}

// streaming (read-once) 16-byte load: weights are touched once per token, keep them out of L2
__device__ __forceinline__ u32x4 ld_stream(const u32x4* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ u32x4 ld16(const void* p) { return *reinterpret_cast<const u32x4*>(p); }
__device__ __forceinline__ void st16(void* p, const u32x4& v) { *reinterpret_cast<u32x4*>(p) = v; }

// ---- successor prefetch for the one-row decode path: the weights of the launches that FOLLOW this one on the stream
// (a decode step knows them).  A span is two runs of 128-byte lines (a run may end at a matrix boundary); worker w of nworkers
// requests lines w, w + nworkers, ... with one default-policy dword load each -- the line travels HBM -> infinity cache (and the
// requesting XCD's L2) -- and never waits for them: the results are discarded, s_endpgm waits for the counters anyway.  The
// destination is an in-out operand so it stays allocated while loads are in flight (gemm_tile.h::prefetch_lines has the story).
struct PfSpan {
    const char* p0 = nullptr;
    const char* p1 = nullptr;
    uint32_t n0 = 0, n1 = 0;            // lines of each run
};
__device__ __forceinline__ void pf_touch(const PfSpan& s, uint32_t worker, uint32_t nworkers) {
    uint32_t d = 0;
    for (uint32_t l = worker; l < s.n0; l += nworkers)
        asm volatile("global_load_dword %0, %1, off" : "+v"(d) : "v"(s.p0 + ((size_t)l << 7)) : "memory");
    for (uint32_t l = worker; l < s.n1; l += nworkers)
        asm volatile("global_load_dword %0, %1, off" : "+v"(d) : "v"(s.p1 + ((size_t)l << 7)) : "memory");
    asm volatile("" :: "v"(d));
}

#define EMU_CHECK_LAUNCH() do { hipError_t e__ = hipGetLastError(); if (e__ != hipSuccess) return (int)e__; } while (0)
