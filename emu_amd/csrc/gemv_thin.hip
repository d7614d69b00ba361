// Few-row (2..16) weight streams on the matrix cores:  out[m, n] = epilogue( sum_k x[m, k] * W[n, k] )
//
// The reference's DEFAULT decoding mode is 5-beam search (Emu2/emu/emu.py:163-172 -> lm.generate(num_beams=5)): every
// LlamaDecoderLayer linear then sees 5 rows.  The v_dot2c block kernel (gemv.hip) spends one VALU chain per (weight row,
// activation row) and is ~70 % VALU-busy at 5 rows, VALU-bound at 8; the first MFMA form (gemv_mfma_kernel) feeds the MFMA from
// registers, which forces 64-byte-per-row loads (16 rows x 64 B per instruction) and holds it at 3-4 TB/s.  Here the weights
// take the HBM-friendly route AND reach the MFMA in its own layout:
//   * a workgroup (4 waves) owns RN = 16 / 32 weight rows and walks K in stages of 256 (512 contiguous bytes per row);
//   * a stage -- the RN weight rows plus the MR = 8 / 16 activation rows -- is moved by LDS-DMA (global_load_lds, 16 bytes per
//     lane, one instruction = two rows x 512 B, no VGPRs), NSTG stages deep, counted vmcnt + one raw s_barrier per stage;
//   * the DMA source address carries a 16-byte-slot XOR swizzle (slot ^ (row & 15)) so that the ds_read_b128 of a
//     v_mfma_f32_16x16x32_bf16 fragment (16 rows x the same k chunk) is bank-conflict free;
//   * wave w multiplies k-steps 2w, 2w + 1 of every stage (A = weight rows, B = activation rows, columns >= M repeat row M - 1
//     and are dropped), so the VALU does nothing in the loop; the four partial 16 x 16 products meet in LDS at the end.
// Algorithmic bytes per call = 2*N*K; the activations add MR / RN of that in L2 -> LDS traffic, nothing in HBM.
#include "common.h"
#include "kernels.h"
#include "gemm_tile.h"

using namespace emu_gemm;

namespace {

// one LDS-DMA instruction whose source is a use-once stream (nt: do not keep the weight bytes in L2 / the Infinity Cache)
__device__ __forceinline__ void glds16_nt(const void* src, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 2);
}

// KCH = 16-byte slots per row and stage: 32 (256 k, one DMA instruction = two rows) or 64 (512 k, one instruction = one row)
template <int EPI, int RN, int MR, int NSTG, int KCH>
__global__ __launch_bounds__(256) void thin_stream_kernel(const GemvArgs a) {
    constexpr int KC = KCH * 8, ROWB = KCH * 16, RPI = 64 / KCH;          // k per stage, bytes per LDS row, rows per instruction
    constexpr int ROWS = RN + MR, ST = ROWS * ROWB, LPT = ROWS / (4 * RPI), RG = RN / 16, KSW = KCH / 16;   // k-steps per wave
    static_assert(ROWS % (4 * RPI) == 0, "every wave issues the same number of DMA instructions per stage");
    static_assert((NSTG - 2) * LPT <= 63, "vmcnt field is 6 bits");
    constexpr int RED = 4 * RG * 64 * 16;
    __shared__ __attribute__((aligned(16))) char smem[NSTG * ST > RED ? NSTG * ST : RED];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * RN;

    // DMA sources: instruction q = j * 4 + wave fills LDS rows q * RPI .. (lane / KCH selects), slot p = lane % KCH receives
    // global chunk p ^ (row & 15).  Rows [0, RN): weights (clamped at N - 1); [RN, ROWS): activations (clamped at M - 1).
    const bf16_t* src[LPT];
#pragma unroll
    for (int j = 0; j < LPT; ++j) {
        const int r = RPI * (j * 4 + wave) + lane / KCH;
        const int c = (lane % KCH) ^ (r & 15);
        const bf16_t* row;
        if (r < RN) { const int n = n0 + r; row = a.W + (size_t)(n < a.N ? n : a.N - 1) * a.ldw; }
        else { const int m = r - RN; row = a.x + (size_t)(m < a.M ? m : a.M - 1) * a.ldx; }
        src[j] = row + c * 8;
    }
    const int nk = a.K / KC;
    auto issue = [&](int kt, int stage) {
        kt = kt < nk ? kt : nk - 1;                    // past-the-end stages re-load the last one (uniform vmcnt counts)
        char* base = smem + stage * ST + wave * 1024;
#pragma unroll
        for (int j = 0; j < LPT; ++j) {
            if (RPI * (j * 4 + 3) + RPI - 1 < RN) glds16_nt(src[j] + kt * KC, base + j * 4096);      // weight rows only: stream
            else glds16(src[j] + kt * KC, base + j * 4096);
        }
    };
    f32x4_t acc[RG];
#pragma unroll
    for (int rg = 0; rg < RG; ++rg) acc[rg] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int rb = RN + (i < a.M ? i : a.M - 1);       // this lane's activation row (B operand column i)
#pragma unroll
    for (int t = 0; t < NSTG - 1; ++t) issue(t, t);
    for (int kt = 0; kt < nk; ++kt) {
        wait_vmcnt<(NSTG - 2) * LPT>();                // this wave's share of stage kt has landed
        __builtin_amdgcn_s_barrier();                  // ... and everyone's; everyone is done reading stage kt - 1
        issue(kt + NSTG - 1, (kt + NSTG - 1) % NSTG);
        const char* sb = smem + (kt % NSTG) * ST;
        bf16x8_t bf[KSW], af[KSW][RG];
#pragma unroll
        for (int s = 0; s < KSW; ++s) {
            const int c = (KSW * wave + s) * 4 + g;    // 16-byte chunk of this lane's 8 k values
            bf[s] = *reinterpret_cast<const bf16x8_t*>(sb + rb * ROWB + ((c ^ (rb & 15)) << 4));
#pragma unroll
            for (int rg = 0; rg < RG; ++rg)
                af[s][rg] = *reinterpret_cast<const bf16x8_t*>(sb + (rg * 16 + i) * ROWB + ((c ^ i) << 4));
        }
#pragma unroll
        for (int s = 0; s < KSW; ++s)
#pragma unroll
            for (int rg = 0; rg < RG; ++rg)
                acc[rg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[s][rg], bf[s], acc[rg], 0, 0, 0);
    }
    wait_vmcnt<0>();                                   // drain the tail DMA before the ring is reused
    __syncthreads();
    f32x4_t (*part)[RG][64] = reinterpret_cast<f32x4_t (*)[RG][64]>(smem);
#pragma unroll
    for (int rg = 0; rg < RG; ++rg) part[wave][rg][lane] = acc[rg];
    __syncthreads();
    // wave w finishes row groups w, w + 4, ...: lane holds C[weight row 4g + r][activation row i], r = 0..3
    const int m = i;
    for (int rg = wave; rg < RG; rg += 4) {
        f32x4_t v = part[0][rg][lane];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const f32x4_t t = part[w][rg][lane];
            v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3];
        }
        if (m >= a.M) continue;
        const int nb = n0 + 16 * rg + 4 * g;           // first of this lane's 4 output columns
        if constexpr (EPI == EPI_SWIGLU) {
#pragma unroll
            for (int r = 0; r < 4; r += 2) {
                const int n = nb + r;
                if (n + 1 < a.N) {
                    float gt = v[r], up = v[r + 1];
                    if (a.bias) { gt += bf2f(a.bias[n]); up += bf2f(a.bias[n + 1]); }
                    gt = bfround(gt); up = bfround(up);
                    a.out[(size_t)m * a.ldo + (n >> 1)] = f2bf(bfround(silu(gt)) * up);
                }
            }
        } else {
            float o[4];
            const bool full = nb + 3 < a.N;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = nb + r;
                float t = v[r];
                if (n < a.N) {
                    if (a.bias) t += bf2f(a.bias[n]);
                    t = bfround(t);
                    if constexpr (EPI == EPI_SILU) t = bfround(silu(t));
                    if constexpr (EPI == EPI_GELU) t = bfround(gelu_erf(t));
                    if constexpr (EPI == EPI_RESID) t = t + bf2f(a.res[(size_t)m * a.ldres + n]);
                }
                o[r] = t;
            }
            bf16_t* dst = a.out + (size_t)m * a.ldo + nb;
            if (full && ((reinterpret_cast<size_t>(dst) & 7) == 0)) {
                uint2 pk;
                pk.x = packbf(o[0], o[1]); pk.y = packbf(o[2], o[3]);
                *reinterpret_cast<uint2*>(dst) = pk;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (nb + r < a.N) dst[r] = f2bf(o[r]);
            }
        }
    }
}

template <int RN, int MR, int NSTG, int KCH>
int launch_thin(const GemvArgs& a, hipStream_t s) {
    const dim3 grid((a.N + RN - 1) / RN), block(256);
#define EMU_TH_CASE(E) case E: hipLaunchKernelGGL((thin_stream_kernel<E, RN, MR, NSTG, KCH>), grid, block, 0, s, a); break;
    switch (a.epi) {
        EMU_TH_CASE(EPI_NONE)
        EMU_TH_CASE(EPI_RESID)
        EMU_TH_CASE(EPI_SWIGLU)
        EMU_TH_CASE(EPI_SILU)
        EMU_TH_CASE(EPI_GELU)
        default: return -22;
    }
#undef EMU_TH_CASE
    EMU_CHECK_LAUNCH();
    return 0;
}

}  // namespace

bool gemv_thin_ok(const GemvArgs& a) {
    return a.M >= 2 && a.M <= 16 && !a.norm_w && !a.wscale && a.K >= 256 && (a.K % 256) == 0 && (a.ldw & 7) == 0 && (a.ldx & 7) == 0 &&
           (reinterpret_cast<size_t>(a.W) & 15) == 0 && (reinterpret_cast<size_t>(a.x) & 15) == 0 &&
           !(a.epi == EPI_SWIGLU && (a.N & 1));
}

// 16 weight rows per workgroup (three co-resident workgroups per CU at <= 8 rows): measured at 5 rows against 32-row
// workgroups and 512-k stages (tools/thin_ab.py, profiles/r03_thin_stream_ab.log): qkv 42.1 vs 45.5 / 47.1 us.
int launch_gemv_thin(const GemvArgs& a, hipStream_t s) {
    if (!gemv_thin_ok(a)) return -22;
    const int variant = (emu_gemm_tune_get() >> 8) & 15;          // A/B aid
    if (a.M <= 8) {
        switch (variant) {
            case 1: return (a.N + 31) / 32 >= 512 ? launch_thin<32, 8, 3, 32>(a, s) : launch_thin<16, 8, 4, 32>(a, s);
            case 2: return launch_thin<16, 8, 6, 32>(a, s);
            case 3: if (a.K % 512 == 0) return launch_thin<16, 8, 3, 64>(a, s); return launch_thin<16, 8, 4, 32>(a, s);
            default: return launch_thin<16, 8, 4, 32>(a, s);
        }
    }
    switch (variant) {
        case 1: return (a.N + 31) / 32 >= 512 ? launch_thin<32, 16, 3, 32>(a, s) : launch_thin<16, 16, 4, 32>(a, s);
        case 2: return launch_thin<16, 16, 3, 32>(a, s);
        default: return launch_thin<16, 16, 4, 32>(a, s);
    }
}
