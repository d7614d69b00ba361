// o_proj of a tensor-parallel shard's one-row decode step with the decode attention's split merge in its PROLOGUE:
//     out[n] = epilogue( sum_k x[k] W[n, k] ),   x[h * 128 + d] = bf16( sum_s f_s o_s[h][d] / sum_s f_s l_s[h] ),  f_s = exp(m_s[h] - max_s m_s[h])
// over the live 128-key splits s that decode_fused_kernel (attention.hip) left in its workspace.  A shard's attention output is
// short (7 heads = 896 values at TP = 8), so every wave of the projection can merge the whole vector itself from L2 while its
// weight rows are on their way from HBM, and the decode_fused_combine_kernel launch -- 4.8 of the 52.7 us a TP = 8 shard's layer
// takes, all of it latency (profiles/r05_tp8_shard_decode_kernel_stats.csv) -- disappears.  MEASURED LEVEL with the two launches
// (3.15 vs 3.15-3.18 ms per token of a TP = 8 shard): the merge is a dependent L2 trip inside the projection, which is what the
// launch cost.  So the engine takes this path only on request (emu_gemm_tune bit 19); it stays as the tested form of the idea.  Same arithmetic, order and rounding
// as the combine kernel followed by gemv_wave_kernel (gemv.hip): bit-identical rows.  One wave per 4 weight rows; every load of the
// workgroup (thread t: the 8 split states of 4 values of the vector, its wave's weight rows) is issued before anything is consumed,
// the merged vector meets the waves in LDS behind one raw barrier, splits beyond the live ones are loaded (stale, possibly NaN)
// and dropped by select.  K = heads_local * 128 <= 1024 (two trips of
// 64 lanes x 8 values), <= 8 splits (context <= 1024), one batch row; the engine falls back to the two launches otherwise.
//
// Replaces (reference call sites): LlamaAttention's softmax / value product and o_proj of a cached one-token step, reached from
// Emu2/emu/emu.py:213-229 (greedy text) and :133-138 (generate_image), under the tensor-parallel plan of emu_amd/tp.py (the reference
// places whole layers on devices instead: Emu2/emu/mixin.py:44-81).
#include "common.h"
#include "kernels.h"

namespace {
constexpr int MG_NS = 8, MG_D = 128, MG_RW = 4;
typedef float f32x4u_t __attribute__((ext_vector_type(4), aligned(8)));    // split rows are 130 floats apart: 8-byte aligned only
typedef float f32x2u_t __attribute__((ext_vector_type(2), aligned(8)));

template <int KITW, int EPI>
__global__ __launch_bounds__(256) void gemv_merge_kernel(const GemvMergeArgs a) {
    // the merged vector is formed ONCE per workgroup (thread t: 4 consecutive values of one head, all 8 split states requested up
    // front) and handed to the four waves through LDS -- a first version merged per wave: 1664 waves x 41 KB of split states from
    // L2 cost the launch what the combine launch had cost (3.15 vs 3.15 ms per token, profiles/r05_tp_emulate_merged_o_proj.log)
    __shared__ __attribute__((aligned(16))) bf16_t xs[1024];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int KV = a.K >> 3;
    const int n0 = (blockIdx.x * 4 + wave) * MG_RW;
    const int e0 = tid * 4;                                             // this thread's 4 values of the merged vector
    const bool mine = e0 < a.K;
    const int ec = mine ? e0 : 0;
    const int h = ec / MG_D, d = ec % MG_D;
    const float* w = a.ws + (size_t)h * a.nsplit * (MG_D + 2);
    f32x4u_t pa[MG_NS];
    f32x2u_t ps[MG_NS];
#pragma unroll
    for (int s = 0; s < MG_NS; ++s) {
        const float* ws = w + (s < a.nsplit ? s : a.nsplit - 1) * (MG_D + 2);         // never past the workspace
        pa[s] = *reinterpret_cast<const f32x4u_t*>(ws + d);
        ps[s] = *reinterpret_cast<const f32x2u_t*>(ws + MG_D);
    }
    u32x4 wv[KITW][MG_RW];
#pragma unroll
    for (int it = 0; it < KITW; ++it) {
        const int vi = lane + 64 * it;
        const int vc = vi < KV ? vi : KV - 1;                         // clamped; the activation is zeroed instead
#pragma unroll
        for (int r = 0; r < MG_RW; ++r) {
            int n = n0 + r;
            n = n < a.N ? n : a.N - 1;
            wv[it][r] = ld_stream(reinterpret_cast<const u32x4*>(a.W + (size_t)n * a.ldw + vc * 8));
        }
    }
    const int nlive = (a.slot[0] + 128) / 128;                         // ceil((slot + 1) / 128): dead splits wrote nothing
    {
        // decode_fused_combine_kernel's statement order: running max, then f, num, den split by split
        float m = -INFINITY;
#pragma unroll
        for (int s = 0; s < MG_NS; ++s) m = s < nlive ? fmaxf(m, ps[s].x) : m;
        float num[4] = {0.f, 0.f, 0.f, 0.f}, den = 0.f;
#pragma unroll
        for (int s = 0; s < MG_NS; ++s) {
            const float ms = ps[s].x;
            const bool live = s < nlive;
            const float f = (!live || ms == -INFINITY) ? 0.f : __expf(ms - m);
            const float o[4] = {pa[s].x, pa[s].y, pa[s].z, pa[s].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) num[j] = live ? fmaf(f, o[j], num[j]) : num[j];
            den = live ? fmaf(f, ps[s].y, den) : den;
        }
        float x[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) x[j] = den > 0.f ? num[j] / den : 0.f;
        if (mine) {
            *reinterpret_cast<volatile uint32_t*>(xs + e0) = packbf(x[0], x[1]);
            *reinterpret_cast<volatile uint32_t*>(xs + e0 + 2) = packbf(x[2], x[3]);
        }
    }
    // (a raw barrier behind an LDS-only wait: __syncthreads would drain vmcnt and serialise behind the weight rows still in flight)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    u32x4 xv[KITW];
#pragma unroll
    for (int it = 0; it < KITW; ++it) {
        const int vi = lane + 64 * it;
        const volatile u32x4* src = reinterpret_cast<const volatile u32x4*>(xs + (vi < KV ? vi : 0) * 8);
        u32x4 t;
        t.x = src->x; t.y = src->y; t.z = src->z; t.w = src->w;
        xv[it] = vi < KV ? t : u32x4{0u, 0u, 0u, 0u};
    }
    float acc[MG_RW];
#pragma unroll
    for (int r = 0; r < MG_RW; ++r) acc[r] = 0.f;
#pragma unroll
    for (int it = 0; it < KITW; ++it)
#pragma unroll
        for (int r = 0; r < MG_RW; ++r) {
            float t = acc[r];
            t = bf16_dot2(wv[it][r].x, xv[it].x, t);
            t = bf16_dot2(wv[it][r].y, xv[it].y, t);
            t = bf16_dot2(wv[it][r].z, xv[it].z, t);
            t = bf16_dot2(wv[it][r].w, xv[it].w, t);
            acc[r] = t;
        }
#pragma unroll
    for (int r = 0; r < MG_RW; ++r) acc[r] = wave_sum(acc[r]);
    if (lane == 0) {
#pragma unroll
        for (int r = 0; r < MG_RW; ++r) {
            const int n = n0 + r;
            if (n < a.N) {
                float v = bfround(acc[r]);
                if constexpr (EPI == EPI_RESID) v = v + bf2f(a.res[n]);
                a.out[n] = f2bf(v);
            }
        }
    }
    // (optional: the merged vector itself, for the callers that look at the attention output -- tests)
    if (a.x_out && blockIdx.x == 0 && wave == 0) {
#pragma unroll
        for (int it = 0; it < KITW; ++it)
            if (lane + 64 * it < KV) st16(a.x_out + (size_t)(lane + 64 * it) * 8, xv[it]);
    }
}

template <int KITW>
int launch_merge(const GemvMergeArgs& a, hipStream_t s) {
    const dim3 grid((a.N + 4 * MG_RW - 1) / (4 * MG_RW)), block(256);
    if (a.epi == EPI_RESID) hipLaunchKernelGGL((gemv_merge_kernel<KITW, EPI_RESID>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((gemv_merge_kernel<KITW, EPI_NONE>), grid, block, 0, s, a);
    EMU_CHECK_LAUNCH();
    return 0;
}
}  // namespace

bool gemv_merge_ok(int heads, int D, int N, int nsplit) {
    return D == MG_D && heads >= 1 && heads * MG_D <= 1024 && N >= 1024 && nsplit >= 1 && nsplit <= MG_NS;
}

int launch_gemv_merge(const GemvMergeArgs& a, hipStream_t s) {
    if (!a.ws || !a.slot || !a.W || !a.out || a.K != a.H * MG_D || (a.ldw & 7) || (a.epi != EPI_NONE && a.epi != EPI_RESID) ||
        (a.epi == EPI_RESID && !a.res) || !gemv_merge_ok(a.H, MG_D, a.N, a.nsplit))
        return -22;
    return (a.K >> 3) <= 64 ? launch_merge<1>(a, s) : launch_merge<2>(a, s);
}
