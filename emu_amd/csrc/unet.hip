// UNet denoise helper kernels (NHWC activations), all HBM-bound: GroupNorm(+SiLU), channel concat, conv_in im2col with
// the scheduler's input scaling, fused classifier-free-guidance + Euler step.
// Replaces (reference call sites): diffusers UNet2DConditionModel GroupNorm/SiLU/cat ops reached from
// Emu2/emu/diffusion.py:136-141; scheduler.scale_model_input (:134), CFG (:144-146), scheduler.step (:149).
#include "common.h"
#include "kernels.h"

namespace {

constexpr int GN_DEPTH = 16;         // pixel rows each thread walks serially in the statistics pass
constexpr int GN_MAXC = 2560;        // widest activation of the SDXL topology (first resnet of the up path: 1280 + 1280)
constexpr int GN_MAXG = 32;         // groups (the reference: norm_num_groups 32)
constexpr int GN_MAXCHUNK = 256;    // pixel chunks per batch element in the statistics pass (32 per fold batch of the apply pass)

// How the 256 threads of a statistics workgroup are laid out for C channels: nv 16-byte channel vectors per pixel row,
// RG row groups side by side (all 256 threads busy for C = 320 as well as 1280), GN_DEPTH rows per thread.
__host__ __device__ inline int gn_row_groups(int C) { const int nv = C >> 3; return nv >= 256 ? 1 : 256 / nv; }
__host__ __device__ inline int gn_rows_per_block(int C) { return GN_DEPTH * gn_row_groups(C); }

// Pass 1 of 2: per (batch, pixel chunk, channel slice) the fp32 (sum, sum of squares) of every GROUP of the slice:
// ws[((b * nchunk + chunk) * G + g) * 2].  The channel axis is cut into gridDim.z slices of whole groups so that even the small
// maps (32 chunks x 2 batch rows) put >= 256 workgroups on the chip.  Channel sums of the chunk meet in LDS and are folded to
// groups in a fixed order (deterministic).  One launch of the former three (per-channel partials / finalize / apply) is gone:
// the apply pass derives mean / rstd itself from these few floats.
__global__ __launch_bounds__(256) void gn_stats_kernel(const bf16_t* __restrict__ x, float* __restrict__ ws, int HW, int C, int G,
                                                       int rows) {
    __shared__ float csum[2 * GN_MAXC];               // per-channel (sum | sum of squares) of this chunk and slice
    __shared__ float part[256 * 16];                  // row groups side by side (narrow rows)
    const int b = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x;
    const int Cs = C / gridDim.z, c0 = blockIdx.z * Cs, Gs = G / gridDim.z;
    const int nv = Cs >> 3;
    const int RG = gn_row_groups(Cs);
    const int r0 = chunk * rows, r1 = min(HW, r0 + rows);
    const int tid = threadIdx.x;
    const bf16_t* xs = x + (size_t)b * HW * C + c0;
    if (RG == 1) {                                   // wide rows: one thread per channel vector (loop when nv > 256)
        for (int vi = tid; vi < nv; vi += 256) {
            float s[8] = {0, 0, 0, 0, 0, 0, 0, 0}, q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll 4
            for (int r = r0; r < r1; ++r) {
                float f[8];
                unpack8(ld16(xs + (size_t)r * C + vi * 8), f);
#pragma unroll
                for (int e = 0; e < 8; ++e) { s[e] += f[e]; q[e] += f[e] * f[e]; }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) { csum[vi * 8 + e] = s[e]; csum[Cs + vi * 8 + e] = q[e]; }
        }
    } else {
        const int tr = tid / nv, vi = tid - tr * nv;
        if (tr < RG) {
            float s[8] = {0, 0, 0, 0, 0, 0, 0, 0}, q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll 4
            for (int r = r0 + tr; r < r1; r += RG) {
                float f[8];
                unpack8(ld16(xs + (size_t)r * C + vi * 8), f);
#pragma unroll
                for (int e = 0; e < 8; ++e) { s[e] += f[e]; q[e] += f[e] * f[e]; }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) { part[tid * 16 + e] = s[e]; part[tid * 16 + 8 + e] = q[e]; }
        }
        __syncthreads();
        if (tid < nv) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float ts = 0.f, tq = 0.f;
                for (int g = 0; g < RG; ++g) { ts += part[(g * nv + tid) * 16 + e]; tq += part[(g * nv + tid) * 16 + 8 + e]; }
                csum[tid * 8 + e] = ts;
                csum[Cs + tid * 8 + e] = tq;
            }
        }
    }
    __syncthreads();
    if (tid < Gs) {
        const int cg = C / G;
        float ts = 0.f, tq = 0.f;
        for (int i = 0; i < cg; ++i) { ts += csum[tid * cg + i]; tq += csum[Cs + tid * cg + i]; }
        float* out = ws + ((size_t)(b * nchunk + chunk) * G + blockIdx.z * Gs + tid) * 2;
        out[0] = ts; out[1] = tq;
    }
}

// Pass 2 of 2: every workgroup first folds the chunk partials of its batch into (mean, rstd) per group (double accumulation,
// fixed order; a few KB from L2) and from them the per-channel scale / shift table y = x * A[c] + B[c] in LDS, then applies it
// (+ SiLU) over its rows.
template <bool SILU>
__global__ __launch_bounds__(256) void gn_apply_kernel(const bf16_t* __restrict__ x, const float* __restrict__ ws,
                                                       const bf16_t* __restrict__ gamma, const bf16_t* __restrict__ beta,
                                                       bf16_t* __restrict__ y, int HW, int C, int G, int nchunk, float eps) {
    __shared__ double red[2][8][GN_MAXG];
    __shared__ float stat[2][GN_MAXG];                 // mean | rstd per group
    __shared__ __attribute__((aligned(16))) float ab[2][GN_MAXC];     // scale | shift per channel
    const int b = blockIdx.y, tid = threadIdx.x;
    {   // chunks x GN_MAXG groups: thread (slice, g) fetches 4 chunk pairs per batch of loads (indices clamped, no branch around a
        // load) and sums them; the 8 slices meet in LDS in a fixed order
        const int g = tid & (GN_MAXG - 1), sl = tid >> 5;
        const float* p = ws + ((size_t)b * nchunk * G + (g < G ? g : 0)) * 2;
        double sd = 0.0, qd = 0.0;
        for (int base = 0; base < nchunk; base += 32) {      // one trip for the UNet's maps; the VAE's megapixel maps take a few
            f32x2_t v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = base + sl + 8 * k;
                v[k] = *reinterpret_cast<const f32x2_t*>(p + (size_t)(c < nchunk ? c : 0) * G * 2);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool in = base + sl + 8 * k < nchunk && g < G;
                sd += in ? (double)v[k][0] : 0.0;
                qd += in ? (double)v[k][1] : 0.0;
            }
        }
        red[0][sl][g] = sd; red[1][sl][g] = qd;
        __syncthreads();
        if (tid < G) {
            double ts = 0.0, tq = 0.0;
#pragma unroll
            for (int k = 0; k < 8; ++k) { ts += red[0][k][tid]; tq += red[1][k][tid]; }
            const double n = (double)HW * (C / G);
            const double mean = ts / n;
            double var = tq / n - mean * mean;
            var = var < 0.0 ? 0.0 : var;
            stat[0][tid] = (float)mean;
            stat[1][tid] = (float)(1.0 / sqrt(var + (double)eps));
        }
        __syncthreads();
        const int cg = C / G;
        for (int c = tid; c < C; c += 256) {
            const int gi = c / cg;
            const float A = bf2f(gamma[c]) * stat[1][gi];
            ab[0][c] = A;
            ab[1][c] = bf2f(beta[c]) - stat[0][gi] * A;
        }
        __syncthreads();
    }
    const int nv = C >> 3;
    const size_t total = (size_t)HW * nv;
    const bf16_t* xb = x + (size_t)b * HW * C;
    bf16_t* yb = y + (size_t)b * HW * C;
    for (size_t i = (size_t)blockIdx.x * 256 + tid; i < total; i += (size_t)gridDim.x * 256) {
        const size_t row = i / nv;
        const int vi = (int)(i - row * nv);
        float f[8];
        unpack8(ld16(xb + row * C + vi * 8), f);
        const f32x4_t a0 = *reinterpret_cast<const f32x4_t*>(&ab[0][vi * 8]), a1 = *reinterpret_cast<const f32x4_t*>(&ab[0][vi * 8 + 4]);
        const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(&ab[1][vi * 8]), b1 = *reinterpret_cast<const f32x4_t*>(&ab[1][vi * 8 + 4]);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = f[e] * (e < 4 ? a0[e & 3] : a1[e & 3]) + (e < 4 ? b0[e & 3] : b1[e & 3]);
            if (SILU) v = silu(bfround(v));               // GroupNorm output is a bf16 tensor before F.silu
            f[e] = v;
        }
        st16(yb + row * C + vi * 8, pack8(f));
    }
}

__global__ __launch_bounds__(256) void concat_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b, bf16_t* __restrict__ out,
                                                     int C1, int C2, size_t total_vec) {
    const int nv = (C1 + C2) >> 3, n1 = C1 >> 3;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total_vec; i += (size_t)gridDim.x * 256) {
        const size_t row = i / nv;
        const int vi = (int)(i - row * nv);
        const u32x4 v = vi < n1 ? ld16(a + row * C1 + vi * 8) : ld16(b + row * C2 + (vi - n1) * 8);
        st16(out + row * (C1 + C2) + vi * 8, v);
    }
}

__global__ __launch_bounds__(256) void prep_input_kernel(const bf16_t* __restrict__ lat, const float* __restrict__ sigmas,
                                                         const int32_t* __restrict__ step, bf16_t* __restrict__ out, int C, int H, int W,
                                                         int Kpad) {
    const float sg = sigmas[*step];
    const int HW = H * W;
    const size_t total = (size_t)HW * Kpad;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int p = (int)(i / Kpad), k = (int)(i - (size_t)p * Kpad);
        bf16_t v = 0;
        if (k < 9 * C) {
            const int tap = k / C, c = k - tap * C, ky = tap / 3, kx = tap - ky * 3;
            const int y = p / W + ky - 1, x = p % W + kx - 1;
            if (y >= 0 && y < H && x >= 0 && x < W) v = f2bf(bf2f(lat[(size_t)c * HW + y * W + x]) / sqrtf(sg * sg + 1.0f));
        }
        out[i] = v;
        out[total + i] = v;                                // cat([latents] * 2)
    }
}

__global__ __launch_bounds__(256) void cfg_euler_kernel(const bf16_t* __restrict__ eps, bf16_t* __restrict__ lat,
                                                        const float* __restrict__ sigmas, const int32_t* __restrict__ step, float g,
                                                        int C, int HW) {
    const int st = *step;
    const float sigma = sigmas[st], dt = sigmas[st + 1] - sigma;
    const int i = blockIdx.x * 256 + threadIdx.x;            // over HW * C
    if (i < HW * C) {
        const int p = i / C, c = i - p * C;
        const float ec = bf2f(eps[(size_t)p * C + c]);                  // cond first (diffusion.py:145)
        const float eu = bf2f(eps[((size_t)HW + p) * C + c]);
        const float e = bfround(eu + bfround(g * bfround(ec - eu)));
        const size_t li = (size_t)c * HW + p;
        const float x = bf2f(lat[li]);
        const float x0 = bfround(x - bfround(sigma * e));               // pred_original_sample
        const float der = bfround(bfround(x - x0) / sigma);
        lat[li] = f2bf(x + bfround(der * dt));
    }
}

__global__ void step_inc_kernel(int32_t* step) { *step += 1; }

__global__ __launch_bounds__(256) void add_silu_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b, bf16_t* __restrict__ sum_out,
                                                       bf16_t* __restrict__ silu_out, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const float s = bfround(bf2f(a[i]) + bf2f(b[i]));
        if (sum_out) sum_out[i] = f2bf(s);
        silu_out[i] = f2bf(silu(s));
    }
}

__global__ __launch_bounds__(256) void gather_step_row_kernel(const bf16_t* __restrict__ table, const int32_t* __restrict__ step,
                                                              bf16_t* __restrict__ out, int rows, int cols) {
    const bf16_t* src = table + (size_t)(*step) * cols;
    for (int i = threadIdx.x; i < rows * cols; i += 256) out[i] = src[i % cols];
}

}  // namespace

size_t gn_ws_floats(int B, int C, int HW) {
    (void)C; (void)HW;
    return (size_t)B * GN_MAXCHUNK * GN_MAXG * 2;      // 128 KiB per batch element: every chunking launch_groupnorm may choose
}

int launch_groupnorm(const bf16_t* x, const bf16_t* gamma, const bf16_t* beta, bf16_t* y, float* ws, int B, int HW, int C,
                     int groups, float eps, int do_silu, hipStream_t s) {
    if (B < 1 || HW < 1 || (C & 7) || C % groups || groups > GN_MAXG || C > GN_MAXC) return -22;
    // GN_MAXCHUNK chunks of whole thread-layout passes per batch element (fewer for small maps): the apply pass folds exactly
    // one batch of loads per thread
    // channel slices of whole groups and whole 16-byte vectors (4, 2 or 1): more workgroups for the statistics pass
    int cs = 4;
    while (cs > 1 && (groups % cs || ((C / cs) & 7) || C / cs < 64)) cs >>= 1;
    const int unit = gn_rows_per_block(C / cs);
    int nchunk = (HW + unit - 1) / unit;
    // 32 chunks (one fold batch in the apply pass) unless the map is so large that 32 x B x cs workgroups would each walk more
    // than ~2 MB (the VAE's 512^2 / 1024^2 maps): then up to GN_MAXCHUNK
    int cap = 32;
    while (cap < GN_MAXCHUNK && (size_t)HW * (C / cs) * 2 / cap > (2u << 20)) cap <<= 1;
    if (nchunk > cap) nchunk = cap;
    const int rows = (HW + nchunk - 1) / nchunk;
    nchunk = (HW + rows - 1) / rows;
    hipLaunchKernelGGL(gn_stats_kernel, dim3(nchunk, B, cs), dim3(256), 0, s, x, ws, HW, C, groups, rows);
    const size_t total = (size_t)HW * (C >> 3);
    const int grid = (int)min((size_t)4096, (total + 255) / 256);
    if (do_silu) hipLaunchKernelGGL(gn_apply_kernel<true>, dim3(grid, B), dim3(256), 0, s, x, ws, gamma, beta, y, HW, C, groups, nchunk, eps);
    else hipLaunchKernelGGL(gn_apply_kernel<false>, dim3(grid, B), dim3(256), 0, s, x, ws, gamma, beta, y, HW, C, groups, nchunk, eps);
    EMU_CHECK_LAUNCH();
    return 0;
}

int launch_concat_channels(const bf16_t* a, const bf16_t* b, bf16_t* out, int rows, int C1, int C2, hipStream_t s) {
    if (rows < 1 || (C1 & 7) || (C2 & 7)) return -22;
    const size_t total = (size_t)rows * ((C1 + C2) >> 3);
    const int grid = (int)min((size_t)8192, (total + 255) / 256);
    hipLaunchKernelGGL(concat_kernel, dim3(grid), dim3(256), 0, s, a, b, out, C1, C2, total);
    EMU_CHECK_LAUNCH();
    return 0;
}

int launch_unet_prep_input(const bf16_t* latents, const float* sigmas, const int32_t* step, bf16_t* out, int C, int H, int W,
                           int Kpad, hipStream_t s) {
    if (Kpad < 9 * C || (Kpad & 7)) return -22;
    const size_t total = (size_t)H * W * Kpad;
    const int grid = (int)min((size_t)8192, (total + 255) / 256);
    hipLaunchKernelGGL(prep_input_kernel, dim3(grid), dim3(256), 0, s, latents, sigmas, step, out, C, H, W, Kpad);
    EMU_CHECK_LAUNCH();
    return 0;
}

int launch_cfg_euler_step(const bf16_t* eps_nhwc, bf16_t* latents, const float* sigmas, int32_t* step, float guidance, int C,
                          int HW, hipStream_t s) {
    hipLaunchKernelGGL(cfg_euler_kernel, dim3((HW * C + 255) / 256), dim3(256), 0, s, eps_nhwc, latents, sigmas, step, guidance, C, HW);
    hipLaunchKernelGGL(step_inc_kernel, dim3(1), dim3(1), 0, s, step);
    EMU_CHECK_LAUNCH();
    return 0;
}

int launch_add_silu(const bf16_t* a, const bf16_t* b, bf16_t* sum_out, bf16_t* silu_out, int n, hipStream_t s) {
    hipLaunchKernelGGL(add_silu_kernel, dim3((n + 255) / 256), dim3(256), 0, s, a, b, sum_out, silu_out, n);
    EMU_CHECK_LAUNCH();
    return 0;
}

int launch_gather_step_row(const bf16_t* table, const int32_t* step, bf16_t* out, int rows, int cols, hipStream_t s) {
    hipLaunchKernelGGL(gather_step_row_kernel, dim3(1), dim3(256), 0, s, table, step, out, rows, cols);
    EMU_CHECK_LAUNCH();
    return 0;
}
