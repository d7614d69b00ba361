// UNet denoise helper kernels (NHWC activations), all HBM-bound: GroupNorm(+SiLU), channel concat, conv_in im2col with
// the scheduler's input scaling, fused classifier-free-guidance + Euler step.
// Replaces (reference call sites): diffusers UNet2DConditionModel GroupNorm/SiLU/cat ops reached from
// Emu2/emu/diffusion.py:136-141; scheduler.scale_model_input (:134), CFG (:144-146), scheduler.step (:149).
#include "common.h"
#include "kernels.h"

namespace {

constexpr int GN_DEPTH = 16;         // pixel rows each thread walks serially in the partial-sum pass

// How the 256 threads of a partial-sum workgroup are laid out for C channels: nv 16-byte channel vectors per pixel row,
// RG row groups side by side (all 256 threads busy for C = 320 as well as 1280), GN_DEPTH rows per thread.
__host__ __device__ inline int gn_row_groups(int C) { const int nv = C >> 3; return nv >= 256 ? 1 : 256 / nv; }
__host__ __device__ inline int gn_rows_per_block(int C) { return GN_DEPTH * gn_row_groups(C); }

// partial sums per (b, chunk, channel): ws[((b*nchunk + chunk)*2 + {0,1})*C + c]; row groups meet in LDS in a fixed
// order (deterministic sums)
__global__ __launch_bounds__(256) void gn_partial_kernel(const bf16_t* __restrict__ x, float* __restrict__ ws, int HW, int C) {
    __shared__ float sm[256 * 16];
    const int b = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x;
    const int nv = C >> 3;
    const int RG = gn_row_groups(C), rows = GN_DEPTH * RG;
    const int r0 = chunk * rows, r1 = min(HW, r0 + rows);
    float* out = ws + ((size_t)(b * nchunk + chunk) * 2) * C;
    const int tid = threadIdx.x;
    if (RG == 1) {                                   // wide rows: one thread per channel vector (loop when nv > 256)
        for (int vi = tid; vi < nv; vi += 256) {
            float s[8] = {0, 0, 0, 0, 0, 0, 0, 0}, q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll 4
            for (int r = r0; r < r1; ++r) {
                float f[8];
                unpack8(ld16(x + ((size_t)b * HW + r) * C + vi * 8), f);
#pragma unroll
                for (int e = 0; e < 8; ++e) { s[e] += f[e]; q[e] += f[e] * f[e]; }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) { out[vi * 8 + e] = s[e]; out[C + vi * 8 + e] = q[e]; }
        }
        return;
    }
    const int tr = tid / nv, vi = tid - tr * nv;
    const bool on = tr < RG;
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0}, q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (on) {
#pragma unroll 4
        for (int r = r0 + tr; r < r1; r += RG) {
            float f[8];
            unpack8(ld16(x + ((size_t)b * HW + r) * C + vi * 8), f);
#pragma unroll
            for (int e = 0; e < 8; ++e) { s[e] += f[e]; q[e] += f[e] * f[e]; }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) { sm[tid * 16 + e] = s[e]; sm[tid * 16 + 8 + e] = q[e]; }
    }
    __syncthreads();
    if (tid < nv) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float ts = 0.f, tq = 0.f;
            for (int g = 0; g < RG; ++g) { ts += sm[(g * nv + tid) * 16 + e]; tq += sm[(g * nv + tid) * 16 + 8 + e]; }
            out[tid * 8 + e] = ts;
            out[C + tid * 8 + e] = tq;
        }
    }
}

// one workgroup per (b, group): reduce chunks and channels in double, emit scale/shift per channel
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ ws, const bf16_t* __restrict__ gamma,
                                                         const bf16_t* __restrict__ beta, float* __restrict__ ab, int nchunk, int HW,
                                                         int C, int groups, float eps) {
    __shared__ double red[2][4];
    const int b = blockIdx.y, g = blockIdx.x, cg = C / groups, tid = threadIdx.x;
    double s = 0.0, q = 0.0;
    for (int i = tid; i < nchunk * cg; i += 256) {
        const int chunk = i / cg, c = g * cg + i % cg;
        const float* p = ws + ((size_t)(b * nchunk + chunk) * 2) * C;
        s += (double)p[c];
        q += (double)p[C + c];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
    if ((tid & 63) == 0) { red[0][tid >> 6] = s; red[1][tid >> 6] = q; }
    __syncthreads();
    s = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    q = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    const double n = (double)HW * cg;
    const double mean = s / n;
    double var = q / n - mean * mean;
    var = var < 0.0 ? 0.0 : var;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    float* A = ab + (size_t)b * 2 * C;
    for (int i = tid; i < cg; i += 256) {
        const int c = g * cg + i;
        const float ga = bf2f(gamma[c]) * rstd;
        A[c] = ga;
        A[C + c] = bf2f(beta[c]) - (float)mean * ga;
    }
}

template <bool SILU>
__global__ __launch_bounds__(256) void gn_apply_kernel(const bf16_t* __restrict__ x, const float* __restrict__ ab,
                                                       bf16_t* __restrict__ y, int HW, int C, size_t total_vec) {
    const int nv = C >> 3;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total_vec; i += (size_t)gridDim.x * 256) {
        const size_t row = i / nv;
        const int vi = (int)(i - row * nv);
        const int b = (int)(row / HW);
        const float* A = ab + (size_t)b * 2 * C + vi * 8;
        float f[8];
        unpack8(ld16(x + row * C + vi * 8), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = f[e] * A[e] + A[C + e];
            if (SILU) v = silu(bfround(v));               // GroupNorm output is a bf16 tensor before F.silu
            f[e] = v;
        }
        st16(y + row * C + vi * 8, pack8(f));
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
    const int rows = gn_rows_per_block(C);
    const int nchunk = (HW + rows - 1) / rows;         // monotone in C and HW: sizing for (max C, max HW) covers every use
    return (size_t)B * nchunk * 2 * C + (size_t)B * 2 * C;
}

int launch_groupnorm(const bf16_t* x, const bf16_t* gamma, const bf16_t* beta, bf16_t* y, float* ws, int B, int HW, int C,
                     int groups, float eps, int do_silu, hipStream_t s) {
    if (B < 1 || HW < 1 || (C & 7) || C % groups) return -22;
    const int rows = gn_rows_per_block(C);
    const int nchunk = (HW + rows - 1) / rows;
    float* ab = ws + (size_t)B * nchunk * 2 * C;
    hipLaunchKernelGGL(gn_partial_kernel, dim3(nchunk, B), dim3(256), 0, s, x, ws, HW, C);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(groups, B), dim3(256), 0, s, ws, gamma, beta, ab, nchunk, HW, C, groups, eps);
    const size_t total = (size_t)B * HW * (C >> 3);
    const int grid = (int)min((size_t)8192, (total + 255) / 256);
    if (do_silu) hipLaunchKernelGGL(gn_apply_kernel<true>, dim3(grid), dim3(256), 0, s, x, ab, y, HW, C, total);
    else hipLaunchKernelGGL(gn_apply_kernel<false>, dim3(grid), dim3(256), 0, s, x, ab, y, HW, C, total);
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
