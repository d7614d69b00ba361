// Row-wise and gather/scatter kernels of the Emu2 hot path.  All HBM-bound: 16-byte vector accesses,
// fp32 math, bf16 rounding points identical to the reference's torch ops.
#include "common.h"
#include "kernels.h"

namespace {

// LlamaRMSNorm (transformers; reached from Emu2/emu/emu.py:133-138): y = bf16(w * bf16(x * rsqrt(mean(x^2)+eps)))
__global__ __launch_bounds__(256) void rmsnorm_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                      bf16_t* __restrict__ y, int cols, int ldx, int ldy, float eps) {
    __shared__ float scratch[4];
    const bf16_t* xr = x + (size_t)blockIdx.x * ldx;
    bf16_t* yr = y + (size_t)blockIdx.x * ldy;
    const int nv = cols >> 3;
    float ss = 0.f;
    for (int vi = threadIdx.x; vi < nv; vi += 256) {
        float f[8];
        unpack8(ld16(xr + vi * 8), f);
#pragma unroll
        for (int j = 0; j < 8; ++j) ss += f[j] * f[j];
    }
    const float rinv = rsqrtf(block_sum<4>(ss, scratch) / (float)cols + eps);
    for (int vi = threadIdx.x; vi < nv; vi += 256) {
        float f[8], g[8];
        unpack8(ld16(xr + vi * 8), f);
        unpack8(ld16(w + vi * 8), g);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = g[j] * bfround(f[j] * rinv);
        st16(yr + vi * 8, pack8(f));
    }
}

// nn.LayerNorm(eps) on a bf16 row (fp32 statistics, one rounding), then optional residual add in bf16:
// y = bf16(res + bf16(LN(x)))   -- ViT post-norm block, Emu2/emu/eva_vit.py:298-300
__global__ __launch_bounds__(256) void layernorm_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                        const bf16_t* __restrict__ b, const bf16_t* res,
                                                        bf16_t* y, int cols, float eps) {
    __shared__ float scratch[4];
    const size_t roff = (size_t)blockIdx.x * cols;
    const bf16_t* xr = x + roff;
    const int nv = cols >> 3;
    float s = 0.f;
    for (int vi = threadIdx.x; vi < nv; vi += 256) {
        float f[8];
        unpack8(ld16(xr + vi * 8), f);
#pragma unroll
        for (int j = 0; j < 8; ++j) s += f[j];
    }
    const float mean = block_sum<4>(s, scratch) / (float)cols;
    float v = 0.f;
    for (int vi = threadIdx.x; vi < nv; vi += 256) {
        float f[8];
        unpack8(ld16(xr + vi * 8), f);
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float d = f[j] - mean; v += d * d; }
    }
    const float rstd = rsqrtf(block_sum<4>(v, scratch) / (float)cols + eps);
    for (int vi = threadIdx.x; vi < nv; vi += 256) {
        float f[8], g[8], bb[8];
        unpack8(ld16(xr + vi * 8), f);
        unpack8(ld16(w + vi * 8), g);
        unpack8(ld16(b + vi * 8), bb);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = (f[j] - mean) * rstd * g[j] + bb[j];
        if (res) {
            float r[8];
            unpack8(ld16(res + roff + vi * 8), r);
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = r[j] + bfround(f[j]);
        }
        st16(y + roff + vi * 8, pack8(f));
    }
}

// embed_tokens(input_ids), Emu2/emu/emu.py:119,193
__global__ __launch_bounds__(256) void embed_gather_kernel(const int32_t* __restrict__ ids, const bf16_t* __restrict__ table,
                                                           bf16_t* __restrict__ out, int hidden, int vocab) {
    int id = ids[blockIdx.x];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    const bf16_t* src = table + (size_t)id * hidden;
    bf16_t* dst = out + (size_t)blockIdx.x * hidden;
    for (int vi = threadIdx.x; vi < (hidden >> 3); vi += 256) st16(dst + vi * 8, ld16(src + vi * 8));
}

// text_embeds[input_ids == IMAGE] = image_embeds, Emu2/emu/emu.py:202-203 (row list precomputed on host)
__global__ __launch_bounds__(256) void scatter_rows_kernel(const bf16_t* __restrict__ src, const int32_t* __restrict__ rows,
                                                           bf16_t* __restrict__ out, int hidden) {
    const bf16_t* s = src + (size_t)blockIdx.x * hidden;
    bf16_t* d = out + (size_t)rows[blockIdx.x] * hidden;
    for (int vi = threadIdx.x; vi < (hidden >> 3); vi += 256) st16(d + vi * 8, ld16(s + vi * 8));
}

// greedy selection: first index of the maximum of float(bf16 logits); optional suppressed id (EOS while
// fewer than min_length tokens exist, Emu2/emu/emu.py:220 -> MinLengthLogitsProcessor)
__global__ __launch_bounds__(1024) void argmax_kernel(const bf16_t* __restrict__ logits, int ld, int vocab, int suppress,
                                                      int32_t* __restrict__ out) {
    __shared__ float sv[1024];
    __shared__ int si[1024];
    const bf16_t* row = logits + (size_t)blockIdx.x * ld;
    const int tid = threadIdx.x;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    auto take = [&](float v, int i) {
        if (i == suppress) v = -INFINITY;
        if (v > best || (v == best && i < bi)) { best = v; bi = i; }
    };
    // 16-byte loads over the aligned body (the row start is 16-byte aligned when ld % 8 == 0), scalar head / tail
    const int head = (int)((16 - (reinterpret_cast<size_t>(row) & 15)) & 15) >> 1;
    const int h = head < vocab ? head : vocab;
    const int nv = (vocab - h) >> 3;
    if (tid < h) take(bf2f(row[tid]), tid);
    for (int vi = tid; vi < nv; vi += 1024) {
        float f[8];
        unpack8(ld16(row + h + vi * 8), f);
#pragma unroll
        for (int j = 0; j < 8; ++j) take(f[j], h + vi * 8 + j);
    }
    for (int i = h + nv * 8 + tid; i < vocab; i += 1024) take(bf2f(row[i]), i);
    sv[tid] = best; si[tid] = bi;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if (tid < o) {
            const float v = sv[tid + o]; const int i = si[tid + o];
            if (v > sv[tid] || (v == sv[tid] && i < si[tid])) { sv[tid] = v; si[tid] = i; }
        }
        __syncthreads();
    }
    if (tid == 0) out[blockIdx.x] = si[0];
}

// PatchEmbed conv (stride = kernel = p) as im2col: out[(b, py, px), k = c*p*p + i*p + j], zero-padded to Kpad.
// Emu2/emu/eva_vit.py:327-335.  One workgroup per patch.
template <bool F32>
__global__ __launch_bounds__(256) void patchify_kernel(const void* __restrict__ image, bf16_t* __restrict__ out,
                                                       int C, int HW, int p, int Kpad) {
    const int g = HW / p;
    const int patch = blockIdx.x;                    // b*g*g + py*g + px
    const int b = patch / (g * g), py = (patch / g) % g, px = patch % g;
    bf16_t* dst = out + (size_t)patch * Kpad;
    const int K = C * p * p;
    for (int k = threadIdx.x; k < Kpad; k += 256) {
        bf16_t v = 0;
        if (k < K) {
            const int c = k / (p * p), i = (k / p) % p, j = k % p;
            const size_t src = (((size_t)b * C + c) * HW + (py * p + i)) * HW + (px * p + j);
            v = F32 ? f2bf(reinterpret_cast<const float*>(image)[src]) : reinterpret_cast<const bf16_t*>(image)[src];
        }
        dst[k] = v;
    }
}

// encode_image pooling, Emu2/emu/emu.py:82-89: drop cls, view [B,C,g,g], avg_pool2d(s, s), row-major flatten
__global__ __launch_bounds__(256) void avgpool_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out,
                                                      int g, int C, int s) {
    const int go = g / s;
    const int o = blockIdx.x;                        // b*go*go + py*go + px
    const int b = o / (go * go), py = (o / go) % go, px = o % go;
    const bf16_t* base = x + ((size_t)b * (g * g + 1) + 1) * C;
    const float inv = 1.f / (float)(s * s);
    for (int vi = threadIdx.x; vi < (C >> 3); vi += 256) {
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < s; ++i)
            for (int j = 0; j < s; ++j) {
                float f[8];
                unpack8(ld16(base + (size_t)((py * s + i) * g + (px * s + j)) * C + vi * 8), f);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += f[e];
            }
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] *= inv;
        st16(out + (size_t)o * C + vi * 8, pack8(acc));
    }
}

// cat(cls, patches) + pos_embed, Emu2/emu/eva_vit.py:406-409 (bf16 add)
__global__ __launch_bounds__(256) void vit_assemble_kernel(const bf16_t* __restrict__ patches, const bf16_t* __restrict__ cls,
                                                           const bf16_t* __restrict__ pos, bf16_t* __restrict__ x, int T, int C) {
    const int b = blockIdx.x / (T + 1), t = blockIdx.x % (T + 1);
    const bf16_t* src = t == 0 ? cls : patches + ((size_t)b * T + (t - 1)) * C;
    const bf16_t* pr = pos + (size_t)t * C;
    bf16_t* dst = x + (size_t)blockIdx.x * C;
    for (int vi = threadIdx.x; vi < (C >> 3); vi += 256) {
        float f[8], q[8];
        unpack8(ld16(src + vi * 8), f);
        unpack8(ld16(pr + vi * 8), q);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] += q[e];
        st16(dst + vi * 8, pack8(f));
    }
}

// softmax over one row per workgroup, fp32 statistics; the row is re-read from L2 for each pass.  Optional additive bias
// (T5 relative-position bias + causal mask, bf16): logits = bf16(x * scale + bias) as the bf16 tensor add of the reference.
__global__ __launch_bounds__(256) void softmax_rows_kernel(bf16_t* __restrict__ x, const bf16_t* __restrict__ bias, int cols, int ld,
                                                          int ld_bias, float scale) {
    __shared__ float scratch[4];
    bf16_t* row = x + (size_t)blockIdx.x * ld;
    const bf16_t* brow = bias ? bias + (size_t)blockIdx.x * ld_bias : nullptr;
    const int nv = cols >> 3;
    auto logits = [&](int vi, float* f) {
        unpack8(ld16(row + vi * 8), f);
        if (brow) {
            float b[8];
            unpack8(ld16(brow + vi * 8), b);
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = bfround(f[j] * scale + b[j]);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] *= scale;
        }
    };
    float m = -INFINITY;
    for (int vi = threadIdx.x; vi < nv; vi += 256) {
        float f[8];
        logits(vi, f);
#pragma unroll
        for (int j = 0; j < 8; ++j) m = fmaxf(m, f[j]);
    }
    m = wave_max(m);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(scratch[0], scratch[1]), fmaxf(scratch[2], scratch[3]));
    float l = 0.f;
    for (int vi = threadIdx.x; vi < nv; vi += 256) {
        float f[8];
        logits(vi, f);
#pragma unroll
        for (int j = 0; j < 8; ++j) l += __expf(f[j] - m);
    }
    l = block_sum<4>(l, scratch);
    const float inv = 1.f / l;
    for (int vi = threadIdx.x; vi < nv; vi += 256) {
        float f[8];
        logits(vi, f);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = __expf(f[j] - m) * inv;
        st16(row + vi * 8, pack8(f));
    }
}

// per-row symmetric fp8 (OCP e4m3fn) quantisation of a packed bf16 weight: scale = amax / 448, q = rne_fp8(w / scale)
// NV > 0: the row (K <= 2048 * NV elements) stays in registers between the amax pass and the conversion (one read of the row:
// the activation rows of the W8A8 GEMMs -- 1025 x 15360 after the ViT's GELU, 2048 x 5120 after the UNet's GEGLU -- are
// quantised on the critical path); NV = 0: any K, two passes (weight matrices at load time).  Same values either way.
template <int NV>
__global__ __launch_bounds__(256) void quant_fp8_rows_kernel(const bf16_t* __restrict__ w, int ldw, uint8_t* __restrict__ q,
                                                             int ldq, float* __restrict__ scale, int K) {
    __shared__ float scratch[4];
    const int n = blockIdx.x, tid = threadIdx.x;
    const bf16_t* src = w + (size_t)n * ldw;
    const int nv = K >> 3;
    float amax = 0.f;
    u32x4 keep[NV > 0 ? NV : 1];
    if constexpr (NV > 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int vi = tid + i * 256;
            keep[i] = vi < nv ? ld16(src + vi * 8) : u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            float f[8];
            unpack8(keep[i], f);
#pragma unroll
            for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(f[j]));
        }
    } else {
        for (int vi = tid; vi < nv; vi += 256) {
            float f[8];
            unpack8(ld16(src + vi * 8), f);
#pragma unroll
            for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(f[j]));
        }
    }
    amax = wave_max(amax);
    __syncthreads();
    if ((tid & 63) == 0) scratch[tid >> 6] = amax;
    __syncthreads();
    amax = fmaxf(fmaxf(scratch[0], scratch[1]), fmaxf(scratch[2], scratch[3]));
    const float sc = amax > 0.f ? amax / 448.0f : 1.0f;
    if (tid == 0) scale[n] = sc;
    uint8_t* dst = q + (size_t)n * ldq;
    auto put = [&](int vi, const u32x4& v) {
        float f[8];
        unpack8(v, f);
        int lo = 0, hi = 0;
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0] / sc, f[1] / sc, lo, false);
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2] / sc, f[3] / sc, lo, true);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4] / sc, f[5] / sc, hi, false);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6] / sc, f[7] / sc, hi, true);
        *reinterpret_cast<uint2*>(dst + vi * 8) = make_uint2((unsigned)lo, (unsigned)hi);
    };
    if constexpr (NV > 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int vi = tid + i * 256;
            if (vi < nv) put(vi, keep[i]);
        }
    } else {
        for (int vi = tid; vi < nv; vi += 256) put(vi, ld16(src + vi * 8));
    }
}

// layernorm_kernel + quant_fp8_rows_kernel in one pass over a row of <= 2048 columns (one 16-byte vector per thread, the row
// stays in registers): y = bf16(res + bf16(LN(x))) as layernorm_kernel writes it (optional: y may be null), and the per-row
// e4m3 quantisation of exactly those bf16 values (scale = amax / 448, 1 for a zero row) -- bit-identical to the two launches.
// The W8A8 modes of the ViT / UNet blocks (emu_vit_use_fp8, emu_unet_use_fp8): the LayerNorm in front of a GEMM already holds
// the whole row, so its output leaves as the GEMM's fp8 operand without a quantise launch.
__global__ __launch_bounds__(256) void layernorm_q8_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                           const bf16_t* __restrict__ b, const bf16_t* res, bf16_t* y,
                                                           uint8_t* __restrict__ q, float* __restrict__ scale, int cols, float eps) {
    __shared__ float scratch[4];
    const size_t roff = (size_t)blockIdx.x * cols;
    const int vi = threadIdx.x;
    const bool on = vi < (cols >> 3);
    float f[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (on) unpack8(ld16(x + roff + vi * 8), f);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += f[j];
    const float mean = block_sum<4>(s, scratch) / (float)cols;
    float v = 0.f;
    if (on) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float d = f[j] - mean; v += d * d; }
    }
    const float rstd = rsqrtf(block_sum<4>(v, scratch) / (float)cols + eps);
    float amax = 0.f;
    if (on) {
        float g[8], bb[8];
        unpack8(ld16(w + vi * 8), g);
        unpack8(ld16(b + vi * 8), bb);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = (f[j] - mean) * rstd * g[j] + bb[j];
        if (res) {
            float r[8];
            unpack8(ld16(res + roff + vi * 8), r);
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = r[j] + bfround(f[j]);
        }
        const auto yb = pack8(f);
        if (y) st16(y + roff + vi * 8, yb);
        unpack8(yb, f);                                  // the bf16 values the quantiser sees
#pragma unroll
        for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(f[j]));
    }
    amax = wave_max(amax);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = amax;
    __syncthreads();
    amax = fmaxf(fmaxf(scratch[0], scratch[1]), fmaxf(scratch[2], scratch[3]));
    const float sc = amax > 0.f ? amax / 448.0f : 1.0f;
    if (threadIdx.x == 0) scale[blockIdx.x] = sc;
    if (on) {
        int lo = 0, hi = 0;
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0] / sc, f[1] / sc, lo, false);
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2] / sc, f[3] / sc, lo, true);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4] / sc, f[5] / sc, hi, false);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6] / sc, f[7] / sc, hi, true);
        *reinterpret_cast<uint2*>(q + roff + vi * 8) = make_uint2((unsigned)lo, (unsigned)hi);
    }
}
// touch [ptr, ptr + bytes) into the infinity cache ahead of the launch that streams it (stand-alone form of common.h::pf_touch)
__global__ __launch_bounds__(256) void prefetch_kernel(PfSpan sp) { pf_touch(sp, blockIdx.x * 256 + threadIdx.x, gridDim.x * 256); }
}  // namespace
int launch_prefetch(const void* ptr, size_t bytes, int workgroups, hipStream_t s) {
    if (!ptr || workgroups < 1 || (bytes >> 7) > 0xffffffffull) return -22;
    PfSpan sp;
    sp.p0 = reinterpret_cast<const char*>(ptr); sp.n0 = (uint32_t)(bytes >> 7);
    hipLaunchKernelGGL(prefetch_kernel, dim3(workgroups), dim3(256), 0, s, sp);
    EMU_CHECK_LAUNCH();
    return 0;
}

int launch_softmax_rows(bf16_t* x, const bf16_t* bias, int rows, int cols, int ld, int ld_bias, float scale, hipStream_t s) {
    if (rows < 1 || (cols & 7) || (ld & 7) || (bias && (ld_bias & 7))) return -22;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(rows), dim3(256), 0, s, x, bias, cols, ld, ld_bias, scale);
    EMU_CHECK_LAUNCH();
    return 0;
}

int launch_rmsnorm(const bf16_t* x, const bf16_t* w, bf16_t* y, int rows, int cols, int ldx, int ldy, float eps, hipStream_t s) {
    if (rows < 1 || (cols & 7) || (ldx & 7) || (ldy & 7)) return -22;
    hipLaunchKernelGGL(rmsnorm_kernel, dim3(rows), dim3(256), 0, s, x, w, y, cols, ldx, ldy, eps);
    EMU_CHECK_LAUNCH();
    return 0;
}
int launch_layernorm(const bf16_t* x, const bf16_t* w, const bf16_t* b, const bf16_t* res, bf16_t* y,
                     int rows, int cols, float eps, hipStream_t s) {
    if (rows < 1 || (cols & 7)) return -22;
    hipLaunchKernelGGL(layernorm_kernel, dim3(rows), dim3(256), 0, s, x, w, b, res, y, cols, eps);
    EMU_CHECK_LAUNCH();
    return 0;
}
int launch_layernorm_q8(const bf16_t* x, const bf16_t* w, const bf16_t* b, const bf16_t* res, bf16_t* y, uint8_t* q, float* scale,
                        int rows, int cols, float eps, hipStream_t s) {
    if (rows < 1 || (cols & 7) || cols > 2048 || !q || !scale) return -22;
    hipLaunchKernelGGL(layernorm_q8_kernel, dim3(rows), dim3(256), 0, s, x, w, b, res, y, q, scale, cols, eps);
    EMU_CHECK_LAUNCH();
    return 0;
}
int launch_embed_gather(const int32_t* ids, const bf16_t* table, bf16_t* out, int n_tok, int hidden, int vocab, hipStream_t s) {
    if (n_tok < 1 || (hidden & 7)) return -22;
    hipLaunchKernelGGL(embed_gather_kernel, dim3(n_tok), dim3(256), 0, s, ids, table, out, hidden, vocab);
    EMU_CHECK_LAUNCH();
    return 0;
}
int launch_scatter_rows(const bf16_t* src, const int32_t* dst_rows, bf16_t* out, int n_rows, int hidden, hipStream_t s) {
    if (n_rows < 1) return 0;
    if (hidden & 7) return -22;
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(n_rows), dim3(256), 0, s, src, dst_rows, out, hidden);
    EMU_CHECK_LAUNCH();
    return 0;
}
int launch_argmax(const bf16_t* logits, int ld, int rows, int vocab, int suppress_id, int32_t* out, hipStream_t s) {
    if (rows < 1 || vocab < 1) return -22;
    hipLaunchKernelGGL(argmax_kernel, dim3(rows), dim3(1024), 0, s, logits, ld, vocab, suppress_id, out);
    EMU_CHECK_LAUNCH();
    return 0;
}
int launch_patchify(const void* image, int image_is_f32, bf16_t* out, int B, int C, int HW, int p, int Kpad, hipStream_t s) {
    if (B < 1 || HW % p || Kpad < C * p * p) return -22;
    const int g = HW / p;
    if (image_is_f32) hipLaunchKernelGGL(patchify_kernel<true>, dim3(B * g * g), dim3(256), 0, s, image, out, C, HW, p, Kpad);
    else hipLaunchKernelGGL(patchify_kernel<false>, dim3(B * g * g), dim3(256), 0, s, image, out, C, HW, p, Kpad);
    EMU_CHECK_LAUNCH();
    return 0;
}
int launch_avgpool_tokens(const bf16_t* x, bf16_t* out, int B, int g, int C, int s, hipStream_t st) {
    if (B < 1 || s < 1 || g % s || (C & 7)) return -22;
    const int go = g / s;
    hipLaunchKernelGGL(avgpool_kernel, dim3(B * go * go), dim3(256), 0, st, x, out, g, C, s);
    EMU_CHECK_LAUNCH();
    return 0;
}
int launch_vit_assemble(const bf16_t* patches, const bf16_t* cls, const bf16_t* pos, bf16_t* x, int B, int T, int C, hipStream_t s) {
    if (B < 1 || (C & 7)) return -22;
    hipLaunchKernelGGL(vit_assemble_kernel, dim3(B * (T + 1)), dim3(256), 0, s, patches, cls, pos, x, T, C);
    EMU_CHECK_LAUNCH();
    return 0;
}
int launch_quant_fp8_rows(const bf16_t* w, int ldw, uint8_t* q, int ldq, float* scale, int N, int K, hipStream_t s) {
    if (N < 1 || K < 8 || (K & 7) || (ldw & 7) || (ldq & 7)) return -22;
    if (K <= 2048) hipLaunchKernelGGL(quant_fp8_rows_kernel<1>, dim3(N), dim3(256), 0, s, w, ldw, q, ldq, scale, K);
    else if (K <= 8192) hipLaunchKernelGGL(quant_fp8_rows_kernel<4>, dim3(N), dim3(256), 0, s, w, ldw, q, ldq, scale, K);
    else if (K <= 16384) hipLaunchKernelGGL(quant_fp8_rows_kernel<8>, dim3(N), dim3(256), 0, s, w, ldw, q, ldq, scale, K);
    else hipLaunchKernelGGL(quant_fp8_rows_kernel<0>, dim3(N), dim3(256), 0, s, w, ldw, q, ldq, scale, K);
    EMU_CHECK_LAUNCH();
    return 0;
}
