// Attention kernels of the Emu2 hot path (gfx950).
//
//  rope_kv      : LLaMA RoPE on q,k (in place, bf16 rounding points of transformers' apply_rotary_pos_emb)
//                 + append of k,v into the [B, H, S_max, D] KV cache.
//  transpose_v  : V[s][d] -> Vt[d][s] (zero padded to a multiple of 64 keys) so that BOTH MFMA operands of
//                 the P.V product are contiguous along the contraction (key) axis.
//  flash_attn   : fused softmax(Q K^T * scale + mask) V for prefill / ViT / UNet shapes, D in {64, 128}.
//                 v_mfma_f32_32x32x16_bf16 with SWAPPED operands (S^T = K Q^T, O^T = Vt P^T): every lane owns
//                 one query column, so row max / row sum / rescale are lane-local (one cross-lane exchange
//                 with lane^32), and P feeds the second MFMA straight from registers.  K rows are loaded
//                 with bits 2<->3 of the MFMA row index swapped so that the accumulator registers
//                 8*ks'..8*ks'+7 of a lane are exactly the 8 consecutive keys its B-fragment needs.
//  decode_attn  : single-query attention over the KV cache, split over the context (HBM-bound), with an
//                 fp32 log-sum-exp combine.
//
// Replaces: transformers LlamaAttention (eager/sdpa) reached from Emu2/emu/emu.py:133-138,213-229;
// Attention.forward naive branch Emu2/emu/eva_vit.py:227-248.
#include "common.h"
#include "kernels.h"

namespace {

// ------------------------------------------------------------------------------------------ rope + kv
__global__ __launch_bounds__(256) void rope_kv_kernel(const RopeKvArgs a) {
    const int row = blockIdx.x;                       // b*T + t
    const int b = row / a.T;
    const int D = a.D, Hl = a.Hl, half = D >> 1;
    const int pos = a.pos[row], slot = a.slot[row];
    bf16_t* q = a.qkv + (size_t)row * 3 * Hl * D;
    bf16_t* k = q + (size_t)Hl * D;
    const bf16_t* v = k + (size_t)Hl * D;
    const bf16_t* cs = a.cos + (size_t)pos * D;
    const bf16_t* sn = a.sin + (size_t)pos * D;
    const int vh = half >> 3;                         // 16-byte vectors per half head
    // items: [0, Hl*vh) rotate q ; [Hl*vh, 2*Hl*vh) rotate k + store ; then v copy
    for (int it = threadIdx.x; it < 2 * Hl * vh; it += 256) {
        const bool isk = it >= Hl * vh;
        const int r = isk ? it - Hl * vh : it;
        const int h = r / vh, vi = r % vh;
        bf16_t* base = (isk ? k : q) + (size_t)h * D + vi * 8;
        float x1[8], x2[8], c[8], s[8], o1[8], o2[8];
        unpack8(ld16(base), x1);
        unpack8(ld16(base + half), x2);
        unpack8(ld16(cs + vi * 8), c);
        unpack8(ld16(sn + vi * 8), s);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            // q_embed = (q * cos) + (rotate_half(q) * sin), every product and the sum rounded to bf16
            o1[j] = bfround(x1[j] * c[j]) + bfround(-x2[j] * s[j]);
            o2[j] = bfround(x2[j] * c[j]) + bfround(x1[j] * s[j]);
        }
        const u32x4 p1 = pack8(o1), p2 = pack8(o2);
        st16(base, p1);
        st16(base + half, p2);
        if (isk) {
            bf16_t* dst = a.kcache + (((size_t)b * Hl + h) * a.S_max + slot) * D + vi * 8;
            st16(dst, p1);
            st16(dst + half, p2);
        }
    }
    const int vd = D >> 3;
    for (int it = threadIdx.x; it < Hl * vd; it += 256) {
        const int h = it / vd, vi = it % vd;
        st16(a.vcache + (((size_t)b * Hl + h) * a.S_max + slot) * D + vi * 8, ld16(v + (size_t)h * D + vi * 8));
    }
}

// ------------------------------------------------------------------------------------------ V transpose
__global__ __launch_bounds__(256) void transpose_v_kernel(const TransposeVArgs a) {
    __shared__ bf16_t tile[64][72];                   // [token][d], padded
    const int s0 = blockIdx.x * 64, d0 = blockIdx.y * 64;
    const int bh = blockIdx.z, b = bh / a.H, h = bh % a.H;
    const bf16_t* src = a.v + (size_t)b * a.v_sb + (size_t)h * a.v_sh;
    for (int q = threadIdx.x; q < 512; q += 256) {    // 64 tokens x 8 vectors
        const int ts = q >> 3, c = q & 7;
        u32x4 val = {0u, 0u, 0u, 0u};
        if (s0 + ts < a.S) val = ld16(src + (size_t)(s0 + ts) * a.v_ss + d0 + c * 8);
        *reinterpret_cast<u32x4*>(&tile[ts][c * 8]) = val;
    }
    __syncthreads();
    bf16_t* dst = a.vt + ((size_t)bh * a.D + d0) * a.S_pad + s0;
    for (int q = threadIdx.x; q < 512; q += 256) {    // 64 d x 8 token-vectors
        const int d = q >> 3, c = q & 7;
        uint32_t w[4];
#pragma unroll
        for (int e = 0; e < 4; ++e)
            w[e] = (uint32_t)tile[c * 8 + 2 * e][d] | ((uint32_t)tile[c * 8 + 2 * e + 1][d] << 16);
        const u32x4 o = {w[0], w[1], w[2], w[3]};
        st16(dst + (size_t)d * a.S_pad + c * 8, o);
    }
}

// ------------------------------------------------------------------------------------------ flash attention
template <int ROWB>                                   // bytes per LDS row: 128 or 256
__device__ __forceinline__ int sw_off(int row, int chunk) {
    if constexpr (ROWB == 256) return row * 256 + ((chunk ^ (row & 15)) << 4);
    else return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}

// NW waves of 32 queries per workgroup share every K / V tile: 4 (128 queries) or 8 (256 queries: two waves per SIMD, so one wave's
// softmax runs beside the other's MFMAs, and half the staging per query -- for launches that would otherwise leave 1.25-1.5 rounds
// of 4-wave workgroups, e.g. the UNet's 32^2 level: 320 workgroups on 256 CUs, 64 CUs with two)
template <int D, int NW>
__global__ __launch_bounds__(NW * 64, NW == 8 ? 2 : 2) void flash_kernel(const FlashArgs a) {
    constexpr int THREADS = NW * 64, QB = NW * 32;
    constexpr int KROWB = D * 2;                      // K tile [64 keys][D]
    constexpr int KCH = D / 8;                        // 16-byte chunks per K row
    constexpr int NKK = D / 16;                       // k-steps of QK^T
    constexpr int NDB = D / 32;                       // 32-wide d blocks of O
    constexpr int KT_BYTES = 64 * KROWB;
    constexpr int VT_BYTES = D * 128;                 // Vt tile [D][64 keys]
    constexpr int NLD = (KT_BYTES / 16) / THREADS;    // 16-byte chunks per thread per tile
    constexpr int SMEM_BYTES = (KT_BYTES + VT_BYTES) > QB * D * 2 ? (KT_BYTES + VT_BYTES) : QB * D * 2;   // K / V tiles, later the O tile
    __shared__ __attribute__((aligned(16))) char smem[SMEM_BYTES];
    char* sK = smem;
    char* sV = smem + KT_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    // Workgroup -> (batch, head, query block).  Block i runs on XCD i % 8 (observed dispatch order): dealt straight, the query
    // blocks of ONE head land on 8 different XCDs and every XCD's L2 has to fetch the K / V of every head (UNet 32^2 level: 40 heads
    // x 256 KB = 10 MB per 4 MB L2: 85 MB fetched for 15.7 MB of operands, every key tile a fabric round trip).  Remapped so that an
    // XCD owns a run of consecutive (head, query block) pairs -- whole heads where the counts allow: K / V cross the fabric once.
    const int nq = (a.Sq + QB - 1) / QB, total = nq * a.H * a.B;
    int lin = blockIdx.x;
    if (a.heavy_first) {
        // causal prefill: a query block's work grows with its index (2 .. 13 key tiles at S = 770), and a long block that starts
        // late IS the kernel's tail.  Every XCD owns whole heads (H * B heads dealt 7,7,..,6,6 over the 8 XCDs; the grid is padded
        // to 8 x ceil(heads / 8) x nq blocks, surplus blocks leave at once) and walks them query-block-major from the LAST block
        // down: its 32 CUs start on the long blocks, the short ones fill in behind.  K / V of a head still stay in one L2.
        const int xcd = lin & 7, j = lin >> 3, heads = a.H * a.B, hq = heads >> 3, hr = heads & 7;
        const int mine = hq + (xcd < hr ? 1 : 0), first = xcd * hq + (xcd < hr ? xcd : hr);
        if (j >= mine * nq) return;
        lin = (first + j % mine) * nq + (nq - 1 - j / mine);
    } else if (a.xcd_remap) {
        const int xcd = lin & 7, q8 = total >> 3, r8 = total & 7;
        lin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (lin >> 3);
    }
    const int qb = lin % nq, bh = lin / nq;
    const int b = bh / a.H, h = bh - b * a.H;
    const int qblk = qb * QB;
    const int q0 = qblk + wave * 32;
    const int off = a.Sk - a.Sq;                      // causal: query i sees keys <= i + off
    const int kstart = a.kstart ? a.kstart[b] : 0;

    // Q fragments (B operand): column = query l31, k = 16*kk + 8*hi + j
    int qi = q0 + l31;
    const int qrow = qi < a.Sq ? qi : a.Sq - 1;
    const bf16_t* qp = a.q + (size_t)b * a.q_sb + (size_t)h * a.q_sh + (size_t)qrow * a.q_ss + 8 * hi;
    bf16x8_t qf[NKK];
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) qf[kk] = *reinterpret_cast<const bf16x8_t*>(qp + 16 * kk);

    const bf16_t* kbase = a.k + (size_t)b * a.k_sb + (size_t)h * a.k_sh;
    const bf16_t* vbase = a.vt + ((size_t)(b * a.H + h) * D) * a.Sk_pad;

    int last_key = a.Sk - 1;
    if (a.causal) {
        int qmax = qblk + QB - 1; qmax = qmax < a.Sq ? qmax : a.Sq - 1;
        last_key = qmax + off < last_key ? qmax + off : last_key;
    }
    const int ntile = last_key < 0 ? 0 : last_key / 64 + 1;

    u32x4 rk[NLD], rv[NLD];
    auto gload = [&](int t) {
        const int kt0 = t * 64;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int q = tid + THREADS * i;
            { const int row = q / KCH, c = q % KCH;
              int key = kt0 + row; key = key < a.Sk ? key : a.Sk - 1;
              rk[i] = ld16(kbase + (size_t)key * a.k_ss + c * 8); }
            { const int row = q >> 3, c = q & 7;
              rv[i] = ld16(vbase + (size_t)row * a.Sk_pad + kt0 + c * 8); }
        }
    };
    auto sstore = [&]() {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int q = tid + THREADS * i;
            st16(sK + sw_off<KROWB>(q / KCH, q % KCH), rk[i]);
            st16(sV + sw_off<128>(q >> 3, q & 7), rv[i]);
        }
    };

    f32x16_t o[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const float sc = a.scale * 1.4426950408889634f;   // softmax in the exp2 domain

    // MFMA row i of the K operand holds key pi(i) = i with bits 2 and 3 swapped
    const int krow = (l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);

    if (ntile > 0) gload(0);
    for (int t = 0; t < ntile; ++t) {
        __syncthreads();                               // previous tile fully consumed
        sstore();
        __syncthreads();
        if (t + 1 < ntile) gload(t + 1);               // overlap next tile's HBM latency with the MFMAs
        const int kt0 = t * 64;

        f32x16_t s[2];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[blk][r] = 0.f;
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) {
                const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(sK + sw_off<KROWB>(blk * 32 + krow, kk * 2 + hi));
                s[blk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[kk], s[blk], 0, 0, 0);
            }
        }
        // lane (query qi) holds keys kt0 + 32*blk + 16*(r>>3) + 8*hi + (r&7).  The softmax is the VALU-heavy part of a
        // tile (it outweighs the 16 MFMAs at D = 64), so: interior tiles skip the mask arithmetic (wave-uniform test),
        // the scale rides in the exponent's fma, exp2 is the bare v_exp_f32, and O is rescaled only when some row's
        // maximum moved.
        const bool interior = kt0 + 63 < a.Sk && kt0 >= kstart && (!a.causal || kt0 + 63 <= q0 + off);
        float mloc = -INFINITY;
        if (interior) {
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, s[blk][r]);
        } else {
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kt0 + 32 * blk + 16 * (r >> 3) + 8 * hi + (r & 7);
                    const bool ok = key < a.Sk && key >= kstart && (!a.causal || key <= qi + off);
                    const float v = ok ? s[blk][r] : -INFINITY;
                    s[blk][r] = v;
                    mloc = fmaxf(mloc, v);
                }
        }
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64)) * sc;          // sc > 0: the maximum commutes with the scale
        const float mnew = fmaxf(m_run, mloc);
        float alpha = 1.f;
        const bool dead = (mnew == -INFINITY);         // every key so far masked for this query
        if (!dead) alpha = __builtin_amdgcn_exp2f(m_run - mnew);    // m_run = -inf -> 0
        const float nm = dead ? 0.f : -mnew;
        float psum = 0.f;
        u32x4 pfw[4];                                  // P as packed bf16 pairs = the B fragments of the PV MFMAs
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                // masked scores are -inf: fma(-inf, sc, nm) = -inf -> exp2 = 0 (dead rows: nm = 0, still 0)
                const float p0 = __builtin_amdgcn_exp2f(fmaf(s[blk][r], sc, nm));
                const float p1 = __builtin_amdgcn_exp2f(fmaf(s[blk][r + 1], sc, nm));
                psum += p0 + p1;
                pfw[blk * 2 + (r >> 3)][(r & 7) >> 1] = packbf(p0, p1);
            }
        l_run = l_run * alpha + psum;
        m_run = mnew;
        if (__any(alpha != 1.f)) {
#pragma unroll
            for (int db = 0; db < NDB; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int db = 0; db < NDB; ++db) {
                const bf16x8_t vf = *reinterpret_cast<const bf16x8_t*>(sV + sw_off<128>(db * 32 + l31, ks * 2 + hi));
                o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, __builtin_bit_cast(bf16x8_t, pfw[ks]), o[db], 0, 0, 0);
            }
    }

    const float ltot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = ltot > 0.f ? 1.f / ltot : 0.f;
    if (a.stage_o) {
        // O leaves through LDS (the K / V tiles are dead): in the accumulator layout a store instruction writes 8 bytes to each of
        // 32 query rows (32 cache lines touched); staged, 16 bytes per lane and whole rows of D values (gemm_tile.h::EpiStage)
        constexpr int ROWB = D * 2, SLOTS = ROWB / 16, KEYM = SLOTS - 1;
        constexpr int O_BYTES = QB * ROWB;             // (8 waves: twice the K / V tiles' LDS -- declared below)
        __syncthreads();                               // every wave is done with the last K / V tile
        const int row = wave * 32 + l31;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u32x2 ov;
                ov.x = packbf(o[db][4 * g] * inv, o[db][4 * g + 1] * inv);
                ov.y = packbf(o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv);
                const int col = db * 32 + 8 * g + 4 * hi;
                *reinterpret_cast<u32x2*>(smem + row * ROWB + ((((col >> 3) ^ row) & KEYM) << 4) + ((col & 7) << 1)) = ov;
            }
        __syncthreads();
        bf16_t* ob = a.o + (size_t)b * a.o_sb + (size_t)h * a.o_sh;
        static_assert(O_BYTES <= SMEM_BYTES, "the O tile fits");
#pragma unroll
        for (int r = 0; r < (QB * SLOTS) / THREADS; ++r) {
            const int idx = r * THREADS + tid, rw = idx / SLOTS, ps = idx % SLOTS, ls = (ps ^ rw) & KEYM;
            const u32x4 v = *reinterpret_cast<const u32x4*>(smem + idx * 16);
            if (qblk + rw < a.Sq) *reinterpret_cast<u32x4*>(ob + (size_t)(qblk + rw) * a.o_ss + ls * 8) = v;
        }
        return;
    }
    if (qi < a.Sq) {
        bf16_t* op = a.o + (size_t)b * a.o_sb + (size_t)h * a.o_sh + (size_t)qi * a.o_ss;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u32x2 ov;
                ov.x = packbf(o[db][4 * g] * inv, o[db][4 * g + 1] * inv);
                ov.y = packbf(o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv);
                *reinterpret_cast<u32x2*>(op + db * 32 + 8 * g + 4 * hi) = ov;
            }
    }
}

// ------------------------------------------------------------------------------------------ decode attention
constexpr int DEC_CHUNK = 256;                         // keys per workgroup

template <int D>
__global__ __launch_bounds__(256) void decode_attn_kernel(const DecodeAttnArgs a, int nsplit) {
    constexpr int VD = D / 8;                          // 16-byte vectors per row
    constexpr int NG = 256 / VD;                       // key groups in the P.V phase
    __shared__ float sq[D];
    __shared__ float sp[DEC_CHUNK];
    __shared__ float red[NG][D + 1];
    __shared__ float scratch[4];
    const int split = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int tid = threadIdx.x;
    const int k0 = split * DEC_CHUNK;
    const int ctx = a.ctx_ptr ? *a.ctx_ptr : a.ctx;
    const int nk = min(DEC_CHUNK, ctx - k0);          // <= 0 for splits beyond the live context
    const int kstart = a.kstart ? a.kstart[b] : 0;
    if (tid < D) sq[tid] = bf2f(a.q[(size_t)b * a.q_sb + (size_t)h * a.q_sh + tid]);
    __syncthreads();
    const size_t rowbase = ((size_t)b * a.H + h) * a.S_max;
    const bf16_t* kc = a.kcache + rowbase * D;
    const bf16_t* vc = a.vcache + rowbase * D;

    float sc = -INFINITY;
    if (tid < nk && (k0 + tid) >= kstart) {
        const bf16_t* kr = kc + (size_t)(k0 + tid) * D;
        float acc = 0.f;
#pragma unroll
        for (int vi = 0; vi < VD; ++vi) {
            float f[8];
            unpack8(ld16(kr + vi * 8), f);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc = fmaf(f[j], sq[vi * 8 + j], acc);
        }
        sc = acc * a.scale;
    }
    // block max
    float m = wave_max(sc);
    __syncthreads();
    if ((tid & 63) == 0) scratch[tid >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(scratch[0], scratch[1]), fmaxf(scratch[2], scratch[3]));
    const float p = (sc == -INFINITY) ? 0.f : __expf(sc - m);
    sp[tid] = p;
    const float l = block_sum<4>(p, scratch);          // contains the barriers that publish sp[]

    const int dv = tid % VD, kg = tid / VD;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int j = kg; j < nk; j += NG) {
        float f[8];
        unpack8(ld16(vc + (size_t)(k0 + j) * D + dv * 8), f);
        const float pj = sp[j];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = fmaf(pj, f[e], acc[e]);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[kg][dv * 8 + e] = acc[e];
    __syncthreads();
    float* w = a.ws + (((size_t)b * a.H + h) * nsplit + split) * (D + 2);
    if (tid < D) {
        float t = 0.f;
#pragma unroll 4
        for (int g = 0; g < NG; ++g) t += red[g][tid];
        w[tid] = t;
    }
    if (tid == 0) { w[D] = m; w[D + 1] = l; }
}

template <int D>
__global__ void decode_combine_kernel(const DecodeAttnArgs a, int nsplit) {
    const int h = blockIdx.x, b = blockIdx.y, d = threadIdx.x;
    const float* w = a.ws + ((size_t)b * a.H + h) * nsplit * (D + 2);
    float m = -INFINITY;
    for (int s = 0; s < nsplit; ++s) m = fmaxf(m, w[s * (D + 2) + D]);
    float num = 0.f, den = 0.f;
    for (int s = 0; s < nsplit; ++s) {
        const float ms = w[s * (D + 2) + D];
        const float f = (ms == -INFINITY) ? 0.f : __expf(ms - m);
        num += f * w[s * (D + 2) + d];
        den += f * w[s * (D + 2) + D + 1];
    }
    a.o[(size_t)b * a.o_sb + (size_t)h * a.o_sh + d] = f2bf(den > 0.f ? num / den : 0.f);
}

}  // namespace

int launch_rope_kv(const RopeKvArgs& a, hipStream_t s) {
    if (a.B < 1 || a.T < 1 || (a.D & 15)) return -22;
    hipLaunchKernelGGL(rope_kv_kernel, dim3(a.B * a.T), dim3(256), 0, s, a);
    EMU_CHECK_LAUNCH();
    return 0;
}

int launch_transpose_v(const TransposeVArgs& a, hipStream_t s) {
    if ((a.D & 63) || (a.S_pad & 63) || a.S_pad < a.S || (a.v_ss & 7) || (a.v_sh & 7) || (a.v_sb & 7)) return -22;
    hipLaunchKernelGGL(transpose_v_kernel, dim3(a.S_pad / 64, a.D / 64, a.B * a.H), dim3(256), 0, s, a);
    EMU_CHECK_LAUNCH();
    return 0;
}

int launch_flash_attn(const FlashArgs& a, hipStream_t s) {
    if (a.Sq < 1 || a.Sk < 1 || (a.Sk_pad & 63) || a.Sk_pad < a.Sk) return -22;
    if ((a.q_ss & 7) || (a.k_ss & 7) || (a.o_ss & 3) || (a.q_sh & 7) || (a.k_sh & 7) || (a.o_sh & 3) ||
        (a.q_sb & 7) || (a.k_sb & 7) || (a.o_sb & 3)) return -22;
    FlashArgs b = a;
    const int tune = emu_gemm_tune_get();              // A/B switches: bit 6 = straight block order, bit 7 = direct O stores,
    b.xcd_remap = !(tune & 64);                        //               bits 12-13 = 1: always 4 waves, 2: always 8 waves
    b.stage_o = !(tune & 128) && !((a.o_ss | a.o_sh | a.o_sb) & 7) && !((uintptr_t)a.o & 15);
    if (a.D != 128 && a.D != 64) return -22;
    // 256-query workgroups where 128-query ones would leave between one and one and a half rounds (D = 64 only: at D = 128 the
    // kernel holds 242 registers, two waves per SIMD do not fit twice)
    const int wg4 = ((a.Sq + 127) / 128) * a.H * a.B;
    bool w8 = a.D == 64 && wg4 > 256 && wg4 <= 400 && a.Sq >= 256;
    if (((tune >> 12) & 3) == 1) w8 = false;
    if (((tune >> 12) & 3) == 2) w8 = a.D == 64;
    // causal launches of more than one round's worth of unequal blocks: longest query blocks first (flash_kernel: heavy_first)
    b.heavy_first = a.causal && !w8 && a.Sq >= 256 && wg4 > 256 && !(tune & (1 << 15));
    const int nqb = (a.Sq + 127) / 128;
    const dim3 grid(w8 ? ((a.Sq + 255) / 256) * a.H * a.B : (b.heavy_first ? 8 * ((a.H * a.B + 7) / 8) * nqb : wg4)), block(w8 ? 512 : 256);
    const bool prof = emu_prof_on();
    if (prof) emu_prof_begin(s);
    if (a.D == 128) hipLaunchKernelGGL((flash_kernel<128, 4>), grid, block, 0, s, b);
    else if (w8) hipLaunchKernelGGL((flash_kernel<64, 8>), grid, block, 0, s, b);
    else hipLaunchKernelGGL((flash_kernel<64, 4>), grid, block, 0, s, b);
    if (prof) emu_prof_end(s, "attn", a.Sq, a.Sk, a.D, a.B * a.H, (a.causal ? 2.0 : 4.0) * a.B * a.H * (double)a.Sq * a.Sk * a.D);
    EMU_CHECK_LAUNCH();
    return 0;
}

int decode_attn_nsplit(int ctx) { return ctx < 1 ? 1 : (ctx + DEC_CHUNK - 1) / DEC_CHUNK; }

namespace {
__global__ void greedy_advance_kernel(const int32_t* cur_ids, int32_t* pos, int32_t* slot, int32_t* ctx,
                                      int32_t* step, int32_t* out_ids, int B) {
    const int b = threadIdx.x;
    const int st = *step;
    if (b < B) {
        out_ids[(size_t)st * B + b] = cur_ids[b];
        pos[b] += 1;
        slot[b] += 1;
    }
    __syncthreads();
    if (b == 0) { *ctx += 1; *step = st + 1; }
}

// generate_image's loop state on the device (emu.py:92-153, KV-cached form): the step's output rows [B, cols] go to out_all[step]
// and to `prev` (the next step's regression input), positions / slots / the step counter advance
__global__ void regress_advance_kernel(const bf16_t* src, bf16_t* out_all, bf16_t* prev, int32_t* pos, int32_t* slot, int32_t* step,
                                       int B, int cols) {
    const int st = *step;
    for (int i = threadIdx.x; i < B * cols; i += blockDim.x) {
        const bf16_t v = src[i];
        out_all[(size_t)st * B * cols + i] = v;
        prev[i] = v;
    }
    if ((int)threadIdx.x < B) { pos[threadIdx.x] += 1; slot[threadIdx.x] += 1; }
    __syncthreads();
    if (threadIdx.x == 0) *step = st + 1;
}

}  // namespace

int launch_regress_advance(const bf16_t* src, bf16_t* out_all, bf16_t* prev, int32_t* pos, int32_t* slot, int32_t* step, int B,
                           int cols, hipStream_t s) {
    if (B < 1 || B > 256 || cols < 1) return -22;
    hipLaunchKernelGGL(regress_advance_kernel, dim3(1), dim3(256), 0, s, src, out_all, prev, pos, slot, step, B, cols);
    EMU_CHECK_LAUNCH();
    return 0;
}

int launch_greedy_advance(const int32_t* cur_ids, int32_t* pos, int32_t* slot, int32_t* ctx, int32_t* step,
                          int32_t* out_ids, int B, hipStream_t s) {
    if (B < 1 || B > 64) return -22;
    hipLaunchKernelGGL(greedy_advance_kernel, dim3(1), dim3(64), 0, s, cur_ids, pos, slot, ctx, step, out_ids, B);
    EMU_CHECK_LAUNCH();
    return 0;
}

namespace {
constexpr int DF_CHUNK = 128;                          // keys per workgroup (32 per wave)

// One 128-key split of SHARED slots (beam search: the prompt's keys, stored in the group's first row b0 only) against the
// queries of all the group's beams (<= NB): the K / V rows are loaded once and stay in registers, every beam's split state is
// formed with the arithmetic and the summation order of the one-row path below (so a shared cache gives the bits of
// replicated ones), the beams' states meet in LDS behind ONE barrier.  No append here: the new token's slot is never shared.
template <int D, int NB>
__device__ __forceinline__ void decode_shared_split(const DecodeFusedArgs& a, int nsplit, int split, int h, int b0) {
    constexpr int LPK = D / 8, KPI = 64 / LPK, ITER = (DF_CHUNK / 4) / KPI, NP = 4 * KPI;
    __shared__ float st[NB][NP][D + 2];
    __shared__ float qs[NB][D];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int g = lane / LPK, dl = lane % LPK, d0 = dl * 8;
    const int nb = a.share_nb, k0 = split * DF_CHUNK;
    const int ctx = a.slot[b0] + 1, kstart = a.kstart ? a.kstart[b0] : 0;
    const size_t hb = ((size_t)b0 * a.H + h) * a.S_max;
    const bf16_t* kc = a.kcache + hb * D;
    const bf16_t* vc = a.vcache + hb * D;
    constexpr int half = D / 2;
    // rotated query of beam tid / LPK, slice tid % LPK (loads first, like everything else here)
    const bool qlane = tid < nb * LPK;
    const int qb = qlane ? tid / LPK : 0, qd0 = (tid % LPK) * 8;
    const int qdp = (qd0 + half) % D, qdc = qd0 % half;
    const bf16_t* qh = a.qkv + (size_t)(b0 + qb) * 3 * a.H * D + (size_t)h * D;
    const int qpos = a.pos[b0 + qb];
    const u32x4 cv = ld16(a.cos + (size_t)qpos * D + qdc), sv = ld16(a.sin + (size_t)qpos * D + qdc);
    const u32x4 q1v = ld16(qh + qd0), q2v = ld16(qh + qdp);
    u32x4 kr[ITER], vr[ITER];
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const int key = k0 + wave * (DF_CHUNK / 4) + it * KPI + g;           // < share_len <= S_max
        kr[it] = ld16(kc + (size_t)key * D + d0);
        vr[it] = ld16(vc + (size_t)key * D + d0);
    }
    if (qlane) {
        float c[8], sn[8], x1[8], x2[8];
        unpack8(cv, c); unpack8(sv, sn); unpack8(q1v, x1); unpack8(q2v, x2);
        const float sgn = qd0 < half ? -1.f : 1.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) qs[qb][qd0 + j] = bfround(bfround(x1[j] * c[j]) + bfround(sgn * x2[j] * sn[j]));
    }
    __syncthreads();
    float q[NB][8];
#pragma unroll
    for (int jb = 0; jb < NB; ++jb)
#pragma unroll
        for (int j = 0; j < 8; ++j) q[jb][j] = jb < nb ? qs[jb][d0 + j] : 0.f;
    float sd[NB][ITER], m[NB];
#pragma unroll
    for (int jb = 0; jb < NB; ++jb) m[jb] = -INFINITY;
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const int key = k0 + wave * (DF_CHUNK / 4) + it * KPI + g;
        const bool valid = key < ctx && key >= kstart;
        float kf[8];
        unpack8(kr[it], kf);
#pragma unroll
        for (int j = 0; j < 8; ++j) kf[j] = valid ? kf[j] : 0.f;
#pragma unroll
        for (int jb = 0; jb < NB; ++jb) {
            float t = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) t = fmaf(kf[j], q[jb][j], t);
            t = LPK == 16 ? row16_sum(t) : row8_sum(t);
            sd[jb][it] = valid ? t * a.scale : -INFINITY;
            m[jb] = fmaxf(m[jb], sd[jb][it]);
        }
    }
    float l[NB], acc[NB][8];
#pragma unroll
    for (int jb = 0; jb < NB; ++jb) {
        l[jb] = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[jb][j] = 0.f;
    }
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        float vf[8];
        unpack8(vr[it], vf);
        const bool dead = sd[0][it] == -INFINITY;                             // validity is the same for every beam
#pragma unroll
        for (int j = 0; j < 8; ++j) vf[j] = dead ? 0.f : vf[j];
#pragma unroll
        for (int jb = 0; jb < NB; ++jb) {
            const float p = dead ? 0.f : __expf(sd[jb][it] - m[jb]);
            l[jb] += p;
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[jb][j] = fmaf(p, vf[j], acc[jb][j]);
        }
    }
#pragma unroll
    for (int jb = 0; jb < NB; ++jb) {
        float* dst = st[jb][wave * KPI + g];
#pragma unroll
        for (int j = 0; j < 8; ++j) dst[d0 + j] = acc[jb][j];
        if (dl == 0) { dst[D] = m[jb]; dst[D + 1] = l[jb]; }
    }
    __syncthreads();
    for (int idx = tid; idx < nb * D; idx += 256) {
        const int jb = idx / D, d = idx - jb * D;
        float num = 0.f, den = 0.f, mt = -INFINITY;
#pragma unroll
        for (int w = 0; w < NP; ++w) mt = fmaxf(mt, st[jb][w][D]);
#pragma unroll
        for (int w = 0; w < NP; ++w) {
            const float f = (st[jb][w][D] == -INFINITY) ? 0.f : __expf(st[jb][w][D] - mt);
            num = fmaf(f, st[jb][w][d], num);
            den = fmaf(f, st[jb][w][D + 1], den);
        }
        float* wout = a.ws + (((size_t)(b0 + jb) * a.H + h) * nsplit + split) * (D + 2);
        wout[d] = num;
        if (d == 0) { wout[D] = mt; wout[D + 1] = den; }
    }
}

// NB > 0: the rows are beam groups (a.share_nb in 2..NB) whose shared slots (the prompt) are stored in the group's first row
// only (DecodeFusedArgs).  The launch is one-dimensional: first one workgroup per (shared split, head, GROUP), then one per
// (other split, head, row) as in the one-row form, whose split that straddles the end of the prompt takes its shared keys
// from the group's first row.  A separate instantiation, so the one-row launches (greedy decode: the headline path) keep
// their registers and occupancy.
template <int D, int NB>
__global__ __launch_bounds__(256) void decode_fused_kernel(const DecodeFusedArgs a, int nsplit) {
    constexpr bool SHARE = NB > 0;
    constexpr int LPK = D / 8;                         // lanes per key row (16 bytes each): 16 (D=128) or 8 (D=64)
    constexpr int KPI = 64 / LPK;                      // keys per wave iteration
    constexpr int ITER = (DF_CHUNK / 4) / KPI;
    constexpr int NP = 4 * KPI;                        // partial states per block: (wave, key group)
    __shared__ float sm[NP][D + 2];
    int split = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    if constexpr (SHARE) {
        const int nsh = a.share_len / DF_CHUNK, ngrp = a.B / a.share_nb;       // whole splits of shared slots
        int L = blockIdx.x;
        if (L < nsh * a.H * ngrp) {
            decode_shared_split<D, NB>(a, nsplit, L % nsh, (L / nsh) % a.H, (L / (nsh * a.H)) * a.share_nb);
            return;
        }
        L -= nsh * a.H * ngrp;
        const int nrest = nsplit - nsh;
        split = nsh + L % nrest;
        h = (L / nrest) % a.H;
        b = L / (nrest * a.H);
    }
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int g = lane / LPK, dl = lane % LPK, d0 = dl * 8;
    // slot, position and left-padding bound are requested together: the early exit below needs the slot only, but a load behind the
    // branch would be a second dependent trip in front of the cos / sin rows (round 6: three trips -> two)
    const int slot = a.slot[b], pos = a.pos[b];
    const int kstart = a.kstart ? a.kstart[b] : 0;
    const int ctx = slot + 1;
    const int k0 = split * DF_CHUNK;
    if (k0 >= ctx) return;                             // split beyond the live context: no work, not counted
    // beams of one prompt: slots [0, share_len) live in the group's first row b0 only
    const int b0 = SHARE ? b - b % a.share_nb : b;
    const int nshare = SHARE ? a.share_len : 0;
    const bf16_t* row = a.qkv + (size_t)b * 3 * a.H * D;
    const bf16_t* qh = row + (size_t)h * D;
    const bf16_t* kh = qh + (size_t)a.H * D;
    const bf16_t* vh = kh + (size_t)a.H * D;
    constexpr int half = D / 2;
    const int dp = (d0 + half) % D, dc = d0 % half;
    const float sgn = d0 < half ? -1.f : 1.f;
    const size_t hb = ((size_t)b * a.H + h) * a.S_max;
    const bf16_t* kc = a.kcache + hb * D;
    const bf16_t* vc = a.vcache + hb * D;
    const long to_b0 = -(long)(b - b0) * a.H * a.S_max * D;        // from this row's cache to the group's first row

    // ---- every load of the block is issued before anything is consumed: RoPE inputs first (the in-order vmcnt lets
    // the rotation start while the K/V rows are still in flight), then this wave's 32 keys (LPK lanes cover one row)
    const u32x4 cv = ld16(a.cos + (size_t)pos * D + dc), sv = ld16(a.sin + (size_t)pos * D + dc);
    const u32x4 q1v = ld16(qh + d0), q2v = ld16(qh + dp), k1v = ld16(kh + d0), k2v = ld16(kh + dp), nvv = ld16(vh + d0);
    // (unconditional, index clamped into the cache: a predicated load becomes a branch whose merge copy makes the
    // compiler drain vmcnt after the first pair; masked keys are zeroed by select below, never multiplied)
    u32x4 kr[ITER], vr[ITER];
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const int key = k0 + wave * (DF_CHUNK / 4) + it * KPI + g;
        const int kc_i = key < a.S_max ? key : a.S_max - 1;
        const long off = (long)kc_i * D + d0 + (SHARE && key < nshare ? to_b0 : 0);   // shared slots: the group's first row
        kr[it] = ld16(kc + off);
        vr[it] = ld16(vc + off);
    }
    // ---- RoPE of q and of the new key (this lane's 8-wide slice), transformers' rounding points
    float c[8], sn[8], q[8], nk[8], nv[8];
    unpack8(cv, c);
    unpack8(sv, sn);
    {
        float x1[8], x2[8];
        unpack8(q1v, x1); unpack8(q2v, x2);
#pragma unroll
        for (int j = 0; j < 8; ++j) q[j] = bfround(bfround(x1[j] * c[j]) + bfround(sgn * x2[j] * sn[j]));
        unpack8(k1v, x1); unpack8(k2v, x2);
#pragma unroll
        for (int j = 0; j < 8; ++j) nk[j] = bfround(bfround(x1[j] * c[j]) + bfround(sgn * x2[j] * sn[j]));
    }
    unpack8(nvv, nv);
    if (split == slot / DF_CHUNK && wave == 0 && g == 0) {       // append the new token to the cache (once)
        st16(a.kcache + (hb + slot) * D + d0, pack8(nk));
        st16(a.vcache + (hb + slot) * D + d0, pack8(nv));
    }
    // ---- scores of this lane group's ITER keys (DPP row reductions, all independent), then a two-pass softmax in
    // registers: no running max / rescale chain
    float sd[ITER];
    float m = -INFINITY;
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const int key = k0 + wave * (DF_CHUNK / 4) + it * KPI + g;
        const bool valid = key < ctx && key >= kstart;
        float kf[8];
        unpack8(kr[it], kf);
#pragma unroll
        for (int j = 0; j < 8; ++j) kf[j] = key == slot ? nk[j] : (valid ? kf[j] : 0.f);
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) t = fmaf(kf[j], q[j], t);
        t = LPK == 16 ? row16_sum(t) : row8_sum(t);
        sd[it] = valid ? t * a.scale : -INFINITY;
        m = fmaxf(m, sd[it]);
    }
    float l = 0.f, acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const int key = k0 + wave * (DF_CHUNK / 4) + it * KPI + g;
        float vf[8];
        unpack8(vr[it], vf);
#pragma unroll
        for (int j = 0; j < 8; ++j) vf[j] = key == slot ? nv[j] : (sd[it] == -INFINITY ? 0.f : vf[j]);
        const float p = sd[it] == -INFINITY ? 0.f : __expf(sd[it] - m);
        l += p;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = fmaf(p, vf[j], acc[j]);
    }
    // ---- the NP (wave, key group) states meet in LDS; D threads merge them
    {
        float* dst = sm[wave * KPI + g];
#pragma unroll
        for (int j = 0; j < 8; ++j) dst[d0 + j] = acc[j];
        if (dl == 0) { dst[D] = m; dst[D + 1] = l; }
    }
    __syncthreads();
    float num = 0.f, den = 0.f, mt = -INFINITY;
    if (tid < D) {
#pragma unroll
        for (int w = 0; w < NP; ++w) mt = fmaxf(mt, sm[w][D]);
#pragma unroll
        for (int w = 0; w < NP; ++w) {
            const float f = (sm[w][D] == -INFINITY) ? 0.f : __expf(sm[w][D] - mt);
            num = fmaf(f, sm[w][tid], num);
            den = fmaf(f, sm[w][D + 1], den);
        }
    }
    // ---- publish the split's state; decode_fused_combine_kernel merges the live splits.  Measured alternatives: an
    // in-kernel last-arriver combine needs device-scope fences (L2 write-backs across XCDs): 3.4x slower per layer; one
    // 16-wave workgroup per head walking the whole context (no partials, one launch): +3.7 us per layer at 52 heads x
    // 800 keys, because 52 CUs pull 0.4 MB each at the per-CU L1 rate while 204 CUs idle.
    //
    // Round 4: the last-arriver merge WITHOUT fences (DecodeFusedArgs::arrive).  What made round 1's attempt slow was the
    // release fence in front of the arrival counter: at agent scope it writes the whole L2's dirty lines back (buffer_wbl2), 364
    // times per layer.  Here the split state itself is stored with agent-scope (write-through, sc1) stores, the workgroup waits
    // for their acknowledgement (vmcnt(0)) and bumps a relaxed agent-scope counter; the workgroup that finds the head's last
    // count reads the states back with agent-scope loads (which do not trust this XCD's L2) and merges them with the combine
    // kernel's arithmetic.  No cache-wide operation anywhere, no waiting (nobody spins: every workgroup but the last simply
    // leaves), one launch less per layer.  Stress-tested against the two-launch form bit for bit (tests/test_gpu_model.py) --
    // and 0.4 % SLOWER per token (94.9 / 94.5 vs 95.3 / 94.9 tokens/s, profiles/r04_decode_tail_merge_ab.log): store
    // acknowledgement, counter round trip and the coherent loads are three dependent trips to the fabric, which is what a launch
    // costs inside a graph.  Off by default (emu_llama_set_decode_tail).
    float* wout = a.ws + (((size_t)b * a.H + h) * nsplit + split) * (D + 2);
    if constexpr (!SHARE) {
        if (a.arrive) {
            __shared__ int s_last;
            if (tid < D) {
                __hip_atomic_store(wout + tid, num, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (tid == 0) {
                    __hip_atomic_store(wout + D, mt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(wout + D + 1, den, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's state has reached the agent's point of coherence
            __syncthreads();
            const int nlive = (slot + DF_CHUNK) / DF_CHUNK;
            if (tid == 0) {
                int* cnt = a.arrive + (size_t)b * a.H + h;
                const int old = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_last = old == nlive - 1;
                if (s_last) __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the next launch
            }
            __syncthreads();
            if (!s_last || tid >= D) return;
            const float* w = a.ws + ((size_t)b * a.H + h) * nsplit * (D + 2);
            auto ldc = [](const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
            float m2 = -INFINITY;
            for (int s2 = 0; s2 < nlive; ++s2) m2 = fmaxf(m2, ldc(w + s2 * (D + 2) + D));
            float num2 = 0.f, den2 = 0.f;
#pragma unroll 4
            for (int s2 = 0; s2 < nlive; ++s2) {
                const float ms = ldc(w + s2 * (D + 2) + D);
                const float f = (ms == -INFINITY) ? 0.f : __expf(ms - m2);
                num2 = fmaf(f, ldc(w + s2 * (D + 2) + tid), num2);
                den2 = fmaf(f, ldc(w + s2 * (D + 2) + D + 1), den2);
            }
            a.o[(size_t)b * a.o_sb + (size_t)h * a.o_sh + tid] = f2bf(den2 > 0.f ? num2 / den2 : 0.f);
            return;
        }
    }
    if (tid < D) {
        wout[tid] = num;
        if (tid == 0) { wout[D] = mt; wout[D + 1] = den; }
    }
}

// NS > 0 (round 6): the launch is pure latency -- slot -> live count -> maxima -> states were three dependent trips -- so every split
// state the launch was SIZED for (nsplit <= NS) is requested up front together with the slot, and the live count only selects among
// values that are already on their way (a dead split's words are stale or never written: loaded, never used).  Same arithmetic in the
// same order as the loop form (NS = 0: more splits than the unrolled forms hold), bit-identical.
template <int D, int NS>
__global__ void decode_fused_combine_kernel(const float* ws, const int32_t* slot, bf16_t* o, long o_sb, long o_sh, int H,
                                            int nsplit) {
    const int h = blockIdx.x, b = blockIdx.y, d = threadIdx.x;
    const float* w = ws + ((size_t)b * H + h) * nsplit * (D + 2);
    if constexpr (NS > 0) {
        const int sl = slot[b];
        float mv[NS], nv[NS], lv[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int sc = s < nsplit ? s : nsplit - 1;            // clamped, not predicated: one batch of loads, no branch
            mv[s] = w[sc * (D + 2) + D];
            nv[s] = w[sc * (D + 2) + d];
            lv[s] = w[sc * (D + 2) + D + 1];
        }
        const int nlive = (sl + DF_CHUNK) / DF_CHUNK;
        float m = -INFINITY;
#pragma unroll
        for (int s = 0; s < NS; ++s) m = s < nlive ? fmaxf(m, mv[s]) : m;
        float num = 0.f, den = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (s < nlive) {
                const float f = (mv[s] == -INFINITY) ? 0.f : __expf(mv[s] - m);
                num = fmaf(f, nv[s], num);
                den = fmaf(f, lv[s], den);
            }
        }
        o[(size_t)b * o_sb + (size_t)h * o_sh + d] = f2bf(den > 0.f ? num / den : 0.f);
        return;
    }
    const int nlive = (slot[b] + DF_CHUNK) / DF_CHUNK;             // ceil((slot + 1) / DF_CHUNK): dead splits wrote nothing
    float m = -INFINITY;
    for (int s = 0; s < nlive; ++s) m = fmaxf(m, w[s * (D + 2) + D]);
    float num = 0.f, den = 0.f;
#pragma unroll 4
    for (int s = 0; s < nlive; ++s) {
        const float ms = w[s * (D + 2) + D];
        const float f = (ms == -INFINITY) ? 0.f : __expf(ms - m);
        num = fmaf(f, w[s * (D + 2) + d], num);
        den = fmaf(f, w[s * (D + 2) + D + 1], den);
    }
    o[(size_t)b * o_sb + (size_t)h * o_sh + d] = f2bf(den > 0.f ? num / den : 0.f);
}
template <int D>
void launch_combine(const DecodeFusedArgs& a, int ns, hipStream_t s) {
    const dim3 grid(a.H, a.B), block(D);
    if (ns <= 8 && !(emu_gemm_tune_get() & (1 << 20)))
        hipLaunchKernelGGL((decode_fused_combine_kernel<D, 8>), grid, block, 0, s, a.ws, a.slot, a.o, a.o_sb, a.o_sh, a.H, ns);
    else if (ns <= 16 && !(emu_gemm_tune_get() & (1 << 20)))
        hipLaunchKernelGGL((decode_fused_combine_kernel<D, 16>), grid, block, 0, s, a.ws, a.slot, a.o, a.o_sb, a.o_sh, a.H, ns);
    else
        hipLaunchKernelGGL((decode_fused_combine_kernel<D, 0>), grid, block, 0, s, a.ws, a.slot, a.o, a.o_sb, a.o_sh, a.H, ns);
}
}  // namespace

size_t decode_fused_ws_floats(int B, int H, int D, int ctx_max) {
    const int ns = ctx_max < 1 ? 1 : (ctx_max + DF_CHUNK - 1) / DF_CHUNK;
    return (size_t)B * H * ns * (D + 2);
}

int launch_decode_fused(const DecodeFusedArgs& a, hipStream_t s) {
    if (a.ctx_max < 1 || a.ctx_max > a.S_max) return -22;
    // shared slots are prompt slots: the new token's slot (< ctx_max) is never one of them
    if (a.share_nb > 1 && (a.share_nb > DECODE_SHARE_MAX || a.B % a.share_nb || a.share_len < 0 || a.share_len >= a.ctx_max)) return -22;
    const int ns = (a.ctx_max + DF_CHUNK - 1) / DF_CHUNK;
    if (a.share_nb > 1) {
        const int nsh = a.share_len / DF_CHUNK;         // < ns: share_len < ctx_max (checked above)
        const dim3 grid(nsh * a.H * (a.B / a.share_nb) + (ns - nsh) * a.H * a.B);
#define EMU_SHARE_CASE(DD, NBT) hipLaunchKernelGGL((decode_fused_kernel<DD, NBT>), grid, dim3(256), 0, s, a, ns)
        if (a.D == 128) {
            if (a.share_nb <= 2) EMU_SHARE_CASE(128, 2);
            else if (a.share_nb <= 4) EMU_SHARE_CASE(128, 4);
            else if (a.share_nb <= 5) EMU_SHARE_CASE(128, 5);
            else EMU_SHARE_CASE(128, 8);
        } else if (a.D == 64) {
            if (a.share_nb <= 2) EMU_SHARE_CASE(64, 2);
            else if (a.share_nb <= 4) EMU_SHARE_CASE(64, 4);
            else if (a.share_nb <= 5) EMU_SHARE_CASE(64, 5);
            else EMU_SHARE_CASE(64, 8);
        } else return -22;
#undef EMU_SHARE_CASE
        if (a.D == 128) launch_combine<128>(a, ns, s);
        else launch_combine<64>(a, ns, s);
        EMU_CHECK_LAUNCH();
        return 0;
    }
    if (a.D == 128) {
        hipLaunchKernelGGL((decode_fused_kernel<128, 0>), dim3(ns, a.H, a.B), dim3(256), 0, s, a, ns);
        if (!a.arrive && !a.skip_combine) launch_combine<128>(a, ns, s);
    } else if (a.D == 64) {
        hipLaunchKernelGGL((decode_fused_kernel<64, 0>), dim3(ns, a.H, a.B), dim3(256), 0, s, a, ns);
        if (!a.arrive) launch_combine<64>(a, ns, s);
    } else return -22;
    EMU_CHECK_LAUNCH();
    return 0;
}

int launch_decode_attn(const DecodeAttnArgs& a, hipStream_t s) {
    const int cmax = a.ctx_max > 0 ? a.ctx_max : a.ctx;
    if (cmax < 1 || cmax > a.S_max || a.ctx > cmax) return -22;
    const int ns = decode_attn_nsplit(cmax);
    if (a.D == 128) {
        hipLaunchKernelGGL(decode_attn_kernel<128>, dim3(ns, a.H, a.B), dim3(256), 0, s, a, ns);
        hipLaunchKernelGGL(decode_combine_kernel<128>, dim3(a.H, a.B), dim3(128), 0, s, a, ns);
    } else if (a.D == 64) {
        hipLaunchKernelGGL(decode_attn_kernel<64>, dim3(ns, a.H, a.B), dim3(256), 0, s, a, ns);
        hipLaunchKernelGGL(decode_combine_kernel<64>, dim3(a.H, a.B), dim3(64), 0, s, a, ns);
    } else return -22;
    EMU_CHECK_LAUNCH();
    return 0;
}
