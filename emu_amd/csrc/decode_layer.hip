// Whole LLaMA decoder layers of a single-row decode step in ONE launch (DecodeLayersArgs, kernels.h).
//
// The multi-launch step (engine.hip) runs six kernels per layer -- qkv GEMV (+RMSNorm), RoPE / KV append / split attention, split
// combine, o_proj GEMV, gate/up GEMV (+RMSNorm, SwiGLU), down GEMV -- plus two all-reduces under tensor parallelism.  Every boundary
// drains the weight stream, pays the launch gap and refills; the attention launches (13.7 us per layer at 1.5 TB/s) stream no weights
// at all; and a TP = 8 shard's layer is eight launches of a few microseconds each.  Here the same workgroups -- same rows per
// workgroup, same per-thread summation order, hence the same bits as the launches they replace -- are ROLES of one grid:
//
//     [ Q: qkv rows | A: (split, head) attention | O: o_proj rows | G: gate/up rows | D: down rows ]  x layers
//
// in dispatch order, so every dependency points to a lower workgroup index (the hardware starts workgroups of one XCD in index
// order and deals indices round-robin over the XCDs: a waiting workgroup can only be waiting for workgroups that were started
// before it and that wait, by induction, for nothing unfinished -- no deadlock; every wait is bounded in wall-clock time anyway).
// A consumer requests its WHOLE weight slice (the immutable operand) first, then waits for its input vector, so the HBM stream
// runs through the dependency edges instead of stopping at them; attention workgroups request their K / V rows before q exists.
//
// Hand-off (MI355X_MICROARCH.md, workgroup visibility): producers publish with agent-scope (sc1, write-through) stores from wave 0,
// wait for the acknowledgement (s_waitcnt vmcnt(0)) and bump a relaxed agent-scope arrival counter; consumers poll the counter from
// one lane (s_sleep between polls) and read the vector with sc1 buffer loads.  No cache-wide fence anywhere.  Arrival counters
// are sharded (16 sub-counters + a top counter per phase: 3-9 thousand arrivals per phase would saturate one word at ~88 / us).
// Under tensor parallelism the workgroup that completes the o_proj / down_proj top counter runs p2p.hip's all-reduce protocol on the
// partial vector (same comm blocks, sequence counters and summation order) and releases the consumers afterwards.
//
// Replaces (reference call sites): the LlamaDecoderLayer loop reached from Emu2/emu/emu.py:133-138, :213-229 at one new token per
// step, and the layer-placement "model parallel" of Emu2/emu/mixin.py:44-81.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int DL_NSUB = 16;                       // arrival sub-counters per phase
constexpr int DL_LINE = 16;                       // ints per counter slot (64 bytes: one slot per cache line pair half)
constexpr int DL_PH_INTS = (2 + DL_NSUB) * DL_LINE;   // top, ready, sub-counters
enum { PH_A = 0, PH_O = 1, PH_G = 2, PH_D = 3, PH_COUNT = 4 };
enum { DL_Q = 0, DL_A = 1, DL_O = 2, DL_G = 3, DL_D = 4 };     // roles, in dispatch order
constexpr int DL_CHUNK = 128;                     // keys per attention workgroup (attention.hip: DF_CHUNK)
constexpr int AUX_SC1 = 16;                       // buffer-op cache policy: agent scope (bypasses L1, coherent at the L2 / fabric)
constexpr int AUX_SYS = 17;                       // sc0 | sc1: system scope (peer GPUs)

using rsrc_t = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ u32x4 ldb16(rsrc_t r, uint32_t off, int aux_is_sys = 0) {
    return aux_is_sys ? __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, AUX_SYS) : __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, AUX_SC1);
}
// streamed (read-once, nt) weight vector: descriptor over the whole matrix, per-lane byte offset within a row, scalar row offset
__device__ __forceinline__ u32x4 ldw16(rsrc_t r, uint32_t voff, uint32_t row_off) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, voff, row_off, 2);
}
__device__ __forceinline__ int ld_agent(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent32(void* p, uint32_t v) {
    __hip_atomic_store(reinterpret_cast<uint32_t*>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent64(void* p, uint32_t lo, uint32_t hi) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), ((unsigned long long)hi << 32) | lo, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
// this wave's stores have reached the agent's point of coherence (inline asm: invisible to the pass that drops a provably
// redundant s_waitcnt in front of a flag store)
__device__ __forceinline__ void stores_acked() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__device__ __forceinline__ int* head_slot(int* cl, int h) { return cl + h * DL_LINE; }
__device__ __forceinline__ int* phase(int* cl, int Hl, int p) { return cl + Hl * DL_LINE + p * DL_PH_INTS; }
__device__ __forceinline__ int nsub_of(int n) { return n < DL_NSUB ? n : DL_NSUB; }

// Every wave waits for itself (uniform loads, no workgroup barrier, so nothing here drains the weight requests in flight).  `seen` is
// the counter as read by a load issued AHEAD of the wave's weight requests (loads return in order: it is back long before they are):
// in the steady state of a phase it already says ready and the wait costs nothing; only the first ~1000 workgroups of a phase poll.
__device__ __forceinline__ int dl_peek(const int* p, int target) { return p ? ld_agent(p) : target; }
__device__ __forceinline__ void dl_wait(const int* p, int target, int seen, const DecodeLayersArgs& a) {
    if (p && __builtin_amdgcn_readfirstlane(seen) < target) {
        const bool dead = __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
        const long long t0 = wall_clock64();
        while (__builtin_amdgcn_readfirstlane(ld_agent(p)) < target) {
            if (dead || wall_clock64() - t0 > a.limit_ticks) {
                if ((threadIdx.x & 63) == 0) atomicAdd(a.err, 1u);
                break;
            }
            __builtin_amdgcn_s_sleep(4);
        }
    }
#ifdef EMU_TRACE
    if (a.trace && threadIdx.x == 0) a.trace[(size_t)blockIdx.x * 4 + 2] = (unsigned long long)wall_clock64();
#endif
}

// arrival of workgroup idx (of n) at a phase; lane 0 of wave 0, after stores_acked().  True for the one workgroup that completes
// the phase (it releases the consumers: the top counter itself without tensor parallelism, the ready word after the all-reduce).
__device__ __forceinline__ bool dl_arrive(int* ph, int idx, int n) {
    const int i = idx % DL_NSUB;
    const int expect = n / DL_NSUB + (i < n % DL_NSUB ? 1 : 0);
    const int old = __hip_atomic_fetch_add(ph + (2 + i) * DL_LINE, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old != expect - 1) return false;
    const int t = __hip_atomic_fetch_add(ph, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return t == nsub_of(n) - 1;
}
// what a consumer of phase ph (n producer workgroups) polls
__device__ __forceinline__ const int* ready_word(int* ph, bool tp) { return tp ? ph + DL_LINE : ph; }
__device__ __forceinline__ int ready_target(int n, bool tp) { return tp ? 1 : nsub_of(n); }

// ------------------------------------------------------------------------------------------------------------------ all-reduce
// p2p.hip's protocol run by ONE 256-thread workgroup on x[0, n) (bf16 partial sums written by this launch with sc1 stores):
// publish the pieces into my comm block (system-scope stores), flag them, wait for every rank's flags (bounded), sum the ranks'
// slots in rank order in fp32, write x back (sc1 stores).  Same pieces, sequence counters, slots and summation order as
// p2p_allreduce_kernel, so the two may alternate (prefill through the kernel, decode through here) and give the same bits.
__device__ __forceinline__ unsigned long long* flag_of(char* block, int slot, int g) {
    return reinterpret_cast<unsigned long long*>(block + 2 * EMU_P2P_SLOT_BYTES) + slot * EMU_P2P_PIECES + g;
}
__device__ void dl_allreduce(const DecodeLayersArgs& a, bf16_t* x, int n, float* sm) {
    const int tid = threadIdx.x;
    const int npieces = (n + EMU_P2P_PIECE - 1) / EMU_P2P_PIECE;             // <= 4 for n <= 16384 (host-checked)
    unsigned long long* s_seq = reinterpret_cast<unsigned long long*>(sm);  // [npieces]
    if (tid < npieces) { const unsigned long long s = a.tp_seq[tid] + 1; a.tp_seq[tid] = s; s_seq[tid] = s; }
    __syncthreads();
    const rsrc_t rx = make_rsrc(x, (uint32_t)n * 2);
    const rsrc_t rmine = make_rsrc(a.tp_block[a.tp_rank], (uint32_t)(2 * EMU_P2P_SLOT_BYTES));
    // ---- 1. publish (n % 8 == 0, host-checked: whole 16-byte vectors)
    for (int v = tid; v * 8 < n; v += 256) {
        const int g = (v * 8) / EMU_P2P_PIECE;
        const uint32_t slot = (uint32_t)(s_seq[g] & 1);
        const u32x4 val = ldb16(rx, (uint32_t)v * 16);
        __builtin_amdgcn_raw_buffer_store_b128(val, rmine, slot * (uint32_t)EMU_P2P_SLOT_BYTES + (uint32_t)v * 16, 0, AUX_SYS);
    }
    stores_acked();
    __syncthreads();
    if (tid < npieces)
        __hip_atomic_store(flag_of(a.tp_block[a.tp_rank], (int)(s_seq[tid] & 1), tid), s_seq[tid], __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
    // ---- 2. wait: thread (r, g) polls rank r's flag of piece g
    if (tid < a.tp_n * npieces) {
        const int r = tid / npieces, g = tid % npieces;
        const unsigned long long s = s_seq[g];
        unsigned long long* f = flag_of(a.tp_block[r], (int)(s & 1), g);
        if (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < s) {
            const bool dead = __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
            const long long t0 = wall_clock64();
            while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < s) {
                if (dead || wall_clock64() - t0 > a.limit_ticks) { atomicAdd(a.err, 1u); break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
    }
    __syncthreads();
    // ---- 3. sum in rank order (fp32), round once
    for (int v = tid; v * 8 < n; v += 256) {
        const int g = (v * 8) / EMU_P2P_PIECE;
        const uint32_t off = (uint32_t)(s_seq[g] & 1) * (uint32_t)EMU_P2P_SLOT_BYTES + (uint32_t)v * 16;
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int r = 0; r < a.tp_n; ++r) {
            const rsrc_t rr = make_rsrc(a.tp_block[r], (uint32_t)(2 * EMU_P2P_SLOT_BYTES));
            float f[8];
            unpack8(ldb16(rr, off, 1), f);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += f[j];
        }
        __builtin_amdgcn_raw_buffer_store_b128(pack8(acc), rx, (uint32_t)v * 16, 0, AUX_SC1);
    }
    stores_acked();
    __syncthreads();
}

// the workgroup that completed phase ph: all-reduce (tensor parallelism) and release.  Every thread of the workgroup calls this with
// the same `closer` (broadcast through LDS by the caller).
__device__ __forceinline__ void dl_close(const DecodeLayersArgs& a, int* ph, bf16_t* x, int n, float* sm) {
    dl_allreduce(a, x, n, sm);
    if (threadIdx.x == 0) __hip_atomic_store(ph + DL_LINE, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ----------------------------------------------------------------------------------------------- Q / G: rows behind an RMSNorm
// gemv_rt_kernel<4, 4, true, EPI, false> (gemv.hip; K <= 8192 = four 256-lane trips): the arrival counter is requested first, then
// the WHOLE weight slice; the counter is back within a microsecond (loads return in order), the gain and the input follow the weights
// into the queue, and the statistics go through a raw s_barrier.  gemv_kernel<4, 1, true, EPI, -1> (the rolling "head" form the
// multi-launch step prefers at TP = 1) sums in the same order, so either launch this replaces gives these bits; inside this grid the
// rolling form measured 25-35 % SLOWER per phase (profiles/r05_decode_fused_timeline_*.log): its input loads sit on the dependency
// chain of trips 2 and 3, and here they are agent-scope loads served by the L2, not L1 hits.
template <int EPI>
__device__ __forceinline__ void role_norm_rows(const DecodeLayersArgs& a, const bf16_t* W, const bf16_t* gain, const bf16_t* x,
                                               bf16_t* out, int N, int blk, const int* wp, int wt, float* sm) {
    constexpr int R = 4, KIT = 4;
    float* red = sm;                 // [4][R]
    float* ssp = sm + 20;            // [4]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int K = a.H;
    const int n0 = blk * R;
    const int seen = dl_peek(wp, wt);
    // one descriptor per operand: lanes beyond a row's end read on into the next row (zero beyond the matrix) against a zero
    // activation (the activation / gain descriptors end at K)
    const rsrc_t rw = make_rsrc(W, (uint32_t)N * (uint32_t)K * 2);
    u32x4 wv[KIT][R];
#pragma unroll
    for (int it = 0; it < KIT; ++it)
#pragma unroll
        for (int r = 0; r < R; ++r) wv[it][r] = ldw16(rw, (uint32_t)(tid + 256 * it) * 16, (uint32_t)(n0 + r) * (uint32_t)K * 2);
    dl_wait(wp, wt, seen, a);
    const rsrc_t rx = make_rsrc(x, (uint32_t)K * 2);
    const rsrc_t rg = make_rsrc(gain, (uint32_t)K * 2);
    u32x4 xv[KIT], gv[KIT];
#pragma unroll
    for (int it = 0; it < KIT; ++it) xv[it] = ldb16(rx, (uint32_t)(tid + 256 * it) * 16);    // beyond K: zero
#pragma unroll
    for (int it = 0; it < KIT; ++it) gv[it] = __builtin_amdgcn_raw_buffer_load_b128(rg, (uint32_t)(tid + 256 * it) * 16, 0, 0);
    float ss = 0.f;
#pragma unroll
    for (int it = 0; it < KIT; ++it) {
        float f[8];
        unpack8(xv[it], f);
#pragma unroll
        for (int j = 0; j < 8; ++j) ss += f[j] * f[j];
    }
    ss = wave_sum(ss);
    if (lane == 0) *reinterpret_cast<volatile float*>(&ssp[wave]) = ss;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const volatile float* sp = ssp;
    const float rinv = rsqrtf((sp[0] + sp[1] + sp[2] + sp[3]) / (float)K + a.eps);
    // the packed vectors are opaque from here: unpack again below instead of keeping 32 floats alive across the reduction (the
    // weight slice holds 64 registers)
#pragma unroll
    for (int it = 0; it < KIT; ++it) asm volatile("" : "+v"(xv[it]));
#pragma unroll
    for (int it = 0; it < KIT; ++it) {                               // xv <- bf16(g * bf16(x * rinv)), the dot2 operand
        float xf[8], g[8];
        unpack8(xv[it], xf);
        unpack8(gv[it], g);
#pragma unroll
        for (int j = 0; j < 8; ++j) xf[j] = g[j] * bfround(xf[j] * rinv);
        xv[it] = pack8(xf);                                          // the reference's second rounding
    }
    __builtin_amdgcn_sched_barrier(0);
    float acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.f;
#pragma unroll
    for (int it = 0; it < KIT; ++it)
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float t = acc[r];
            t = bf16_dot2(wv[it][r].x, xv[it].x, t);
            t = bf16_dot2(wv[it][r].y, xv[it].y, t);
            t = bf16_dot2(wv[it][r].z, xv[it].z, t);
            t = bf16_dot2(wv[it][r].w, xv[it].w, t);
            acc[r] = t;
        }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const float v = wave_sum(acc[r]);
        if (lane == 0) red[wave * R + r] = v;
    }
    __syncthreads();
    if (tid == 0) {
        float f[R];
#pragma unroll
        for (int r = 0; r < R; ++r) f[r] = red[r] + red[R + r] + red[2 * R + r] + red[3 * R + r];
        if constexpr (EPI == EPI_SWIGLU) {
            // rows (2j, 2j + 1) = (gate_j, up_j); out[n0 / 2 + j] = bf16(bf16(silu(bf16 gate)) * bf16 up)
            const float o0 = bfround(silu(bfround(f[0]))) * bfround(f[1]);
            const float o1 = bfround(silu(bfround(f[2]))) * bfround(f[3]);
            st_agent32(out + (n0 >> 1), packbf(o0, o1));
        } else {
            st_agent64(out + n0, packbf(bfround(f[0]), bfround(f[1])), packbf(bfround(f[2]), bfround(f[3])));
        }
        stores_acked();
    }
}

// ----------------------------------------------------------------------------------------------------------- O / D: plain rows
// Block form: R rows per workgroup, KIT 256-lane trips, every weight load up front (gemv_kernel<2, 1, false, *, 4> for K <= 8192,
// gemv_rt_kernel<2, 9, false, *, false> for K <= 18432: same per-thread slices and summation order; rows are independent, so R is
// free -- o_proj takes 4 to keep as many bytes in flight per workgroup as the 2 x 9 trips of down_proj).
template <int R, int KIT>
__device__ __forceinline__ void role_rows_block(const DecodeLayersArgs& a, const bf16_t* W, const bf16_t* x, const bf16_t* res,
                                                bf16_t* out, int N, int K, int blk, const int* wp, int wt, float* sm) {
    static_assert(R == 2 || R == 4, "2 or 4 rows per workgroup");
    float* red = sm;                 // [4][R]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int n0 = blk * R;
    const int seen = dl_peek(wp, wt);
    const rsrc_t rw = make_rsrc(W, (uint32_t)N * (uint32_t)K * 2);
    u32x4 wv[KIT][R];
#pragma unroll
    for (int it = 0; it < KIT; ++it)
#pragma unroll
        for (int r = 0; r < R; ++r) wv[it][r] = ldw16(rw, (uint32_t)(tid + 256 * it) * 16, (uint32_t)(n0 + r) * (uint32_t)K * 2);
    dl_wait(wp, wt, seen, a);
    const rsrc_t rx = make_rsrc(x, (uint32_t)K * 2);
    u32x4 xv[KIT];
#pragma unroll
    for (int it = 0; it < KIT; ++it) xv[it] = ldb16(rx, (uint32_t)(tid + 256 * it) * 16);
    u32x2 rq = {0u, 0u};
    if (res && tid == 0) {
        const rsrc_t rr = make_rsrc(res, (uint32_t)N * 2);
        if constexpr (R == 4) rq = __builtin_amdgcn_raw_buffer_load_b64(rr, (uint32_t)n0 * 2, 0, AUX_SC1);
        else rq.x = __builtin_amdgcn_raw_buffer_load_b32(rr, (uint32_t)n0 * 2, 0, AUX_SC1);
    }
    float acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.f;
#pragma unroll
    for (int it = 0; it < KIT; ++it)
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float t = acc[r];
            t = bf16_dot2(wv[it][r].x, xv[it].x, t);
            t = bf16_dot2(wv[it][r].y, xv[it].y, t);
            t = bf16_dot2(wv[it][r].z, xv[it].z, t);
            t = bf16_dot2(wv[it][r].w, xv[it].w, t);
            acc[r] = t;
        }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const float v = wave_sum(acc[r]);
        if (lane == 0) red[wave * R + r] = v;
    }
    __syncthreads();
    if (tid == 0) {
        float v[R];
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = bfround(red[r] + red[R + r] + red[2 * R + r] + red[3 * R + r]);
        if (res) {
            v[0] += bflo(rq.x); v[1] += bfhi(rq.x);
            if constexpr (R == 4) { v[2] += bflo(rq.y); v[3] += bfhi(rq.y); }
        }
        if constexpr (R == 4) st_agent64(out + n0, packbf(v[0], v[1]), packbf(v[2], v[3]));
        else st_agent32(out + n0, packbf(v[0], v[1]));
        stores_acked();
    }
}

// Wave form (gemv_wave_kernel<4, KITW, *>: K <= 2560, the o_proj / down_proj of a tensor-parallel shard): every wave owns 4 whole
// rows, 16 rows per workgroup, no cross-wave sum.
template <int KITW>
__device__ __forceinline__ void role_rows_wave(const DecodeLayersArgs& a, const bf16_t* W, const bf16_t* x, const bf16_t* res,
                                               bf16_t* out, int N, int K, int blk, const int* wp, int wt) {
    constexpr int RW = 4;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int n0 = (blk * 4 + __builtin_amdgcn_readfirstlane(wave)) * RW;
    const int seen = dl_peek(wp, wt);
    const rsrc_t rw = make_rsrc(W, (uint32_t)N * (uint32_t)K * 2);
    u32x4 wv[KITW][RW];
#pragma unroll
    for (int it = 0; it < KITW; ++it)
#pragma unroll
        for (int r = 0; r < RW; ++r) wv[it][r] = ldw16(rw, (uint32_t)(lane + 64 * it) * 16, (uint32_t)(n0 + r) * (uint32_t)K * 2);
    dl_wait(wp, wt, seen, a);
    const rsrc_t rx = make_rsrc(x, (uint32_t)K * 2);
    u32x4 xv[KITW];
#pragma unroll
    for (int it = 0; it < KITW; ++it) xv[it] = ldb16(rx, (uint32_t)(lane + 64 * it) * 16);
    u32x2 rq = {0u, 0u};
    if (res && lane == 0 && n0 < N) rq = __builtin_amdgcn_raw_buffer_load_b64(make_rsrc(res, (uint32_t)N * 2), (uint32_t)n0 * 2, 0, AUX_SC1);
    float acc[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) acc[r] = 0.f;
#pragma unroll
    for (int it = 0; it < KITW; ++it)
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            float t = acc[r];
            t = bf16_dot2(wv[it][r].x, xv[it].x, t);
            t = bf16_dot2(wv[it][r].y, xv[it].y, t);
            t = bf16_dot2(wv[it][r].z, xv[it].z, t);
            t = bf16_dot2(wv[it][r].w, xv[it].w, t);
            acc[r] = t;
        }
#pragma unroll
    for (int r = 0; r < RW; ++r) acc[r] = wave_sum(acc[r]);
    if (lane == 0 && n0 < N) {                                       // N % 4 == 0 (host-checked): whole groups of 4 rows
        float v[RW];
#pragma unroll
        for (int r = 0; r < RW; ++r) v[r] = bfround(acc[r]);
        if (res) { v[0] += bflo(rq.x); v[1] += bfhi(rq.x); v[2] += bflo(rq.y); v[3] += bfhi(rq.y); }
        st_agent64(out + n0, packbf(v[0], v[1]), packbf(v[2], v[3]));
    }
    stores_acked();
    __syncthreads();                                                 // every wave's rows acknowledged before wave 0 arrives
}

// --------------------------------------------------------------------------------------------------------------- A: attention
// decode_fused_kernel<128, 0> (attention.hip) for one row: the split's K / V rows are requested before the head's q / k / v
// exist; the head's last split to arrive merges the live splits (decode_fused_combine_kernel's arithmetic) and publishes the head.
// Returns true (block-uniform) when this workgroup published the head's output.
__device__ __forceinline__ bool role_attention(const DecodeLayersArgs& a, bf16_t* kcl, bf16_t* vcl, int split, int h, int* cl, int nsplit,
                                               bool wait_q, float* smraw) {
    constexpr int D = 128, LPK = D / 8, KPI = 64 / LPK, ITER = (DL_CHUNK / 4) / KPI, NP = 4 * KPI;
    float (*sm)[D + 2] = reinterpret_cast<float (*)[D + 2]>(smraw);
    int* s_last = reinterpret_cast<int*>(smraw + NP * (D + 2));
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int g = lane / LPK, dl = lane % LPK, d0 = dl * 8;
    const int slot = a.slot[0], ctx = slot + 1;
    const int k0 = split * DL_CHUNK;
    if (k0 >= ctx) return false;                       // split beyond the live context: no work, not counted
    const int kstart = a.kstart ? a.kstart[0] : 0;
    const int pos = a.pos[0];
    constexpr int half = D / 2;
    const int dp = (d0 + half) % D, dc = d0 % half;
    const float sgn = d0 < half ? -1.f : 1.f;
    const size_t hb = (size_t)h * a.S_max;
    const bf16_t* kc = kcl + hb * D;
    const bf16_t* vc = vcl + hb * D;
    const int* qp = wait_q ? head_slot(cl, h) : nullptr;
    const int seen = dl_peek(qp, 3 * D / 4);
    const u32x4 cv = ld16(a.cos + (size_t)pos * D + dc), sv = ld16(a.sin + (size_t)pos * D + dc);
    u32x4 kr[ITER], vr[ITER];
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const int key = k0 + wave * (DL_CHUNK / 4) + it * KPI + g;
        const int kc_i = key < a.S_max ? key : a.S_max - 1;
        const long off = (long)kc_i * D + d0;
        kr[it] = ld16(kc + off);
        vr[it] = ld16(vc + off);
    }
    dl_wait(qp, 3 * D / 4, seen, a);                   // the 96 four-row workgroups of this head's q, k and v rows
    const int HD = a.Hl * D;
    const rsrc_t rq = make_rsrc(a.qkv, (uint32_t)(3 * HD) * 2);
    const uint32_t qo = (uint32_t)(h * D) * 2, ko = qo + (uint32_t)HD * 2, vo = ko + (uint32_t)HD * 2;
    const u32x4 q1v = ldb16(rq, qo + d0 * 2), q2v = ldb16(rq, qo + dp * 2), k1v = ldb16(rq, ko + d0 * 2), k2v = ldb16(rq, ko + dp * 2),
                nvv = ldb16(rq, vo + d0 * 2);
    // (register budget: the 16 K / V vectors hold 64 registers; the rotated key and the new value stay PACKED -- exact, they are
    // bf16 values -- and are selected per key before the unpack)
    float q[8];
    u32x4 nkp;
    const u32x4 nvp = nvv;
    {
        float c[8], sn[8], x1[8], x2[8];
        unpack8(cv, c);
        unpack8(sv, sn);
        unpack8(q1v, x1); unpack8(q2v, x2);
#pragma unroll
        for (int j = 0; j < 8; ++j) q[j] = bfround(bfround(x1[j] * c[j]) + bfround(sgn * x2[j] * sn[j]));
        __builtin_amdgcn_sched_barrier(0);
        unpack8(k1v, x1); unpack8(k2v, x2);
#pragma unroll
        for (int j = 0; j < 8; ++j) x1[j] = bfround(bfround(x1[j] * c[j]) + bfround(sgn * x2[j] * sn[j]));
        nkp = pack8(x1);
    }
    if (split == slot / DL_CHUNK && wave == 0 && g == 0) {       // append the new token to the cache (once)
        st16(kcl + (hb + slot) * D + d0, nkp);
        st16(vcl + (hb + slot) * D + d0, nvp);
    }
    __builtin_amdgcn_sched_barrier(0);
    float sd[ITER];
    float m = -INFINITY;
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const int key = k0 + wave * (DL_CHUNK / 4) + it * KPI + g;
        const bool valid = key < ctx && key >= kstart;
        float kf[8];
        unpack8(key == slot ? nkp : kr[it], kf);
#pragma unroll
        for (int j = 0; j < 8; ++j) kf[j] = (key == slot || valid) ? kf[j] : 0.f;
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) t = fmaf(kf[j], q[j], t);
        t = row16_sum(t);
        sd[it] = valid ? t * a.scale : -INFINITY;
        m = fmaxf(m, sd[it]);
    }
    float l = 0.f, acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const int key = k0 + wave * (DL_CHUNK / 4) + it * KPI + g;
        float vf[8];
        unpack8(key == slot ? nvp : vr[it], vf);
#pragma unroll
        for (int j = 0; j < 8; ++j) vf[j] = (key == slot || sd[it] != -INFINITY) ? vf[j] : 0.f;
        const float p = sd[it] == -INFINITY ? 0.f : __expf(sd[it] - m);
        l += p;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = fmaf(p, vf[j], acc[j]);
    }
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
    // ---- publish the split's state (agent-scope stores), arrive; the head's last split merges (attention.hip, round 4)
    float* wout = a.ws + ((size_t)h * nsplit + split) * (D + 2);
    if (tid < D) {
        __hip_atomic_store(wout + tid, num, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid == 0) {
            __hip_atomic_store(wout + D, mt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(wout + D + 1, den, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    stores_acked();
    __syncthreads();
    const int nlive = (slot + DL_CHUNK) / DL_CHUNK;
    if (tid == 0) {
        const int old = __hip_atomic_fetch_add(head_slot(cl, h) + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *s_last = old == nlive - 1;
    }
    __syncthreads();
    if (!*s_last) return false;
    float o = 0.f;
    if (tid < D) {
        const float* w = a.ws + (size_t)h * nsplit * (D + 2);
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
        o = bfround(den2 > 0.f ? num2 / den2 : 0.f);
        const float o_next = __shfl_down(o, 1);
        if (!(tid & 1)) st_agent32(a.attn + (size_t)h * D + tid, packbf(o, o_next));
    }
    stores_acked();
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(phase(cl, a.Hl, PH_A), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
}

// ------------------------------------------------------------------------------------------------------------------- the grid
template <bool WAVE>
__device__ __forceinline__ int decode_layers_body(const DecodeLayersArgs& a, float* sm, int& s_closer) {
    const int tid = threadIdx.x;
    int b = blockIdx.x;
    const int li = b / a.per_layer;
    b -= li * a.per_layer;
    const int layer = a.layer0 + li;
    const DecodeLayerPtrs L = a.table[layer];
    int* cl = a.cnt + (size_t)layer * a.cnt_stride;
    const bool tp = a.tp_n > 0;
    const int HD = a.Hl * 128;
    // roles [role0, role1) of every layer are in this launch; a role waits only for producers of its own launch (the others
    // finished behind a kernel boundary)
    const int r0 = a.role0;
    if (r0 <= DL_Q) {
        if (b < a.nQ) {
            // input: hA, from a previous launch (first layer of this one) or from the previous layer's down rows
            int* phd = phase(li > 0 ? cl - a.cnt_stride : cl, a.Hl, PH_D);
            role_norm_rows<EPI_NONE>(a, L.wqkv, L.ln1, a.hA, a.qkv, 3 * HD, b, li > 0 ? ready_word(phd, tp) : nullptr,
                                     ready_target(a.nD, tp), sm);
            if (tid == 0) {
                const int h = ((b * 4) % HD) >> 7;
                __hip_atomic_fetch_add(head_slot(cl, h), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            return DL_Q;
        }
        b -= a.nQ;
    }
    if (r0 <= DL_A && a.role1 > DL_A) {
        if (b < a.nA) {
            const int nsplit = a.nA / a.Hl;
            role_attention(a, a.kcache + (size_t)layer * a.kv_layer, a.vcache + (size_t)layer * a.kv_layer, b % nsplit, b / nsplit, cl, nsplit,
                           r0 <= DL_Q, sm);
            return DL_A;
        }
        b -= a.nA;
    }
    bool closer = false;
    int role = DL_D;
    int* ph = nullptr;
    bf16_t* vec = nullptr;
    if (r0 <= DL_O && a.role1 > DL_O && b < a.nO) {
        role = DL_O;
        ph = phase(cl, a.Hl, PH_O);
        vec = a.hB;
        const int* wp = r0 <= DL_A ? phase(cl, a.Hl, PH_A) : nullptr;
        if constexpr (WAVE) role_rows_wave<2>(a, L.wo, a.attn, a.epi_res ? a.hA : nullptr, a.hB, a.H, HD, b, wp, a.Hl);
        else role_rows_block<4, 4>(a, L.wo, a.attn, a.epi_res ? a.hA : nullptr, a.hB, a.H, HD, b, wp, a.Hl, sm);
        if (tid == 0) closer = dl_arrive(ph, b, a.nO);
    } else {
        if (r0 <= DL_O && a.role1 > DL_O) b -= a.nO;
        if (r0 <= DL_G && b < a.nG) {
            int* pho = phase(cl, a.Hl, PH_O);
            role_norm_rows<EPI_SWIGLU>(a, L.wgu, L.ln2, a.hB, a.act, 2 * a.Fl, b, r0 <= DL_O ? ready_word(pho, tp) : nullptr,
                                       ready_target(a.nO, tp), sm);
            if (tid == 0) dl_arrive(phase(cl, a.Hl, PH_G), b, a.nG);
            return DL_G;
        }
        if (r0 <= DL_G) b -= a.nG;
        ph = phase(cl, a.Hl, PH_D);
        vec = a.hA;
        const int* phg = r0 <= DL_G ? phase(cl, a.Hl, PH_G) : nullptr;
        if constexpr (WAVE) role_rows_wave<5>(a, L.wdown, a.act, a.epi_res ? a.hB : nullptr, a.hA, a.H, a.Fl, b, phg, nsub_of(a.nG));
        else role_rows_block<2, 9>(a, L.wdown, a.act, a.epi_res ? a.hB : nullptr, a.hA, a.H, a.Fl, b, phg, nsub_of(a.nG), sm);
        if (tid == 0) closer = dl_arrive(ph, b, a.nD);
    }
    if (!tp) return role;
    if (tid == 0) s_closer = closer;
    __syncthreads();
    if (s_closer) dl_close(a, ph, vec, a.H, sm);
    return role;
}

template <bool WAVE>
__global__ __launch_bounds__(256, 4) void decode_layers_kernel(const DecodeLayersArgs a) {
    __shared__ __attribute__((aligned(16))) float sm[16 * 130 + 8];
    __shared__ int s_closer;
#ifdef EMU_TRACE
    // tools/decode_trace.py (the -DEMU_TRACE twin library only): per-workgroup timeline {role | layer << 8, entry, input ready, exit}
    // in 100 MHz ticks
    if (a.trace && threadIdx.x == 0) {
        a.trace[(size_t)blockIdx.x * 4 + 1] = (unsigned long long)wall_clock64();
        a.trace[(size_t)blockIdx.x * 4 + 2] = 0;
    }
    const int role = decode_layers_body<WAVE>(a, sm, s_closer);
    if (a.trace && threadIdx.x == 0) {
        a.trace[(size_t)blockIdx.x * 4] = (unsigned long long)(role | ((a.layer0 + blockIdx.x / a.per_layer) << 8));
        a.trace[(size_t)blockIdx.x * 4 + 3] = (unsigned long long)wall_clock64();
    }
#else
    (void)decode_layers_body<WAVE>(a, sm, s_closer);
#endif
}

}  // namespace

size_t decode_layers_cnt_ints(int layers, int Hl) { return (size_t)layers * (size_t)(Hl * DL_LINE + PH_COUNT * DL_PH_INTS); }

bool decode_layers_ok(const DecodeLayersArgs& a) {
    const int HD = a.Hl * 128;
    if (a.H < 8 || a.H > 8192 || (a.H & 7) || a.Hl < 1 || (a.Fl & 7) || a.Fl < 8) return false;
    const bool wave = HD <= 2560 && a.Fl <= 2560 && HD <= 1024 && a.H >= 1024;   // o_proj on <= 2 wave trips, down on <= 5
    if (!wave && (HD > 8192 || a.Fl > 18432)) return false;
    if (a.tp_n > 0 && ((size_t)a.H * 2 > EMU_P2P_SLOT_BYTES || a.H > 4 * EMU_P2P_PIECE)) return false;
    return true;
}

int launch_decode_layers(DecodeLayersArgs a, hipStream_t s) {
    if (!decode_layers_ok(a) || a.nlayers < 1 || !a.table || !a.cnt || !a.err) return -22;
    const int HD = a.Hl * 128;
    a.wave_od = (HD <= 1024 && a.Fl <= 2560 && a.H >= 1024) ? 1 : 0;
    a.nQ = 3 * HD / 4;
    a.nA = ((a.ctx_max + DL_CHUNK - 1) / DL_CHUNK) * a.Hl;
    a.nO = a.wave_od ? (a.H + 15) / 16 : (a.H + 3) / 4;
    a.nG = a.Fl / 2;
    a.nD = a.wave_od ? (a.H + 15) / 16 : (a.H + 1) / 2;
    if (a.role1 <= a.role0) { a.role0 = DL_Q; a.role1 = DL_D + 1; }
    if (a.role0 < DL_Q || a.role1 > DL_D + 1) return -22;
    const int counts[5] = {a.nQ, a.nA, a.nO, a.nG, a.nD};
    a.per_layer = 0;
    for (int r = a.role0; r < a.role1; ++r) a.per_layer += counts[r];
    if (a.nlayers > 1 && (a.role0 != DL_Q || a.role1 != DL_D + 1)) return -22;     // layer-to-layer waits need every role in the launch
    if (a.tp_n > 0 && a.role1 <= DL_O) return -22;
    a.cnt_stride = a.Hl * DL_LINE + PH_COUNT * DL_PH_INTS;
    if (a.ctx_max < 1 || a.ctx_max > a.S_max) return -22;
    const dim3 grid((unsigned)a.per_layer * (unsigned)a.nlayers), block(256);
    if (a.wave_od) hipLaunchKernelGGL(decode_layers_kernel<true>, grid, block, 0, s, a);
    else hipLaunchKernelGGL(decode_layers_kernel<false>, grid, block, 0, s, a);
    EMU_CHECK_LAUNCH();
    return 0;
}
