// Persistent weight-streaming engine for chains of one-row GEMVs (a decode step's projections): ONE launch of one resident
// workgroup per CU, in which the weight stream never waits for an activation.
//
// Why: a tensor-parallel shard's decode layer is eight dependent launches of 4-14 us, each a single wave of workgroups whose time is
// first-byte latency + drain; the kernel durations SUM to the token time (profiles/r05_tp8_shard_decode_kernel_stats.csv: 52 us per
// layer for 137 MB = 21 us of streaming), and neither fewer launches (round 5) nor weights touched into the infinity cache ahead of
// their launch (round 6: profiles/r06_mall_gemv_probe_touch_then_gemv.log, 4-13 %) recover the difference: what bounds a launch is
// the dependent chain in front of its first byte.  Here the chain and the stream are decoupled (MI355X_MICROARCH.md, rows
// engine-vs-launches / prefetch-credit / ldsdma-fill / nt-weights / allgather):
//   * wave 0 of every workgroup is a LOADER: it walks this CU's share of every op's weight matrix in op order -- contiguous row
//     blocks, one FILL = whole rows (or one segment of a long row) of at most 16 KiB -- with `global_load_lds` (LDS-DMA, 16 bytes per
//     lane, nt: no VGPRs, no L2 / infinity-cache allocation) into a ring of 16 KiB LDS slots, three fills in flight on a counted
//     vmcnt, and runs ahead of the consumers ACROSS op boundaries until the ring is full: while the consumers wait for an
//     activation vector, up to a ring of the next op's weights (8 x 16 KiB per CU = 32 MB per chip) is already on the die;
//   * waves 1..3 are CONSUMERS: a fill's rows are multiplied against the activation vector in LDS (v_dot2c on ds_read_b128 operands,
//     DPP wave reductions) with the arithmetic of the launch kernels they replace -- the same column-to-lane assignment, the same
//     order of partial sums -- so the results are bit-identical (launch_gemv's partition is mirrored by emu_gemv_partition);
//   * an op's output vector travels to every CU through 4-byte GRANULES {0x0001, bf16}: one agent-scope (sc1, write-through) store
//     per value, the data is the flag; the first consumer wave of every CU sweeps the array with 16-byte sc1 loads until every tag
//     is set (the arrays are zeroed by a memset node in front of the launch: a tag needs no epoch), normalises (RMSNorm, the
//     reference's rounding points) into LDS, and releases its CU's consumers.  No counters, no fences, no cache-wide operation;
//   * every wait is bounded in wall-clock time (give-up counter; the results are garbage then, the host checks).
// The launch needs every workgroup resident (grid = CU count, one workgroup per CU by LDS size): a device shared with other
// processes cannot run it (the host keeps the launches there).
//
// Replaces (reference call sites): the LlamaDecoderLayer linears + RMSNorm reached from Emu2/emu/emu.py:133-138 and :213-229 at
// one new token per step, under the SURVEY 8e shard plan.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int ENG_THREADS = 512;                   // 8 waves: nload loaders + (8 - nload) consumers
constexpr int ENG_SLOT = 16384;                    // bytes per ring slot
constexpr int ENG_LAG = 3;                         // a loader's fills in flight behind the one it publishes
constexpr int ENG_MAXJ = 13;                       // 16-byte column vectors per lane and row (13 x 64 x 16 B = 13 KiB: K <= 6656)
constexpr int ENG_FLAGS = 256;                     // bytes of LDS words behind the ring and the activation buffer
constexpr int AUX_SC1 = 16;
// LDS words (index into the flags block)
constexpr int F_READY = 0, F_DONE = 16, F_XREADY = 32, F_CONSDONE = 33, F_GATHER = 34, F_XPARTS = 35;

typedef __attribute__((address_space(3))) volatile uint32_t lds_u32;

__device__ __forceinline__ void glds16_nt(const void* src, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 2);
}
__device__ __forceinline__ void glds16_def(const void* src, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
// the loader's LDS words go through inline asm: a compiler-visible LDS access next to LDS-DMA in flight draws an s_waitcnt vmcnt(0)
// (the pass cannot tell the word from a DMA destination), which would serialise the ring (profiles/r01_gemv_variants.log)
__device__ __forceinline__ uint32_t lds_ld(uint32_t byte_addr) {
    uint32_t v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(byte_addr) : "memory");
    return v;
}
__device__ __forceinline__ void lds_st(uint32_t byte_addr, uint32_t v) {
    asm volatile("ds_write_b32 %0, %1" ::"v"(byte_addr), "v"(v) : "memory");
}
__device__ __forceinline__ void wait_vmcnt_dyn(int n) {                   // n is wave-uniform
#define EMU_W(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
    switch (n) {
        EMU_W(0) EMU_W(1) EMU_W(2) EMU_W(3) EMU_W(4) EMU_W(5) EMU_W(6) EMU_W(7) EMU_W(8) EMU_W(9) EMU_W(10) EMU_W(11) EMU_W(12)
        EMU_W(13) EMU_W(14) EMU_W(15) EMU_W(16) EMU_W(17) EMU_W(18) EMU_W(19) EMU_W(20) EMU_W(21) EMU_W(22) EMU_W(23) EMU_W(24)
        EMU_W(25) EMU_W(26) EMU_W(27) EMU_W(28) EMU_W(29) EMU_W(30) EMU_W(31) EMU_W(32) EMU_W(33) EMU_W(34) EMU_W(35) EMU_W(36)
        EMU_W(37) EMU_W(38) EMU_W(39) EMU_W(40) EMU_W(41) EMU_W(42) EMU_W(43) EMU_W(44) EMU_W(45) EMU_W(46) EMU_W(47) EMU_W(48)
        EMU_W(49) EMU_W(50) EMU_W(51) EMU_W(52) EMU_W(53) EMU_W(54) EMU_W(55) EMU_W(56) EMU_W(57) EMU_W(58) EMU_W(59) EMU_W(60)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
#undef EMU_W
}

// this CU's rows of an op (the fill geometry itself comes from the host: EngOp::rps / fill_bytes / ni / grp)
__device__ __forceinline__ void cu_rows(const EngOp& o, int cu, int& row0, int& nrows, int& nfills) {
    const int unit = o.epi == EPI_SWIGLU ? 2 : 1;
    row0 = (cu * o.q + (cu < o.rem ? cu : o.rem)) * unit;
    nrows = (o.q + (cu < o.rem ? 1 : 0)) * unit;
    nfills = cu < o.rem ? o.nfills_hi : o.nfills_lo;
}

// bounded spinning: the clock is read every 64th spin only
struct Spin {
    long long t0;
    unsigned n = 0;
    bool dead = false;
    __device__ __forceinline__ bool give_up(const EngArgs& a) {
        if (dead) return true;
        if ((++n & 63) == 0 && wall_clock64() - t0 > a.limit_ticks) {
            if ((threadIdx.x & 63) == 0) atomicAdd(a.err, 1u);
            dead = true;
        }
        return dead;
    }
};

// tools (EngArgs::dbg bit 1): per-CU timeline, 32 x u64 of wall_clock64 behind the give-up counter (err + 2 as u64): 0 entry, 1 first /
// 2 last fill issued, 3 loader done, 4 + op: activation vector ready, 10 + op: consumer 0 done with the op, 16 + op: the last consumer
// done with the op (last writer), 22 + op: consumer 0's sweep of the op's input complete
__device__ __forceinline__ void eng_trace(const EngArgs& a, int slot) {
    if ((a.dbg & 2) && (threadIdx.x & 63) == 0)
        reinterpret_cast<unsigned long long*>(a.err)[1 + (size_t)blockIdx.x * 32 + slot] = (unsigned long long)wall_clock64();
}

// ------------------------------------------------------------------------------------------------------------------- loader
__device__ __forceinline__ void loader_wave(const EngArgs& a, int wave, int nload, char* ring, uint32_t flags_addr) {
    const int lane = threadIdx.x & 63, cu = blockIdx.x, nslot = a.nslot;
    // my fills in flight, oldest first: global index and DMA instruction count (a wave can have 63 requests outstanding)
    int q0 = -1, q1 = -1, q2 = -1, q3 = -1, q4 = -1, c1 = 0, c2 = 0, c3 = 0, c4 = 0, npend = 0;
    Spin sp; sp.t0 = wall_clock64();
    int g = 0;
    if (wave == 0) eng_trace(a, 0);
    auto publish_oldest = [&]() {                        // wait until q0 has landed (requests behind it: c1 + c2 + c3), publish, pop
        wait_vmcnt_dyn((npend > 1 ? c1 : 0) + (npend > 2 ? c2 : 0) + (npend > 3 ? c3 : 0) + (npend > 4 ? c4 : 0));
        lds_st(flags_addr + (F_READY + q0 % nslot) * 4, (uint32_t)(q0 + 1));
        q0 = q1; q1 = q2; q2 = q3; q3 = q4; q4 = -1;
        c1 = c2; c2 = c3; c3 = c4; c4 = 0;
        --npend;
    };
    bool first = true;
    for (int oi = 0; oi < a.nops; ++oi) {
        const EngOp& o = a.op[oi];
        int row0, nrows, nfills;
        cu_rows(o, cu, row0, nrows, nfills);
        const int fill_bytes = o.fill_bytes, ni = o.ni;
        const char* base = reinterpret_cast<const char*>(o.W) + (size_t)row0 * o.K * 2;
        const char* wend = reinterpret_cast<const char*>(o.W) + (size_t)o.N * o.K * 2 - 16;
        for (int f = 0; f < nfills; ++f, ++g) {
            if (nload > 1 && g % nload != wave) continue;
            const int slot = g % nslot;
            if (g >= nslot) {                            // the slot's previous fill (g - nslot) must have been consumed
                const uint32_t need = (uint32_t)(g - nslot + 1);
                while (lds_ld(flags_addr + (F_DONE + slot) * 4) < need) {
                    // a loader that has to wait for a slot first publishes what it still holds back: a fill is otherwise published
                    // only behind the NEXT one's requests, so at every stall (ring full at a hand-off) each loader sat on a landed
                    // fill the consumers were waiting for
                    while (npend > 0) publish_oldest();
                    if (sp.give_up(a)) break;
                    __builtin_amdgcn_s_sleep(2);
                }
            }
            if (first) {
                // the launch's first activation vector (and gain) is requested by consumer 0 at entry: let those requests into the
                // CU's memory pipeline ahead of the stream (a gather queued behind 50 KiB of DMA took 7.5 us instead of 2)
                while (lds_ld(flags_addr + F_GATHER * 4) == 0) {
                    if (sp.give_up(a)) break;
                    __builtin_amdgcn_s_sleep(1);
                }
                first = false;
                if (wave == 0) eng_trace(a, 1);
            }
            const char* fb = base + (size_t)f * fill_bytes;
            const char* src = fb + lane * 16;
            char* dst = ring + (size_t)slot * ENG_SLOT;
            if (a.dbg & 0x400) {                         // tools: default cache policy instead of nt
                for (int i = 0; i < ni; ++i) {
                    const char* p = src + (size_t)i * 1024;
                    glds16_def(p < wend ? p : wend, dst + i * 1024);
                }
            } else if (fb + (size_t)ni * 1024 <= wend + 16) {   // (wave-uniform) the whole fill lies inside the matrix
                int i = 0;
                for (; i + 4 <= ni; i += 4) {
                    glds16_nt(src + (size_t)i * 1024, dst + i * 1024);
                    glds16_nt(src + (size_t)(i + 1) * 1024, dst + (i + 1) * 1024);
                    glds16_nt(src + (size_t)(i + 2) * 1024, dst + (i + 2) * 1024);
                    glds16_nt(src + (size_t)(i + 3) * 1024, dst + (i + 3) * 1024);
                }
                for (; i < ni; ++i) glds16_nt(src + (size_t)i * 1024, dst + i * 1024);
            } else {
                for (int i = 0; i < ni; ++i) {           // the matrix ends inside this fill's last KiB: clamp the lanes beyond it
                    const char* p = src + (size_t)i * 1024;
                    glds16_nt(p < wend ? p : wend, dst + i * 1024);
                }
            }
            if (npend == 0) q0 = g; else if (npend == 1) { q1 = g; c1 = ni; } else if (npend == 2) { q2 = g; c2 = ni; }
            else if (npend == 3) { q3 = g; c3 = ni; } else { q4 = g; c4 = ni; }
            ++npend;
            // while this CU gathers an activation vector the stream is thinned to one fill in flight (MI355X_MICROARCH.md, gather-pass:
            // a sweep queued behind the CU's own refill burst takes 1.0-1.7 us per pass instead of 0.3-0.65)
            const bool thin = (a.dbg & 0x1000) ? true : ((a.dbg & 4) ? false : lds_ld(flags_addr + F_GATHER * 4) == 2);
            // fills in flight per loader: all loaders together must leave ring slots for landed fills (a full ring of requests whose
            // consumers wait for an older fill of ANOTHER loader is a deadlock: measured, bounded by the time-out)
            int lag = (a.dbg & 0x100) ? 4 : ((a.dbg & 0x200) ? 2 : ((a.dbg & 0x800) ? 1 : ENG_LAG));
            if (nload * (lag + 1) > nslot - 1) lag = (nslot - 1) / nload - 1;
            if (lag < 1) lag = 1;
            while (npend > (thin ? 1 : (lag * ni <= 60 ? lag : 3))) publish_oldest();
        }
    }
    if (wave == 0) eng_trace(a, 2);
    while (npend > 0) publish_oldest();
    if (wave == 0) eng_trace(a, 3);
}

// ----------------------------------------------------------------------------------------------------------------- consumers
// One op on one consumer wave.  JX = column vectors per lane (compile time: the row loop is straight-line code, all LDS reads of a
// row in flight at once); VW4 = the launch kernels' column partition (see the header).
template <int JX, bool VW4>
__device__ __forceinline__ void consume_op(const EngArgs& a, const EngOp& o, int oi, int c, int ncons, int& gbase, char* ring, char* xbuf,
                                           char* xraw, uint32_t seq0, lds_u32* fl, Spin& sp) {
    const int lane = threadIdx.x & 63, cu = blockIdx.x, nslot = a.nslot;
    int row0, nrows, nfills;
    cu_rows(o, cu, row0, nrows, nfills);
    const int K = o.K, KV = K >> 3, rps = o.rps, grp = o.grp;
    u32x4* xb = reinterpret_cast<u32x4*>(xbuf);
    // ---------------------------------------------------------------- activation vector -> LDS, then release
    // Vectors written before the launch: consumer 0 loads them.  Vectors of this launch (granules): EVERY consumer wave sweeps its
    // share of the array (chunks of 256 granules, c, c + ncons, ...) as soon as it is done with the previous op -- one round trip per
    // wave instead of four sequential 8 KiB passes of one wave (7-9 us per edge, measured) -- and parks the values in registers until
    // all consumers of this CU have left the previous vector; consumer 0 finishes (sum copy, RMSNorm) and releases.
    u32x4 gv[JX];
    if (c == 0 && o.gain) {
#pragma unroll
        for (int j = 0; j < JX; ++j) { const int vi = lane + 64 * j; gv[j] = ld16(o.gain + (size_t)(vi < KV ? vi : KV - 1) * 8); }
    }
    const int per = (K + (int)gridDim.x - 1) / (int)gridDim.x, e0 = cu * per;         // this CU's share of the vector
    if (o.x_src == 0) {
        if (c == 0) {
            while (fl[F_CONSDONE] < (uint32_t)(ncons * oi)) {
                if (sp.give_up(a)) break;
                __builtin_amdgcn_s_sleep(1);
            }
            u32x4 xl[JX];
#pragma unroll
            for (int j = 0; j < JX; ++j) { const int vi = lane + 64 * j; xl[j] = ld16(o.xg + (size_t)(vi < KV ? vi : KV - 1) * 8); }
            if (oi == 0 && lane == 0) fl[F_GATHER] = 1;  // the requests are in the queue: the loader may start
#pragma unroll
            for (int j = 0; j < JX; ++j) { const int vi = lane + 64 * j; xb[vi] = vi < KV ? xl[j] : u32x4{0u, 0u, 0u, 0u}; }
        }
    } else {
        if (c == 0 && lane == 0) fl[F_GATHER] = 2;       // thin the loader while this CU gathers
        if (o.x_src == 2 && c == 0) {
            // tensor-parallel sum of my share: every rank's partial granules of this all-reduce (tag = its number), summed in
            // rank order in fp32 and rounded once -- p2p_allreduce_kernel's arithmetic -- then published like any op output
            const uint32_t arn = seq0 + (uint32_t)o.ar_k, tag = (arn & 0x7fffu) | 0x8000u;
            const size_t aoff = (size_t)(arn & (ENG_AR_BUFS - 1)) * ENG_AR_MAXLEN * 4;
            for (int l0 = 0; l0 < per; l0 += 64) {
                const int e = e0 + l0 + lane;
                const bool mine = l0 + lane < per && e < K;
                uint32_t v[EMU_ENG_MAX_RANKS];
                for (;;) {
                    bool ok = true;
#pragma unroll
                    for (int r = 0; r < EMU_ENG_MAX_RANKS; ++r) {
                        v[r] = 0;
                        if (r < a.tp_n && mine) {
                            v[r] = __hip_atomic_load(reinterpret_cast<const uint32_t*>(a.comm[r] + aoff) + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                            ok &= (v[r] >> 16) == tag;
                        }
                    }
                    if (__all(ok) || sp.give_up(a)) break;
                    __builtin_amdgcn_s_sleep(2);
                }
                float acc = 0.f;
#pragma unroll
                for (int r = 0; r < EMU_ENG_MAX_RANKS; ++r)
                    if (r < a.tp_n) acc += bf2f((bf16_t)(v[r] & 0xffffu));
                if (mine) __hip_atomic_store(o.xgran + e, 0x10000u | f2bf(acc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(o.xgran), 0, (uint32_t)K * 4, 0x00020000);
        const int nch = (K + 255) >> 8;                  // chunks of 256 granules (1 KiB, one load instruction)
        constexpr int MYCH = 6;                          // chunks per wave and pass: 26 chunks (K = 6656) over 5 consumers in one pass
        bool vacated = false;
        for (int k0 = 0; c + k0 * ncons < nch; k0 += MYCH) {
            u32x4 gr[MYCH];
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int k = 0; k < MYCH; ++k)           // beyond the array: zeros (descriptor bound)
                    gr[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, (uint32_t)((c + (k0 + k) * ncons) * 1024 + lane * 16), 0, AUX_SC1);
#pragma unroll
                for (int k = 0; k < MYCH; ++k) {
                    const int e = (c + (k0 + k) * ncons) * 256 + lane * 4;
                    ok &= (e + 0 >= K) || (gr[k].x >> 16) != 0;
                    ok &= (e + 1 >= K) || (gr[k].y >> 16) != 0;
                    ok &= (e + 2 >= K) || (gr[k].z >> 16) != 0;
                    ok &= (e + 3 >= K) || (gr[k].w >> 16) != 0;
                }
                if (__all(ok) || sp.give_up(a)) break;
                __builtin_amdgcn_s_sleep(4);
            }
            if (c == 0 && k0 == 0) eng_trace(a, 22 + oi);
            while (!vacated && fl[F_CONSDONE] < (uint32_t)(ncons * oi)) {   // every consumer of this CU has left the previous vector
                if (sp.give_up(a)) break;
                __builtin_amdgcn_s_sleep(1);
            }
            vacated = true;
#pragma unroll
            for (int k = 0; k < MYCH; ++k) {
                const int ch = c + (k0 + k) * ncons;
                if (ch < nch) {                          // 4 values = 8 bytes at element ch * 256 + lane * 4 (zeros beyond K)
                    uint2 pk;
                    pk.x = (gr[k].x & 0xffffu) | (gr[k].y << 16);
                    pk.y = (gr[k].z & 0xffffu) | (gr[k].w << 16);
                    reinterpret_cast<uint2*>(xbuf)[ch * 64 + lane] = pk;
                }
            }
        }
        while (!vacated && fl[F_CONSDONE] < (uint32_t)(ncons * oi)) {       // (a wave without a chunk of its own)
            if (sp.give_up(a)) break;
            __builtin_amdgcn_s_sleep(1);
        }
        if (c == ncons - 1)
            for (int e = nch * 256 + lane; e < JX * 512; e += 64) reinterpret_cast<bf16_t*>(xbuf)[e] = 0;   // columns beyond the last chunk
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add((__attribute__((address_space(3))) uint32_t*)(fl + F_XPARTS), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (c == 0) {
        while (fl[F_XPARTS] < (uint32_t)(ncons * (oi + 1))) {           // every wave's share of the vector is in LDS
            if (sp.give_up(a)) break;
            __builtin_amdgcn_s_sleep(1);
        }
        if (o.x_src != 0) {
            if (lane == 0) fl[F_GATHER] = 1;
            if (o.sum_out)                               // the vector as plain bf16 for later launches: this CU's share
                for (int l = lane; l < per && e0 + l < K; l += 64) o.sum_out[e0 + l] = reinterpret_cast<const bf16_t*>(xbuf)[e0 + l];
        }
        if (o.keep_raw) {                                // a later op's residual reads the un-normalised vector from LDS
#pragma unroll
            for (int j = 0; j < JX; ++j) reinterpret_cast<u32x4*>(xraw)[lane + 64 * j] = xb[lane + 64 * j];
        }
        if (o.gain) {
            // RMSNorm with the launch kernels' arithmetic: thread t of their 256 owns columns t, t + 256, ...; its sum of squares runs
            // over its columns in order, the four waves' DPP sums are added in wave order
            float sw[4] = {0.f, 0.f, 0.f, 0.f};
            u32x4 xr[JX];
#pragma unroll
            for (int j = 0; j < JX; ++j) xr[j] = xb[lane + 64 * j];
#pragma unroll
            for (int j = 0; j < JX; ++j) {
                float f[8];
                unpack8(xr[j], f);
#pragma unroll
                for (int e = 0; e < 8; ++e) sw[j & 3] += f[e] * f[e];
            }
            const float s0 = wave_sum(sw[0]), s1 = wave_sum(sw[1]), s2 = wave_sum(sw[2]), s3 = wave_sum(sw[3]);
            const float rinv = rsqrtf((s0 + s1 + s2 + s3) / (float)K + o.eps);
#pragma unroll
            for (int j = 0; j < JX; ++j) {
                float xf[8], gg[8];
                unpack8(xr[j], xf);
                unpack8(gv[j], gg);
#pragma unroll
                for (int e = 0; e < 8; ++e) xf[e] = gg[e] * bfround(xf[e] * rinv);
                const int vi = lane + 64 * j;
                xb[vi] = vi < KV ? pack8(xf) : u32x4{0u, 0u, 0u, 0u};
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if (lane == 0) fl[F_XREADY] = (uint32_t)(oi + 1);
        eng_trace(a, 4 + oi);
    }
    while (fl[F_XREADY] < (uint32_t)(oi + 1)) {
        if (sp.give_up(a)) break;
        __builtin_amdgcn_s_sleep(1);
    }
    // ---------------------------------------------------------------- this consumer's fills
    u32x4 xv[JX];                                        // the activation vector, this lane's columns: constant over the op
#pragma unroll
    for (int j = 0; j < JX; ++j) xv[j] = xb[lane + 64 * j];               // zero beyond K (the gather pads)
    const bool tail_masked = (JX * 64 != KV);            // the last vector row reaches beyond the row: the next row's bytes
    const int ngroups = (nfills + grp - 1) / grp;
    // residual values of this wave's rows, requested before the first fill is waited for: a global load in the epilogue of every
    // fill was a dependent trip of 1.5-2 us per fill on the wave (lane r: row r of the wave's k-th fill)
    constexpr int RESK = 8;
    float resv[RESK];
    if (o.epi == EPI_RESID && o.res_src == 0) {
#pragma unroll
        for (int k = 0; k < RESK; ++k) {
            const int row_f = (c + k * ncons) * rps;                       // grp == 1 for every op with a residual
            const int nr = nrows - row_f < rps ? nrows - row_f : rps;
            resv[k] = (c + k * ncons < ngroups && lane < nr) ? bf2f(o.res[row0 + row_f + lane]) : 0.f;
        }
    }
    int kk = 0;
    for (int gi = c; gi < ngroups; gi += ncons, ++kk) {
        float gate_val = 0.f;
        for (int fi = 0; fi < grp; ++fi) {
            const int f = gi * grp + fi;
            if (f >= nfills) break;
            const int g = gbase + f, slot = g % nslot;
            while (fl[F_READY + slot] != (uint32_t)(g + 1)) {
                if (sp.give_up(a)) break;
                __builtin_amdgcn_s_sleep(1);
            }
            const char* sb = ring + (size_t)slot * ENG_SLOT;
            const int row_f = f * rps;                                      // first row of the fill (CU-local)
            const int nrow_f = (a.dbg & 1) ? 0 : (nrows - row_f < rps ? nrows - row_f : rps);
            float myval = 0.f;                                              // lane r: value of the fill's row r
            for (int r = 0; r < nrow_f; ++r) {
                const u32x4* wr = reinterpret_cast<const u32x4*>(sb + (size_t)r * KV * 16);
                u32x4 wv[JX];
#pragma unroll
                for (int j = 0; j < JX; ++j) wv[j] = wr[lane + 64 * j];
                if (tail_masked && lane + 64 * (JX - 1) >= KV) wv[JX - 1] = u32x4{0u, 0u, 0u, 0u};
                float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < JX; ++j) {
                    // the launch kernels' thread t = w * 64 + lane owns column vectors t + 256 c: vector lane + 64 j belongs to wave
                    // w = j & 3 (block kernels) or to the one wave (wave kernel)
                    const int w = VW4 ? (j & 3) : 0;
                    float t = acc[w];
                    t = bf16_dot2(wv[j].x, xv[j].x, t);
                    t = bf16_dot2(wv[j].y, xv[j].y, t);
                    t = bf16_dot2(wv[j].z, xv[j].z, t);
                    t = bf16_dot2(wv[j].w, xv[j].w, t);
                    acc[w] = t;
                }
                float tot;
                if (VW4) tot = ((wave_sum(acc[0]) + wave_sum(acc[1])) + wave_sum(acc[2])) + wave_sum(acc[3]);
                else tot = wave_sum(acc[0]);
                if (lane == r) myval = tot;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // the slot's reads have returned: it may be refilled
            if (lane == 0) fl[F_DONE + slot] = (uint32_t)(g + 1);
            // ---------------------------------------------------------------- epilogue: lane r owns row r of the fill
            if (o.epi == EPI_SWIGLU) {
                // rows (2 j, 2 j + 1) = (gate_j, up_j): a pair is two one-row fills of a group, or neighbouring rows of a fill
                if (rps == 1) {
                    const float v = readlane_f(myval, 0);
                    if (fi == 0) gate_val = v;
                    else if (lane == 0) {
                        const int n = row0 + row_f - 1;                     // the gate row
                        const float gt = bfround(gate_val), up = bfround(v);
                        const bf16_t ov = f2bf(bfround(silu(gt)) * up);
                        if (o.out_dst == 0) o.out[n >> 1] = ov;
                        else __hip_atomic_store(o.ogran + (n >> 1), 0x10000u | ov, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                } else {
                    const float up = __shfl_down(myval, 1);
                    if (!(lane & 1) && lane + 1 < nrow_f) {
                        const int n = row0 + row_f + lane;
                        const float gt = bfround(myval), u = bfround(up);
                        const bf16_t ov = f2bf(bfround(silu(gt)) * u);
                        if (o.out_dst == 0) o.out[n >> 1] = ov;
                        else __hip_atomic_store(o.ogran + (n >> 1), 0x10000u | ov, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            } else if (lane < nrow_f) {
                const int n = row0 + row_f + lane;
                float v = bfround(myval);
                if (o.epi == EPI_RESID) {
                    float rv;
                    if (o.res_src == 1) rv = bf2f(reinterpret_cast<const bf16_t*>(xraw)[n]);
                    else if (kk < RESK) {
                        rv = resv[0];
#pragma unroll
                        for (int k = 1; k < RESK; ++k) rv = kk == k ? resv[k] : rv;
                    } else rv = bf2f(o.res[n]);
                    v = v + rv;
                }
                const bf16_t ov = f2bf(v);
                if (o.out_dst == 0) o.out[n] = ov;
                else if (o.out_dst == 1) __hip_atomic_store(o.ogran + n, 0x10000u | ov, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else {                                   // my partial sum of this launch's ar_k-th all-reduce: system scope (peer GPUs read it)
                    const uint32_t arn = seq0 + (uint32_t)o.ar_k, tag = (arn & 0x7fffu) | 0x8000u;
                    uint32_t* dst = reinterpret_cast<uint32_t*>(a.comm[a.tp_rank] + (size_t)(arn & (ENG_AR_BUFS - 1)) * ENG_AR_MAXLEN * 4);
                    __hip_atomic_store(dst + n, (tag << 16) | ov, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
        }
    }
    gbase += nfills;
    if (c == 0) eng_trace(a, 10 + oi);
    eng_trace(a, 16 + oi);
    if (lane == 0)                                       // done with this op's activation vector
        __hip_atomic_fetch_add((__attribute__((address_space(3))) uint32_t*)(fl + F_CONSDONE), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <bool VW4>
__device__ __forceinline__ void consume_op_jx(int jx, const EngArgs& a, const EngOp& o, int oi, int c, int ncons, int& gbase, char* ring,
                                              char* xbuf, char* xraw, uint32_t seq0, lds_u32* fl, Spin& sp) {
    switch (jx) {
#define EMU_J(J) case J: consume_op<J, VW4>(a, o, oi, c, ncons, gbase, ring, xbuf, xraw, seq0, fl, sp); break;
        EMU_J(1) EMU_J(2) EMU_J(3) EMU_J(4) EMU_J(5) EMU_J(6) EMU_J(7) EMU_J(8) EMU_J(9) EMU_J(10) EMU_J(11) EMU_J(12) EMU_J(13)
#undef EMU_J
        default: break;
    }
}

// ---------------------------------------------------------------------------------------------------------------- the launch
__global__ __launch_bounds__(ENG_THREADS, 1) void decode_engine_kernel(const EngArgs a_by_value) {
    // the descriptors are indexed at run time: read them where they are (the kernel-argument segment, scalar loads) -- indexing the
    // by-value parameter would make the compiler copy the whole block to scratch memory first
    const EngArgs& a = *(const EngArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* ring = smem;
    char* xbuf = smem + (size_t)a.nslot * ENG_SLOT;
    char* xraw = xbuf + a.xbytes;
    lds_u32* fl = (lds_u32*)(__attribute__((address_space(3))) char*)(xraw + a.xbytes);
    const uint32_t flags_addr = (uint32_t)(size_t)fl;
    if (tid < ENG_FLAGS / 4) fl[tid] = 0;
    // all-reduces this CU has been through before this launch (the same number on every CU of every rank: they run the same launches
    // in the same order): numbers this launch's all-reduces, i.e. their tags and comm arrays.  Every wave reads it ahead of the
    // barrier, one lane advances it behind the barrier.
    const uint32_t seq0 = a.seq ? a.seq[blockIdx.x] : 0u;
    __syncthreads();
    if (a.seq && a.n_ar && tid == 0) a.seq[blockIdx.x] = seq0 + (uint32_t)a.n_ar;
    const int nload = a.nload;                           // loader waves (1 or 2); the others consume
    if (wave < nload) { loader_wave(a, wave, nload, ring, flags_addr); return; }
    const int c = wave - nload, ncons = ENG_THREADS / 64 - nload;        // consumer 0 also gathers the activation vectors
    Spin sp; sp.t0 = wall_clock64();
    int gbase = 0;                                       // global fill index of the op's first fill on this CU
    for (int oi = 0; oi < a.nops; ++oi) {
        const EngOp& o = a.op[oi];
        const int jx = ((o.K >> 3) + 63) >> 6;
        if (o.vw == 4) consume_op_jx<true>(jx, a, o, oi, c, ncons, gbase, ring, xbuf, xraw, seq0, fl, sp);
        else consume_op_jx<false>(jx, a, o, oi, c, ncons, gbase, ring, xbuf, xraw, seq0, fl, sp);
    }
}

}  // namespace

size_t decode_engine_lds_bytes(const EngArgs& a) { return (size_t)a.nslot * ENG_SLOT + 2 * (size_t)a.xbytes + ENG_FLAGS; }

int launch_decode_engine(EngArgs a, hipStream_t s) {
    if (a.nops < 1 || a.nops > ENG_MAX_OPS || !a.err || a.ncu < 1 || a.nload < 0 || a.nload > 6) return -22;
    if (a.nload == 0) a.nload = 3;
    int kmax = 0;
    for (int i = 0; i < a.nops; ++i) {
        EngOp& o = a.op[i];
        if (!o.W || o.N < a.ncu * 2 || (o.K & 7) || (o.vw != 1 && o.vw != 4)) return -22;
        if (o.epi != EPI_NONE && o.epi != EPI_RESID && o.epi != EPI_SWIGLU) return -22;
        if (o.epi == EPI_SWIGLU && (o.N & 1)) return -22;
        if (o.K * 2 > ENG_MAXJ * 1024) return -95;       // rows longer than 13 KiB would need segments: not built (K <= 6656)
        if (o.x_src < 0 || o.x_src > 2 || o.out_dst < 0 || o.out_dst > 2 || (o.x_src == 0 ? !o.xg : !o.xgran)) return -22;
        if (o.out_dst == 0 ? !o.out : (o.out_dst == 1 && !o.ogran)) return -22;
        if (o.epi == EPI_RESID && o.res_src == 0 && !o.res) return -22;
        if (o.x_src == 2 || o.out_dst == 2) {            // an all-reduce of this launch: the ranks' comm areas and the counters
            if (a.tp_n < 1 || a.tp_n > EMU_ENG_MAX_RANKS || a.tp_rank < 0 || a.tp_rank >= a.tp_n || !a.seq || o.ar_k < 0 || o.ar_k >= a.n_ar) return -22;
            if ((o.x_src == 2 ? o.K : o.N) > ENG_AR_MAXLEN || a.n_ar > ENG_AR_BUFS / 2) return -22;
            for (int r = 0; r < a.tp_n; ++r)
                if (!a.comm[r]) return -22;
        }
        // fill geometry (the same for every CU): whole rows per 16 KiB slot; a (gate, up) pair of one-row fills goes to one consumer
        const int unit = o.epi == EPI_SWIGLU ? 2 : 1, units = o.N / unit, rowb = o.K * 2;
        o.q = units / a.ncu; o.rem = units % a.ncu;
        o.rps = ENG_SLOT / rowb;
        if (o.rps > 32) o.rps = 32;
        if (unit == 2 && o.rps > 1) o.rps &= ~1;
        o.fill_bytes = o.rps * rowb;
        o.ni = (o.fill_bytes + 1023) >> 10;
        o.grp = (unit == 2 && o.rps == 1) ? 2 : 1;
        o.nfills_lo = (o.q * unit + o.rps - 1) / o.rps;
        o.nfills_hi = ((o.q + 1) * unit + o.rps - 1) / o.rps;
        kmax = o.K > kmax ? o.K : kmax;
    }
    a.xbytes = ((kmax + 511) / 512) * 1024;              // whole 64-lane vector rows
    int nslot = (160 * 1024 - 2 * a.xbytes - ENG_FLAGS) / ENG_SLOT;
    if (nslot > 8) nslot = 8;
    if (nslot < 4) return -22;
    a.nslot = nslot;
    const size_t lds = decode_engine_lds_bytes(a);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(decode_engine_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return -5;
        attr_set = true;
    }
    hipLaunchKernelGGL(decode_engine_kernel, dim3(a.ncu), dim3(ENG_THREADS), lds, s, a);
    EMU_CHECK_LAUNCH();
    return 0;
}
