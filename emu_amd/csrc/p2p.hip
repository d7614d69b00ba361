// One-shot peer-to-peer all-reduce for the tensor-parallel decoder's small messages (decode: 13 KB per all-reduce, 120 per
// token) over xGMI, instead of 120 latency-bound RCCL collectives per token.
//
// Every rank owns one COMM BLOCK in its own HBM (uncached, exported with hipIpcGetMemHandle and mapped by every peer):
// two data slots of P2P_SLOT_BYTES and 64-bit flags per slot.  A message is cut into 8 KiB pieces, one workgroup and one
// 16-byte vector per lane each (13 KB = 2 workgroups, 256 KiB = 32), and every piece runs the protocol on its own counter,
// flag and slot region.  All-reduce number s of a piece (a device-side counter, so a captured hipGraph replays correctly)
// uses slot s & 1:
//   1. copy my partial vector into MY slot with system-scope (write-through) stores, wait for their acknowledgement, flag[slot] = s;
//   2. for every rank r in rank order (my own included): wait until r's flag[slot] >= s, read r's slot over xGMI with
//      system-scope loads, add in fp32 -- the same order on every rank, so all ranks hold bit-identical sums;
//   3. round to bf16 in place.
// Point-to-point xGMI means every rank reads the other N - 1 vectors directly (N - 1 links busy per rank, one hop, no
// ring): for 13 KB the cost is one flag round trip + one remote read, not 2 (N - 1) ring steps.
// Slot reuse is safe with two slots: a rank rewrites slot s & 1 at all-reduce s + 2, which it reaches only after finishing
// s + 1, which needed every peer's s + 1 vector, which a peer publishes only after its reads of all-reduce s are done.
// Every wait is BOUNDED in wall-clock time (s_memrealtime, 100 MHz): past the limit a give-up counter is raised, the kernel
// proceeds, and every later all-reduce skips its wait, so a dead peer or a platform problem costs one time-out, never a hung
// GPU.  The host reads the counter (emu_tp_p2p_giveups) after the self-test at initialisation -- a failed self-test keeps the
// path on RCCL -- and after every generation, where a non-zero count is an error (the sums are garbage).
// Messages larger than a slot go through it in chunks (tests on one GPU); production sends them to RCCL (engine.hip).
//
// Replaces (with emu_amd/tp.py) the reference's layer-placement "model parallel": Emu2/emu/mixin.py:14-85, chat.py:235-283.
#include <string.h>

#include "common.h"
#include "kernels.h"

namespace {

__device__ unsigned int g_p2p_giveups = 0;

struct P2pPeers {
    char* block[EMU_P2P_MAX_RANKS];                 // every rank's comm block as mapped in THIS process (own block included)
    int n, rank;
    long long limit_ticks;                          // wait bound in s_memrealtime ticks (100 MHz)
    int fenced;                                     // 1: the exchange is bracketed by two system-scope fences (see the kernel)
};

// piece g of a message: elements [g * P2P_PIECE, (g + 1) * P2P_PIECE), one 16-byte vector per lane of workgroup g, which runs
// its own instance of the protocol (own sequence counter, own flag per slot, own 8 KiB of each slot)
constexpr int P2P_PIECE = EMU_P2P_PIECE;
constexpr int P2P_PIECES = EMU_P2P_PIECES;

__device__ __forceinline__ unsigned long long* flag_of(char* block, int slot, int g) {
    return reinterpret_cast<unsigned long long*>(block + 2 * EMU_P2P_SLOT_BYTES) + slot * P2P_PIECES + g;
}

__global__ __launch_bounds__(512) void p2p_allreduce_kernel(P2pPeers p, unsigned long long* seq_counter, bf16_t* x, int n) {
    __shared__ unsigned long long s_sh;
    const int tid = threadIdx.x, g = blockIdx.x;
    const int e0 = g * P2P_PIECE + tid * 8;                                 // my 8 elements
    const bool full = e0 + 8 <= n, part = e0 < n;
    // my share of the partial vector is requested BEFORE the sequence counter makes its round trip through LDS (round 6: the two
    // loads were two dependent trips of a launch that is nothing but dependent trips)
    u32x4 mine = {0u, 0u, 0u, 0u};
    if (full) mine = ld16(x + e0);
    if (tid == 0) { s_sh = seq_counter[g] + 1; seq_counter[g] = s_sh; }   // launches of one stream are ordered: no race
    __syncthreads();
    const unsigned long long s = s_sh;
    const int slot = (int)(s & 1);
    const size_t off = (size_t)slot * EMU_P2P_SLOT_BYTES + (size_t)e0 * 2;
    // Two forms of the exchange, chosen per launch (P2pPeers::fenced):
    //  * fenced (the default of a fresh comm block): the HIP memory model's own release / acquire -- __threadfence_system() between
    //    the payload stores and the flag, and again between the last flag poll and the payload loads.  ~3.5 us each of a 6.8 us launch.
    //  * fence-free (round 5): no cache-wide operation at all.  The slot is written with system-scope (sc0 sc1, write-through) stores
    //    whose acknowledgement the wave waits for before the flag goes out, and read with system-scope loads, which no cache level
    //    serves -- MI355X_MICROARCH.md's "{sc0 sc1 stores and loads both sides}" form.  That rests on gfx950's write-through / ack
    //    behaviour, not on the memory model, and ranks that share one GPU (every run this repository has had) share an L2 and never
    //    exercise xGMI -- so the host switches it on only after a SOAK of this form passed on every rank of the job it is about to
    //    serve (emu_amd/llama.py::_init_p2p: hundreds of back-to-back all-reduces of sequence-dependent data, every word checked).
    // Both forms use the same sc0 sc1 stores and loads, so the fences are purely additional ordering.
    const __amdgpu_buffer_rsrc_t rmine = __builtin_amdgcn_make_buffer_rsrc(p.block[p.rank], 0, (uint32_t)(2 * EMU_P2P_SLOT_BYTES), 0x00020000);
    // ---- 1. publish my piece
    if (part) {
        u32x4 v;
        if (full) v = mine;
        else {
            bf16_t t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int j = 0; e0 + j < n; ++j) t[j] = x[e0 + j];
            v = *reinterpret_cast<u32x4*>(t);
        }
        __builtin_amdgcn_raw_buffer_store_b128(v, rmine, (uint32_t)off, 0, 17);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                      // this wave's share has reached the system's point of coherence
    if (p.fenced) __threadfence_system();                                  // release
    __syncthreads();
    if (tid == 0) __hip_atomic_store(flag_of(p.block[p.rank], slot, g), s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    // ---- 2. wait for every rank's piece of this sequence number
    if (tid < p.n) {
        unsigned long long* f = flag_of(p.block[tid], slot, g);
        const bool dead = __hip_atomic_load(&g_p2p_giveups, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < s) {
            if (dead || wall_clock64() - t0 > p.limit_ticks) { atomicAdd(&g_p2p_giveups, 1u); break; }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    if (p.fenced) __threadfence_system();                                  // acquire (every wave: the loads below are its own)
    __syncthreads();
    // ---- 3. sum in rank order, fp32
    if (part) {
        u32x4 v[EMU_P2P_MAX_RANKS];
#pragma unroll
        for (int r = 0; r < EMU_P2P_MAX_RANKS; ++r)
            if (r < p.n)
                v[r] = __builtin_amdgcn_raw_buffer_load_b128(
                    __builtin_amdgcn_make_buffer_rsrc(p.block[r], 0, (uint32_t)(2 * EMU_P2P_SLOT_BYTES), 0x00020000), (uint32_t)off, 0, 17);
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int r = 0; r < EMU_P2P_MAX_RANKS; ++r)
            if (r < p.n) {
                float f[8];
                unpack8(v[r], f);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += f[j];
            }
        if (full) st16(x + e0, pack8(acc));
        else for (int j = 0; e0 + j < n; ++j) x[e0 + j] = f2bf(acc[j]);
    }
}

}  // namespace

static size_t emu_p2p_engine_offset() {
    return (2 * EMU_P2P_SLOT_BYTES + 2 * P2P_PIECES * sizeof(unsigned long long) + 255) / 256 * 256;
}
constexpr int EMU_ENG_SEQ_WORDS = 1024;           // per-CU all-reduce counters of the engine (>= CUs of any device)

struct EmuP2p {
    char* mine = nullptr;                           // my comm block (device memory owned by this object)
    char* block[EMU_P2P_MAX_RANKS] = {};
    bool opened[EMU_P2P_MAX_RANKS] = {};
    unsigned long long* seq = nullptr;              // device-side sequence counters, one per piece (plain device memory)
    unsigned int* eng_seq = nullptr;                // decode_engine.hip: all-reduces every CU has been through (plain device memory)
    int n = 0, rank = 0;
    long long limit_ticks = 10LL * 100000000LL;     // 10 s
    int fenced = 1;                                 // emu_p2p_set_fenced: the memory-model form until the host's soak cleared the other
};

EmuP2p* emu_p2p_create(int rank, int n, void* handle64_out) {
    if (n < 1 || n > EMU_P2P_MAX_RANKS || rank < 0 || rank >= n) return nullptr;
    EmuP2p* p = new EmuP2p();
    p->n = n; p->rank = rank;
    const size_t bytes = emu_p2p_engine_offset() + EMU_P2P_ENG_BYTES;      // [slot 0 | slot 1 | flags | engine area (decode_engine.hip)]
    void* ptr = nullptr;
    // uncached (what RCCL uses for its own peer-visible buffers on gfx94x/gfx950), else fine-grained, else plain device memory:
    // the kernel's system-scope release / acquire pairs are sufficient for any of the three.
    if (hipExtMallocWithFlags(&ptr, bytes, hipDeviceMallocUncached) != hipSuccess) {
        (void)hipGetLastError();
        if (hipExtMallocWithFlags(&ptr, bytes, hipDeviceMallocFinegrained) != hipSuccess) {
            (void)hipGetLastError();
            if (hipMalloc(&ptr, bytes) != hipSuccess) { delete p; return nullptr; }
        }
    }
    p->mine = reinterpret_cast<char*>(ptr);
    if (hipMemset(p->mine, 0, bytes) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&p->seq), P2P_PIECES * 8) != hipSuccess ||
        hipMemset(p->seq, 0, P2P_PIECES * 8) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&p->eng_seq), EMU_ENG_SEQ_WORDS * 4) != hipSuccess ||
        hipMemset(p->eng_seq, 0, EMU_ENG_SEQ_WORDS * 4) != hipSuccess) { emu_p2p_destroy(p); return nullptr; }
    (void)hipDeviceSynchronize();
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle is 64 bytes");
    hipIpcMemHandle_t h;
    if (hipIpcGetMemHandle(&h, p->mine) != hipSuccess) { emu_p2p_destroy(p); return nullptr; }
    memcpy(handle64_out, &h, 64);
    p->block[rank] = p->mine;
    return p;
}

int emu_p2p_open(EmuP2p* p, const void* handles) {
    if (!p || !handles) return -22;
    for (int r = 0; r < p->n; ++r) {
        if (r == p->rank) continue;
        hipIpcMemHandle_t h;
        memcpy(&h, reinterpret_cast<const char*>(handles) + 64 * r, 64);
        void* ptr = nullptr;
        hipError_t e = hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) return (int)e;
        p->block[r] = reinterpret_cast<char*>(ptr);
        p->opened[r] = true;
    }
    return 0;
}

void emu_p2p_destroy(EmuP2p* p) {
    if (!p) return;
    for (int r = 0; r < p->n; ++r)
        if (p->opened[r]) (void)hipIpcCloseMemHandle(p->block[r]);
    if (p->mine) (void)hipFree(p->mine);
    if (p->seq) (void)hipFree(p->seq);
    if (p->eng_seq) (void)hipFree(p->eng_seq);
    delete p;
}

int emu_p2p_allreduce(EmuP2p* p, bf16_t* x, size_t n, hipStream_t s) {
    if (!p || !x) return -22;
    for (int r = 0; r < p->n; ++r)
        if (!p->block[r]) return -107;                                     // peers not mapped yet
    P2pPeers peers;
    for (int r = 0; r < EMU_P2P_MAX_RANKS; ++r) peers.block[r] = r < p->n ? p->block[r] : nullptr;
    peers.n = p->n; peers.rank = p->rank; peers.limit_ticks = p->limit_ticks; peers.fenced = p->fenced;
    const size_t chunk = EMU_P2P_SLOT_BYTES / 2;                           // elements per slot
    for (size_t o = 0; o < n; o += chunk) {
        const int m = (int)(n - o < chunk ? n - o : chunk);
        hipLaunchKernelGGL(p2p_allreduce_kernel, dim3((m + P2P_PIECE - 1) / P2P_PIECE), dim3(512), 0, s, peers, p->seq, x + o, m);
    }
    EMU_CHECK_LAUNCH();
    return 0;
}

bool emu_p2p_view(EmuP2p* p, char** block8, unsigned long long** seq, int* n, int* rank, long long* limit_ticks) {
    if (!p) return false;
    for (int r = 0; r < p->n; ++r)
        if (!p->block[r]) return false;
    for (int r = 0; r < EMU_P2P_MAX_RANKS; ++r) block8[r] = r < p->n ? p->block[r] : nullptr;
    *seq = p->seq; *n = p->n; *rank = p->rank; *limit_ticks = p->limit_ticks;
    return true;
}

bool emu_p2p_engine_view(EmuP2p* p, char** area8, unsigned int** seq, int* n, int* rank) {
    if (!p || !p->eng_seq) return false;
    for (int r = 0; r < p->n; ++r)
        if (!p->block[r]) return false;
    for (int r = 0; r < EMU_P2P_MAX_RANKS; ++r) area8[r] = r < p->n ? p->block[r] + emu_p2p_engine_offset() : nullptr;
    *seq = p->eng_seq; *n = p->n; *rank = p->rank;
    return true;
}

void emu_p2p_set_fenced(EmuP2p* p, int fenced) { if (p) p->fenced = fenced != 0; }
int emu_p2p_fenced(const EmuP2p* p) { return p ? p->fenced : -1; }
void emu_p2p_set_timeout_ms(EmuP2p* p, int ms) { if (p && ms > 0) p->limit_ticks = (long long)ms * 100000LL; }

unsigned int emu_p2p_giveups_read() {
    unsigned int v = 0;
    (void)hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_p2p_giveups), sizeof v);
    return v;
}
