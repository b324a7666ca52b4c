// One-shot collectives over peer memory for the tensor-parallel token pass (the 70B
// configuration of BASELINE.json).  The reference has no counterpart: it is one process on one core; the two exchanges
// wrap the row-parallel GEMVs `x += wo . xb` (/root/reference/llama2.f90:603-605) and `x += w2 . hb` (:618-620), the
// third returns the vocabulary-parallel logits (:634-636).
//
// Why not the library ring: the message is an E-vector (32 KB at 70B) 160 times per token.  xGMI is a full mesh of
// point-to-point links (7 per GPU), so every rank can hand its partial to every peer in ONE hop; a ring pays 2(P-1)
// dependent hops of pure latency for the same 32 KB.  ncclAllReduce stays selectable as the baseline (llmk.hip).
//
// Protocol (placement- and timing-independent, cdna_hip_programming.md Guideline 16 form R2 carried across devices):
//   * every rank owns an INBOX in its own HBM (fine-grained allocation), mapped into its peers (hipIpc between
//     processes, plain pointers / hipDeviceEnablePeerAccess inside one process);
//   * all-reduce k: thread i of rank r stores the 8-byte granule {epoch, partial_r[i]} into slot r of every peer's
//     inbox with ONE system-scope store, then adds the P partials of element i in RANK ORDER (its own from registers,
//     the others as their granules' tags turn to `epoch`) and applies the residual x[i] += sum.  Every rank adds the
//     same numbers in the same order: the replicated residual stream stays bit-identical on all ranks, run to run;
//   * a granule is one naturally aligned 8-byte store: tag and value arrive together, no flag, no fence;
//   * two inbox halves alternate by call parity.  Rank A can only start call k+2 after it finished k+1, which needed
//     rank B's k+1 granules, which B sends only after it finished reading call k: a half is never overwritten while a
//     peer still reads it.  (Checked exhaustively for P = 2, 3 over every interleaving of the ranks' sends and reads,
//     all-gather included, by tests/tp_cpu_model.py::check_interleavings; with ONE half the same checker finds the
//     overwrite.)  Element i's granules are touched by thread i of each rank only, so the argument is per element and
//     needs no ordering between the threads of a launch;
//   * epochs are unique per (token serial, call index) and never 0;
//   * every spin is bounded by WALL-CLOCK time (the constant-rate counter wall_clock64 reads; the limit in its ticks comes
//     from the host, TpPeers::timeout_ticks): a spin COUNT means a different time on every topology -- eight rank
//     processes time-slicing one GPU stretch a poll from 1 us to milliseconds.  A timeout raises the ctx's sticky error
//     word (LLMK_E_TIMEOUT) instead of hanging: 0x3 in bits 28-31, the rank that was waited for in bits 24-27, the low 24
//     bits of the epoch below (llmk.hip decodes it into serial / exchange index for the message).
//
// JIT (measurement/verification only: llmk_tp_p2p_stress): wave-uniform pseudo-random s_sleep bursts before the sends,
// between them, before the reads and between those -- ranks and waves drift apart by up to a whole exchange, which is
// what a slower link or a time-sliced GPU does to them.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace llmk {

constexpr int TP_MAX_RANKS = 8;

struct TpPeers {
    unsigned long long* inbox[TP_MAX_RANKS];   // rank r's inbox as THIS device addresses it (inbox[me] = local)
    unsigned long long timeout_ticks;          // bound of every spin, in wall_clock64 ticks (llmk.hip: 20 s unless LLMK_TP_TIMEOUT_MS)
};

// inbox layout (granules): all-reduce [2 halves][P slots][E]  |  all-gather [2 halves][V]
__host__ __device__ inline size_t tp_inbox_granules(int P, int E, int V) { return (size_t)2 * P * E + (size_t)2 * V; }

__device__ __forceinline__ unsigned tp_err_code(int src, unsigned epoch) { return 0x30000000u | ((unsigned)src << 24) | (epoch & 0xffffffu); }

__device__ __forceinline__ void tp_send(unsigned long long* p, unsigned epoch, float v) {
    __hip_atomic_store(p, ((unsigned long long)epoch << 32) | (unsigned long long)__float_as_uint(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
}
// returns false on timeout (err word raised); src = the rank whose granule is awaited (for the error word)
__device__ __forceinline__ bool tp_recv(const unsigned long long* p, unsigned epoch, float* v, unsigned* err, unsigned long long limit, int src) {
    unsigned long long t0 = 0;
    for (unsigned spin = 0;; ++spin) {
        const unsigned long long g = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((unsigned)(g >> 32) == epoch) { *v = __uint_as_float((unsigned)g); return true; }
        if ((spin & 255) == 255) {
            if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
            const unsigned long long now = wall_clock64();
            if (t0 == 0) t0 = now;
            else if (now - t0 > limit) {
                __hip_atomic_store(err, tp_err_code(src, epoch), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
        __builtin_amdgcn_s_sleep(2);
    }
}

// wave-uniform jitter: 0..31 bursts of s_sleep 16 (64 clocks each) = up to ~15 us, from an LCG the wave carries along
__device__ __forceinline__ void tp_jitter(unsigned& s) {
    s = s * 1664525u + 1013904223u;
    const unsigned n = __builtin_amdgcn_readfirstlane((s >> 24) & 31u);
    for (unsigned k = 0; k < n; ++k) __builtin_amdgcn_s_sleep(16);
}
__device__ __forceinline__ unsigned tp_jitter_seed(unsigned jseed, int me, unsigned epoch) {
    return __builtin_amdgcn_readfirstlane(jseed ^ ((unsigned)me * 0x9E3779B1u) ^ (epoch * 0x85EBCA77u) ^ ((blockIdx.x * 8u + (threadIdx.x >> 6)) * 0xC2B2AE3Du));
}

// x[i] += sum over ranks of part_r[i]            (the residual adds of llama2.f90:603-605 and :618-620)
// tokpos[2] = token serial (device memory: the launch is replayable from a hipGraph); call = index of this exchange
// inside the token pass (2 per layer), ncalls = exchanges per token (epoch = serial * (ncalls + 1) + call + 1)
template <bool JIT>
__global__ __launch_bounds__(256) void tp_allreduce_add_kernel(TpPeers peers, const float* __restrict__ part, float* __restrict__ x,
                                                               const int* __restrict__ tokpos, int call, int ncalls, int me, int P,
                                                               int E, unsigned* err, unsigned jseed) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= E) return;
    const unsigned epoch = (unsigned)tokpos[2] * (unsigned)(ncalls + 1) + (unsigned)call + 1u;
    const size_t half = (size_t)(call & 1) * P * E;
    const float mine = part[i];
    unsigned js = 0;
    if constexpr (JIT) { js = tp_jitter_seed(jseed, me, epoch); tp_jitter(js); }
    for (int r = 1; r < P; ++r) {                 // start with the next rank: the P ranks do not all hit the same peer first
        const int dst = (me + r) % P;
        tp_send(peers.inbox[dst] + half + (size_t)me * E + i, epoch, mine);
        if constexpr (JIT) tp_jitter(js);
    }
    float sum = 0.f;
    bool ok = true;
    for (int r = 0; r < P; ++r) {                 // rank order: identical sums on every rank
        float v = mine;
        if (r != me) ok = tp_recv(peers.inbox[me] + half + (size_t)r * E + i, epoch, &v, err, peers.timeout_ticks, r) && ok;
        sum += v;
        if constexpr (JIT) tp_jitter(js);
    }
    if (ok) x[i] += sum;
}

// logits[0..V) = concatenation of the ranks' Vl-row slices                     (classifier, llama2.f90:634-636)
template <bool JIT>
__global__ __launch_bounds__(256) void tp_allgather_kernel(TpPeers peers, float* logits, const int* __restrict__ tokpos,
                                                           int ncalls, int me, int P, int E, int V, unsigned* err, unsigned jseed) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int Vl = V / P, owner = j < V ? j / Vl : -1;
    const unsigned serial = (unsigned)tokpos[2];
    const unsigned epoch = serial * (unsigned)(ncalls + 1) + (unsigned)ncalls + 1u;
    const size_t base = (size_t)2 * P * E + (size_t)(serial & 1) * V;
    unsigned js = 0;
    if constexpr (JIT) { js = tp_jitter_seed(jseed, me, epoch); tp_jitter(js); }
    // Two separate statements, NOT if/else: a wave can hold owners and receivers side by side, and with an if/else the
    // hardware may run the spinning side first -- the owners' stores then never issue, and the peer's mirror-image wave
    // waits for them while this one waits for the peer's (observed on MI355X: the first rank to arrive hung).  Every
    // lane's sends are issued before any lane starts to wait.
    if (owner == me) {
        const float v = logits[j];
        for (int r = 1; r < P; ++r) tp_send(peers.inbox[(me + r) % P] + base + j, epoch, v);
    }
    if constexpr (JIT) tp_jitter(js);
    if (owner >= 0 && owner != me) {
        float v;
        if (tp_recv(peers.inbox[me] + base + j, epoch, &v, err, peers.timeout_ticks, owner)) logits[j] = v;
    }
}

}  // namespace llmk
