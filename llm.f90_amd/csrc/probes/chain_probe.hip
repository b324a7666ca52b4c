// Hardware probe (not product code): what does a DEPENDENT chain of latency-bound GEMV-like kernels cost on gfx950 --
// the shape of the multi-kernel token pass (csrc/kernels.h) and of a Llama-2-70B tensor-parallel rank's layer in particular
// (60 MB per layer = 9.4 us of HBM at 6.4 TB/s, 35.9 us measured as five kernels) -- where does the time of
// such a layer go, and does it help to have kernel k+1 ALREADY RUNNING while kernel k finishes (round-3 verdict item 4: "measure,
// don't estimate" a fused rank path)?
//   serial     all kernels on ONE stream (what llmk.hip's eager path does); with per-workgroup time stamps in a separate pass:
//              end of kernel k -> first start of k+1 (the kernel boundary), spread of the workgroups' starts, start -> input
//              staged in LDS, staged -> end
//   graph      the same chain captured into one hipGraph
//   fused      SEVERAL stages per launch (2 / 5 = a layer / 10 / 20): workgroups [first[r], first[r+1]) play stage r; they are
//              dispatched in blockIdx order, so all of stage r is on the chip before the first workgroup of stage r+1 starts to
//              wait.  The dependency is a device-side completion counter: every workgroup of stage r adds 1 when its outputs are
//              out; stage r+1 requests its first PF weight vectors per lane (weights do not depend on r), THEN waits for the
//              counter to reach r's grid size (one poller per workgroup, poll_sleep x s_sleep 8 between polls), then stages its
//              input vector and streams the rest.  Ways to make r's outputs visible across the XCDs' L2s:
//                fence   plain stores + release (buffer_wbl2 sc1) on the counter, acquire (buffer_inv sc1) behind the wait
//                sc1     outputs stored and inputs loaded with the sc1 bit (write-through / miss-always, as the persistent
//                        kernel's granules), counter relaxed behind s_waitcnt vmcnt(0): no cache-wide operation
//                sc1 stores + buffer_inv   outputs stored sc1, inputs loaded plainly behind one buffer_inv sc1 per workgroup
//   two streams (CHAIN_TWO_STREAMS=1)   kernels alternate between TWO streams (k, k+2, .. on A; k+1, k+3, .. on B: no queue-level
//              dependency between neighbours), the same counters between them; eager, and as one hipGraph with two branches
// A stage = a [rows x row_bytes] matrix streamed once by rows / rpw waves (one workgroup = 4 waves), each wave folding its
// slice against the staged input vector (LDS) into rpw output words: integer arithmetic, so the final vector of every mode
// must equal the serial one bit for bit -- a stale input anywhere in the 400-kernel chain changes it.
// Every wait is bounded by wall-clock time (5 ms): a deadlock (two neighbours that do not fit the chip together, a graph whose
// branches are run one after the other) ends as an error word, not as a hang.  Output: one JSON object per line.
//   hipcc --offload-arch=gfx950 -O3 chain_probe.hip -o chain_probe && ./chain_probe [layers=80] [reps=4]
//   CHAIN_FUSED=n runs only the first n fused variants.
// Measured on MI355X (profiles/r04_chain_probe_*.jsonl): a kernel boundary is 1.2-1.4 us, the workgroups of a
// launch start within 0.2 us, staging 32 KB of input costs 1.1-1.6 us, a 33 MB stage then streams at 6.7-7.5 TB/s; every
// device-side dependency tried is SLOWER than the boundary it replaces (fused layer 70-85 us against 29-30 us serial; two
// streams 85-165 us; the two-branch graph runs its branches one after the other).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("{\"error\":\"%s\",\"line\":%d}\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

typedef unsigned v4u __attribute__((ext_vector_type(4)));

struct StageArgs {
    const v4u* W;                 // this stage's matrix of this layer
    unsigned nv;                  // 16-byte vectors per wave (rpw * row_bytes / 16)
    int rpw;                      // outputs per wave
    int nwaves;                   // waves with work (rows / rpw)
    const unsigned* in;           // input vector, n_in words (n_in % 4 == 0, n_in >= 256)
    int n_in;
    unsigned* out;                // nwaves * rpw words
    unsigned* flags;              // completion counters
    int wait_idx;                 // counter of the producer (-1: none)
    unsigned wait_count;          // = the producer's grid size
    int done_idx;                 // this kernel's counter (-1: none)
    unsigned long long limit;     // wall_clock64 ticks
    unsigned* err;
    unsigned long long* stamps;   // [workgroup][4] {start, end of wait, end, input staged} of this kernel (null: none)
    int poll_sleep;               // s_sleep 8 units between polls of the counter
};

enum { M_PLAIN = 0, M_FENCE = 1, M_SC1 = 2, M_HYB = 3 };   // M_HYB: outputs stored sc1, counter relaxed; consumer: buffer_inv sc1 behind the wait, plain loads

__device__ __forceinline__ v4u ld_nt(const v4u* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

template <int MODE, int PF>
__device__ __forceinline__ void stage_body(const StageArgs& a, int bid, unsigned* xs) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int gw = bid * 4 + wid;
    if (a.stamps && tid == 0) a.stamps[bid * 4 + 0] = wall_clock64();
    const bool live = gw < a.nwaves;
    const v4u* w = a.W + (size_t)(live ? gw : 0) * a.nv;
    const unsigned nv = a.nv;
    // ---- the first PF vectors per lane: requested before anything that depends on the producer ------------------------------
    v4u pf[PF > 0 ? PF : 1];
#pragma unroll
    for (int j = 0; j < PF; ++j) {
        const unsigned i = min((unsigned)lane + 64u * j, nv - 1);
        pf[j] = ld_nt(w + i);
    }
    // ---- wait for the producer ----------------------------------------------------------------------------------------------
    if (MODE != M_PLAIN && a.wait_idx >= 0) {
        if (tid == 0) {
            unsigned long long t0 = 0;
            for (unsigned spin = 0;; ++spin) {
                if (__hip_atomic_load(a.flags + a.wait_idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= a.wait_count) break;
                if ((spin & 63) == 63) {
                    const unsigned long long now = wall_clock64();
                    if (t0 == 0) t0 = now;
                    else if (now - t0 > a.limit) { __hip_atomic_store(a.err, 0x80000000u | (unsigned)a.wait_idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                }
                for (int z = 0; z < a.poll_sleep; ++z) __builtin_amdgcn_s_sleep(8);
            }
        }
        __syncthreads();
        if (MODE == M_FENCE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (MODE == M_HYB) asm volatile("buffer_inv sc1" ::: "memory");
    }
    if (a.stamps && tid == 0) a.stamps[bid * 4 + 1] = wall_clock64();
    // ---- stage the input vector: eight of a thread's loads in flight at once (all of them up to 8192 words) -------------------
    {
        const int n4 = a.n_in >> 2;
        const __amdgpu_buffer_rsrc_t rs = rsrc(a.in, (unsigned)a.n_in * 4u);
        for (int base = 0; base < n4; base += 8 * 256) {
            v4u v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = min(base + tid + u * 256, n4 - 1);
                if (MODE == M_SC1) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, i * 16, 0, 16);   // aux 16 = sc1
                else v[u] = reinterpret_cast<const v4u*>(a.in)[i];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = base + tid + u * 256;
                if (i < n4) reinterpret_cast<v4u*>(xs)[i] = v[u];
            }
        }
    }
    __syncthreads();
    if (a.stamps && tid == 0) a.stamps[bid * 4 + 3] = wall_clock64();
    // ---- stream the slice ------------------------------------------------------------------------------------------------------
    unsigned acc = 0;
    int xi = lane * 4;                       // word index into xs: advances 256 per vector column, wraps at n_in
    const int n_in = a.n_in;
    auto fold = [&](const v4u& q, bool on) {
        const v4u x = *reinterpret_cast<const v4u*>(xs + xi);
        if (on) acc += (q.x ^ x.x) + (q.y ^ x.y) * 3u + (q.z ^ x.z) * 5u + (q.w ^ x.w) * 7u;
        xi += 256; if (xi >= n_in) xi -= n_in;
    };
#pragma unroll
    for (int j = 0; j < PF; ++j) fold(pf[j], (unsigned)lane + 64u * j < nv);
    unsigned i = (unsigned)lane + 64u * PF;
    for (; i + 64u * 7 < nv; i += 64u * 8) {
        v4u q[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) q[j] = ld_nt(w + i + 64u * j);
#pragma unroll
        for (int j = 0; j < 8; ++j) fold(q[j], true);
    }
    for (; i + 64u < nv; i += 64u * 2) {
        v4u q[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) q[j] = ld_nt(w + i + 64u * j);
#pragma unroll
        for (int j = 0; j < 2; ++j) fold(q[j], true);
    }
    for (; i < nv; i += 64u) fold(ld_nt(w + i), true);
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (live && lane < a.rpw) {
        const unsigned val = acc * 2654435761u + (unsigned)lane * 40503u;
        const int o = gw * a.rpw + lane;
        if (MODE == M_SC1 || MODE == M_HYB) __builtin_amdgcn_raw_buffer_store_b32(val, rsrc(a.out, (unsigned)(a.nwaves * a.rpw) * 4u), o * 4, 0, 16);
        else a.out[o] = val;
    }
    // ---- signal ----------------------------------------------------------------------------------------------------------------
    if (MODE != M_PLAIN && a.done_idx >= 0) {
        if (MODE == M_SC1 || MODE == M_HYB) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            if (MODE == M_FENCE) __hip_atomic_fetch_add(a.flags + a.done_idx, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            else __hip_atomic_fetch_add(a.flags + a.done_idx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (a.stamps && tid == 0) a.stamps[bid * 4 + 2] = wall_clock64();
}

template <int MODE, int PF>
__global__ __launch_bounds__(256) void stage_kernel(StageArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned xs[];
    stage_body<MODE, PF>(a, blockIdx.x, xs);
}

// SEVERAL stages in ONE launch: workgroups [first[r], first[r+1]) play stage r.  Workgroups are dispatched in blockIdx order,
// so every workgroup of stage r is on the chip (or done) before the first of stage r+1 starts to wait for stage r's counter.
constexpr int MAX_FUSED = 20;
struct FusedArgs { StageArgs st[MAX_FUSED]; int first[MAX_FUSED + 1]; int n; };
template <int MODE, int PF>
__global__ __launch_bounds__(256) void fused_kernel(FusedArgs f) {
    extern __shared__ __attribute__((aligned(16))) unsigned xs[];
    int r = 0;
    while (r + 1 < f.n && (int)blockIdx.x >= f.first[r + 1]) ++r;
    stage_body<MODE, PF>(f.st[r], (int)blockIdx.x - f.first[r], xs);
}

__global__ void fill_kernel(unsigned* p, size_t n, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u ^ seed ^ (unsigned)(i >> 32) * 40503u;
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = x;
    }
}

struct Stage { const char* name; int rows, row_bytes, rpw, n_in; };
// one tensor-parallel rank of Llama-2-70B q4_0 (E 8192, 8 query heads + 1 kv head of 128, hidden slice 3584): DESIGN.md section 5
static const Stage RANK70[5] = {
    {"qkv", 1280, 4608, 1, 8192},      // 5.9 MB
    {"attn", 1024, 1024, 16, 1280},    // 1.0 MB (K/V of one kv head at a few hundred positions)
    {"wo", 8192, 576, 8, 1024},        // 4.7 MB
    {"w13", 7168, 4608, 4, 8192},      // 33.0 MB
    {"w2", 8192, 2016, 4, 3584},       // 16.5 MB
};
// Llama-2-7B q4_0, whole model on one GPU (the multi-kernel path of config 3)
static const Stage L7B[5] = {
    {"qkv", 12288, 2304, 8, 4096},     // 28.3 MB
    {"attn", 4096, 2048, 16, 4096},    // 8.4 MB
    {"wo", 4096, 2304, 4, 4096},       // 9.4 MB
    {"w13", 22016, 2304, 8, 4096},     // 50.7 MB
    {"w2", 4096, 6192, 2, 11008},      // 25.4 MB
};

// the rank's layer again with fewer, fatter waves (about one workgroup per CU) and with more, thinner ones: how the time of a
// stage depends on its grid
static const Stage RANK70_FAT[5] = {
    {"qkv", 1280, 4608, 2, 8192}, {"attn", 1024, 1024, 16, 1280}, {"wo", 8192, 576, 16, 1024}, {"w13", 7168, 4608, 7, 8192}, {"w2", 8192, 2016, 8, 3584},
};
static const Stage RANK70_THIN[5] = {
    {"qkv", 1280, 4608, 1, 8192}, {"attn", 1024, 1024, 8, 1280}, {"wo", 8192, 576, 4, 1024}, {"w13", 7168, 4608, 2, 8192}, {"w2", 8192, 2016, 2, 3584},
};

template <int MODE, int PF>
static void launch(hipStream_t st, const Stage& s, const StageArgs& a) {
    const int grid = (a.nwaves + 3) / 4;
    hipLaunchKernelGGL((stage_kernel<MODE, PF>), dim3(grid), dim3(256), (size_t)s.n_in * 4, st, a);
}
static void launch_mp(int mode, int pf, hipStream_t st, const Stage& s, const StageArgs& a) {
#define L_(M_, P_) if (mode == M_ && pf == P_) return launch<M_, P_>(st, s, a);
    L_(M_PLAIN, 2) L_(M_PLAIN, 0) L_(M_SC1, 0) L_(M_SC1, 2) L_(M_FENCE, 0) L_(M_FENCE, 2)
#undef L_
    printf("{\"error\":\"no instantiation\",\"mode\":%d,\"pf\":%d}\n", mode, pf); exit(1);
}
static void launch_fused(int mode, int pf, hipStream_t st, const FusedArgs& f, size_t lds) {
    const dim3 grid(f.first[f.n]), block(256);
#define L_(M_, P_) if (mode == M_ && pf == P_) { hipLaunchKernelGGL((fused_kernel<M_, P_>), grid, block, lds, st, f); return; }
    L_(M_FENCE, 0) L_(M_SC1, 0) L_(M_SC1, 2) L_(M_HYB, 0) L_(M_HYB, 2)
#undef L_
    printf("{\"error\":\"no instantiation\",\"mode\":%d,\"pf\":%d}\n", mode, pf); exit(1);
}

int main(int argc, char** argv) {
    const int L = argc > 1 ? atoi(argv[1]) : 80;           // layers per pass (the weights of L layers are distinct: no cache reuse)
    const int REPS = argc > 2 ? atoi(argv[2]) : 4;
    int ncu = 0; hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0)); ncu = prop.multiProcessorCount;
    hipStream_t sa, sb; CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    hipEvent_t ej; CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const unsigned long long limit = 500000ull;             // 5 ms of the 100 MHz constant clock
    for (int shape = 0; shape < 4; ++shape) {
        const Stage* S = shape == 0 ? RANK70 : shape == 1 ? L7B : shape == 2 ? RANK70_FAT : RANK70_THIN;
        const char* sname = shape == 0 ? "llama2-70b rank of 8" : shape == 1 ? "llama2-7b" : shape == 2 ? "llama2-70b rank of 8, fat waves" : "llama2-70b rank of 8, thin waves";
        const int Ls = shape != 1 ? L : (L < 32 ? L : 32);
        size_t layer_bytes = 0, off[5], max_lds = 0;
        for (int s = 0; s < 5; ++s) { off[s] = layer_bytes; layer_bytes += (size_t)S[s].rows * S[s].row_bytes; if ((size_t)S[s].n_in * 4 > max_lds) max_lds = (size_t)S[s].n_in * 4; }
        char* W; CK(hipMalloc(&W, layer_bytes * Ls));
        hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, sa, (unsigned*)W, layer_bytes * Ls / 4, 12345u + shape);
        unsigned* vec[5];                                   // vec[s] = output of stage s (input of stage s + 1; vec[4] feeds qkv)
        for (int s = 0; s < 5; ++s) { CK(hipMalloc(&vec[s], 131072)); }
        const int nk = Ls * 5;                              // kernels per pass
        const int nflags = (REPS + 1) * nk + 8;
        unsigned *flags, *err; CK(hipMalloc(&flags, (size_t)nflags * 4)); CK(hipMalloc(&err, 4));
        constexpr int MAXG = 1024;                            // workgroups per kernel, at most
        unsigned long long* stamps; CK(hipMalloc(&stamps, (size_t)nk * MAXG * 4 * 8));
        std::vector<double> hx((size_t)nk * 4);      // per kernel: spread of the workgroups' starts, mean start -> staged, mean staged -> end, workgroups
        std::vector<unsigned long long> hst((size_t)nk * 3), hraw((size_t)nk * MAXG * 4);
        int poll_sleep = 1;
        CK(hipStreamSynchronize(sa));
        std::vector<unsigned> ref(8192), got(8192);
        double serial_us = 0;

        auto args_of = [&](int rep, int l, int s, bool chain, bool stamp) {
            StageArgs a;
            a.W = reinterpret_cast<const v4u*>(W + layer_bytes * l + off[s]);
            a.rpw = S[s].rpw; a.nwaves = S[s].rows / S[s].rpw; a.nv = (unsigned)((size_t)S[s].rpw * S[s].row_bytes / 16);
            a.in = vec[(s + 4) % 5]; a.n_in = S[s].n_in; a.out = vec[s];
            a.flags = flags; a.limit = limit; a.err = err;
            const int k = (rep * Ls + l) * 5 + s;           // index of this kernel in the chain
            a.done_idx = chain ? k : -1;
            a.wait_idx = (chain && k > 0) ? k - 1 : -1;
            const int ps = (s + 4) % 5;
            a.wait_count = (unsigned)((S[ps].rows / S[ps].rpw + 3) / 4);
            a.stamps = stamp ? stamps + (size_t)(l * 5 + s) * MAXG * 4 : nullptr;
            a.poll_sleep = poll_sleep;
            return a;
        };
        auto reset = [&]() {
            CK(hipMemsetAsync(flags, 0, (size_t)nflags * 4, sa)); CK(hipMemsetAsync(err, 0, 4, sa));
            hipLaunchKernelGGL(fill_kernel, dim3(32), dim3(256), 0, sa, vec[4], (size_t)8192, 777u);      // x of the first layer
            CK(hipStreamSynchronize(sa));
        };
        auto stamp_init = [&]() { CK(hipMemset(stamps, 0, (size_t)nk * MAXG * 4 * 8)); };
        auto stamp_reduce = [&]() {   // per kernel: first start, last end-of-wait, last end over its workgroups
            CK(hipMemcpy(hraw.data(), stamps, (size_t)nk * MAXG * 4 * 8, hipMemcpyDeviceToHost));
            for (int k = 0; k < nk; ++k) {
                unsigned long long st = ~0ull, we = 0, en = 0, ls = 0;
                double a = 0, b2 = 0; int n = 0;
                for (int b = 0; b < MAXG; ++b) {
                    const unsigned long long* q = &hraw[((size_t)k * MAXG + b) * 4];
                    if (q[0] == 0) continue;
                    if (q[0] < st) st = q[0];
                    if (q[0] > ls) ls = q[0];
                    if (q[1] > we) we = q[1];
                    if (q[2] > en) en = q[2];
                    a += (double)(q[3] - q[1]); b2 += (double)(q[2] - q[3]); ++n;
                }
                hst[3 * k] = st; hst[3 * k + 1] = we; hst[3 * k + 2] = en;
                hx[4 * k] = (double)(ls - st) / 100.0; hx[4 * k + 1] = n ? a / n / 100.0 : 0; hx[4 * k + 2] = n ? b2 / n / 100.0 : 0; hx[4 * k + 3] = n;
            }
        };
        auto report = [&](const char* how, int mode, int pf, int per_launch, float ms, bool is_ref) {
            CK(hipMemcpy(got.data(), vec[4], 8192 * 4, hipMemcpyDeviceToHost));
            unsigned e = 0; CK(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost));
            if (is_ref) ref = got;
            const bool same = memcmp(ref.data(), got.data(), 8192 * 4) == 0;
            const double us = ms * 1000.0 / ((double)REPS * Ls);
            if (is_ref) serial_us = us;
            printf("{\"probe\":\"chain\",\"shape\":\"%s\",\"how\":\"%s\",\"visibility\":\"%s\",\"prefetch_vectors\":%d,\"stages_per_launch\":%d,\"layers\":%d,\"reps\":%d,"
                   "\"poll_sleep\":%d,\"layer_MB\":%.1f,\"us_per_layer\":%.2f,\"vs_serial\":%.3f,\"hbm_floor_us\":%.2f,\"equal_to_serial\":%s,\"err\":\"0x%x\",\"cus\":%d}\n",
                   sname, how, mode == M_PLAIN ? "stream order" : mode == M_FENCE ? "fence" : mode == M_SC1 ? "sc1" : "sc1 stores + buffer_inv", pf, per_launch, Ls, REPS, poll_sleep, layer_bytes / 1e6, us,
                   serial_us > 0 ? us / serial_us : 1.0, layer_bytes / 8e12 * 1e6, same ? "true" : "false", e, ncu);
            fflush(stdout);
        };
        // per stage, averaged over the layers 4 .. Ls-5 of one stamped pass: from the producer's last end to this kernel's first
        // start (gap), from there to the end of its last wait (wait), from there to its last end (body); 100 MHz ticks -> us
        auto timeline = [&](const char* how, int mode, int pf, int per_launch) {
            stamp_reduce();
            double gap[5] = {0}, wait[5] = {0}, body[5] = {0}, spread[5] = {0}, stg[5] = {0}, strm[5] = {0}; int n = 0;
            for (int l = 4; l < Ls - 4; ++l, ++n)
                for (int s = 0; s < 5; ++s) {
                    const int k = l * 5 + s;
                    const double pe = (double)hst[3 * (k - 1) + 2], st = (double)hst[3 * k], we = (double)hst[3 * k + 1], en = (double)hst[3 * k + 2];
                    spread[s] += hx[4 * k]; stg[s] += hx[4 * k + 1]; strm[s] += hx[4 * k + 2];
                    gap[s] += (st - pe) / 100.0; wait[s] += ((we > pe ? we : pe) - (st > pe ? st : pe)) / 100.0; body[s] += (en - (we > pe ? we : pe)) / 100.0;
                }
            printf("{\"probe\":\"chain-timeline\",\"shape\":\"%s\",\"how\":\"%s\",\"visibility\":\"%s\",\"prefetch_vectors\":%d,\"stages_per_launch\":%d,\"stages\":[",
                   sname, how, mode == M_PLAIN ? "stream order" : mode == M_FENCE ? "fence" : "sc1", pf, per_launch);
            for (int s = 0; s < 5; ++s)
                printf("%s{\"name\":\"%s\",\"first_start_after_producer_end_us\":%.2f,\"ready_after_producer_end_us\":%.2f,\"body_us\":%.2f,"
                       "\"workgroup_start_spread_us\":%.2f,\"mean_wait_end_to_input_staged_us\":%.2f,\"mean_staged_to_end_us\":%.2f}", s ? "," : "", S[s].name,
                       gap[s] / n, wait[s] / n, body[s] / n, spread[s] / n, stg[s] / n, strm[s] / n);
            printf("]}\n"); fflush(stdout);
        };

        // ---- serial, one stream (warm-up pass first) -----------------------------------------------------------------------------
        for (int pf : {2, 0}) {
            for (int timed = 0; timed < 2; ++timed) {
                reset();
                CK(hipEventRecord(e0, sa));
                for (int r = 0; r < (timed ? REPS : 1); ++r)
                    for (int l = 0; l < Ls; ++l)
                        for (int s = 0; s < 5; ++s) launch_mp(M_PLAIN, pf, sa, S[s], args_of(r, l, s, false, false));
                CK(hipEventRecord(e1, sa)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (timed) report("serial", M_PLAIN, pf, 1, ms, pf == 2);
            }
            reset(); stamp_init();
            for (int l = 0; l < Ls; ++l)
                for (int s = 0; s < 5; ++s) launch_mp(M_PLAIN, pf, sa, S[s], args_of(0, l, s, false, true));
            CK(hipStreamSynchronize(sa));
            timeline("serial", M_PLAIN, pf, 1);
        }
        // ---- the same chain as one hipGraph per pass -----------------------------------------------------------------------------
        {
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(sa, hipStreamCaptureModeThreadLocal));
            for (int l = 0; l < Ls; ++l)
                for (int s = 0; s < 5; ++s) launch_mp(M_PLAIN, 0, sa, S[s], args_of(0, l, s, false, false));
            CK(hipStreamEndCapture(sa, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            reset(); CK(hipGraphLaunch(ge, sa)); CK(hipStreamSynchronize(sa));
            reset();
            CK(hipEventRecord(e0, sa));
            for (int r = 0; r < REPS; ++r) CK(hipGraphLaunch(ge, sa));
            CK(hipEventRecord(e1, sa)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            report("graph", M_PLAIN, 0, 1, ms, false);
            CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        }
        // ---- two streams, device-side counters (CHAIN_TWO_STREAMS=1): eager, then as one hipGraph with two branches -------------------
        if (getenv("CHAIN_TWO_STREAMS") && shape < 2)
            for (int mode : {M_FENCE, M_SC1})
                for (int pf : {0, 2}) {
                    poll_sleep = 1;
                    for (int timed = 0; timed < 2; ++timed) {
                        reset(); CK(hipStreamSynchronize(sb));
                        CK(hipEventRecord(e0, sa)); CK(hipStreamWaitEvent(sb, e0, 0));
                        int k = 0;
                        for (int r = 0; r < (timed ? REPS : 1); ++r)
                            for (int l = 0; l < Ls; ++l)
                                for (int s = 0; s < 5; ++s, ++k) launch_mp(mode, pf, (k & 1) ? sb : sa, S[s], args_of(r, l, s, true, false));
                        CK(hipEventRecord(ej, sb)); CK(hipStreamWaitEvent(sa, ej, 0));
                        CK(hipEventRecord(e1, sa)); CK(hipEventSynchronize(e1));
                        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                        if (timed) report("two streams", mode, pf, 1, ms, false);
                    }
                    hipGraph_t g; hipGraphExec_t ge;      // the counters are zeroed by a memset node every other node depends on
                    CK(hipStreamBeginCapture(sa, hipStreamCaptureModeThreadLocal));
                    CK(hipMemsetAsync(flags, 0, (size_t)nk * 4, sa));
                    CK(hipEventRecord(ej, sa)); CK(hipStreamWaitEvent(sb, ej, 0));
                    int k = 0;
                    for (int l = 0; l < Ls; ++l)
                        for (int s = 0; s < 5; ++s, ++k) launch_mp(mode, pf, (k & 1) ? sb : sa, S[s], args_of(0, l, s, true, false));
                    CK(hipEventRecord(ej, sb)); CK(hipStreamWaitEvent(sa, ej, 0));
                    CK(hipStreamEndCapture(sa, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
                    reset(); CK(hipGraphLaunch(ge, sa)); CK(hipStreamSynchronize(sa));
                    unsigned e = 0; CK(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost));
                    if (e) { printf("{\"probe\":\"chain\",\"shape\":\"%s\",\"how\":\"two streams, one graph\",\"visibility\":\"%s\",\"prefetch_vectors\":%d,\"warmup_err\":\"0x%x\"}\n",
                                    sname, mode == M_FENCE ? "fence" : "sc1", pf, e); fflush(stdout); }
                    else {
                        reset();
                        CK(hipEventRecord(e0, sa));
                        for (int r = 0; r < REPS; ++r) CK(hipGraphLaunch(ge, sa));
                        CK(hipEventRecord(e1, sa)); CK(hipEventSynchronize(e1));
                        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                        report("two streams, one graph", mode, pf, 1, ms, false);
                    }
                    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
                }
        // ---- several stages per launch, one stream: the counters order the stages INSIDE a launch, the stream orders the launches ---
        struct Var { int mode, pf, per, sleep; };
        const Var vars[] = {{M_SC1, 0, 5, 1}, {M_SC1, 0, 5, 4}, {M_SC1, 0, 5, 16}, {M_SC1, 2, 5, 16}, {M_HYB, 0, 5, 4}, {M_HYB, 0, 5, 16}, {M_HYB, 2, 5, 16},
                            {M_SC1, 0, 2, 16}, {M_HYB, 0, 2, 16}, {M_HYB, 0, 20, 16}, {M_FENCE, 0, 5, 16}};
        const int nvars = shape >= 2 ? 0 : getenv("CHAIN_FUSED") ? atoi(getenv("CHAIN_FUSED")) : (int)(sizeof(vars) / sizeof(vars[0]));
        for (int vi = 0; vi < nvars; ++vi) {
            const Var& v = vars[vi];
            bool bad = false;
            poll_sleep = v.sleep;
            for (int pass = 0; pass < 3 && !bad; ++pass) {   // warm-up, timed, stamped
                reset();
                if (pass == 2) stamp_init();
                const int reps = pass == 1 ? REPS : 1;
                CK(hipEventRecord(e0, sa));
                for (int r = 0; r < reps; ++r)
                    for (int k0 = 0; k0 < nk; k0 += v.per) {
                        FusedArgs f; f.n = 0; f.first[0] = 0;
                        for (int k = k0; k < k0 + v.per && k < nk; ++k) {
                            StageArgs a = args_of(r, k / 5, k % 5, true, pass == 2);
                            if (k == k0) a.wait_idx = -1;        // the launch before this one has finished: stream order
                            f.st[f.n] = a; f.first[f.n + 1] = f.first[f.n] + (a.nwaves + 3) / 4; ++f.n;
                        }
                        launch_fused(v.mode, v.pf, sa, f, max_lds);
                    }
                CK(hipEventRecord(e1, sa)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                unsigned e = 0; CK(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost));
                if (pass == 0 && e) { printf("{\"probe\":\"chain\",\"shape\":\"%s\",\"how\":\"fused\",\"visibility\":\"%s\",\"prefetch_vectors\":%d,\"stages_per_launch\":%d,\"warmup_err\":\"0x%x\"}\n",
                                             sname, v.mode == M_FENCE ? "fence" : "sc1", v.pf, v.per, e); fflush(stdout); bad = true; }
                if (pass == 1) report("fused", v.mode, v.pf, v.per, ms, false);
                if (pass == 2) timeline("fused", v.mode, v.pf, v.per);
            }
        }
        CK(hipFree(W)); CK(hipFree(flags)); CK(hipFree(err)); CK(hipFree(stamps));
        for (int s = 0; s < 5; ++s) CK(hipFree(vec[s]));
    }
    return 0;
}
