// Persistent whole-token kernel for gfx950: the complete `transformer` pass
// (/root/reference/llama2.f90:480-640) in ONE launch, one 8-wave workgroup per CU.
//
// Why: with one kernel per GEMV the token is bounded by ~1.55 us of boundary + ramp per launch
// (5 launches per layer) -- the weight stream stops at every dependency.  Weights
// do not depend on activations, so here the stream never stops: each of the 7 STREAMING waves of a CU
// walks a static list of 8 KB row tiles (its share of qkv, wo, w1|w3, w2 of every layer, then the
// classifier) and always has TK_NB tiles requested ahead in registers (non-temporal 16-byte loads),
// across phase and layer boundaries.  The one SERVICE wave per CU never touches the weight stream
// (so its polls are not queued behind 40 KB of outstanding loads -- vmcnt retires in order): it
// gathers the phase's input vector from the exchange buffers into LDS, applies rmsnorm, and after
// the streaming waves have dotted their tiles against it, runs the epilogue (RoPE / SwiGLU /
// residual) and publishes the CU's outputs.
//
// Exchange = 8-byte {value, epoch-tag} granules written with ONE agent-scope (sc1, write-through)
// store and swept with agent-scope loads until every tag matches: no flags, no fences, placement
// independent (cdna_hip_programming.md Guideline 16, form R2).  Epochs are unique per
// (token serial, phase), so buffers are never reset.  Every spin is bounded; a timeout raises a
// sticky error word and lets the kernel drain.
//
// Work split per phase: CU c owns a contiguous row range of the phase's matrix (w1|w3, w2 and the
// classifier: R/256 rows each; QKV and wo: split over the 224 CUs that do NOT run attention, see
// TkShape); tile t of that range goes to streaming wave t % 7.  A tile is one row x up to 2048
// columns; its wave-reduced partial dot goes to LDS and the service wave folds the parts of a row.
// The x fragment a wave dots its tiles against is the same for every tile of a phase and lives in
// registers (TkX).  DESIGN.md section 3b has the measurements behind each of these choices.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace llmk {

constexpr int TK_NCU = 256;               // one workgroup per CU
constexpr int TK_WAVES = 8;               // 7 streaming + 1 service (2 waves/SIMD: 256 VGPRs each)
constexpr int TK_NB = 5;                  // register tiles a streaming wave keeps requested ahead (5 x 8 KB)
constexpr int TK_NS = TK_WAVES - 1;
constexpr int TK_THREADS = TK_WAVES * WAVE;
constexpr int TK_TCOLS = 8;               // 16-byte vector columns per tile (8 x 64 lanes x 4 floats = 2048)
constexpr unsigned TK_SPIN_LIMIT = 1u << 22;
constexpr int TK_TRACE_N = 16 * 64;       // stamps per CU: 16 per layer, first 64 layers
// Timing aids (wall-clock stamps of the service wave, "do not wait for tags") exist only in the debug build
// (make -C llm.f90_amd debug -> libllmk_debug.so, -DLLMK_TK_DEBUG): in the product library every use below folds away.
#ifdef LLMK_TK_DEBUG
constexpr bool TK_DEBUG = true;
#else
constexpr bool TK_DEBUG = false;
#endif

struct TokenArgs {
    const float* emb;        // [V][E]
    const float* rms_att;    // [L][E]
    const float* rms_ffn;    // [L][E]
    const float* rms_final;  // [E]
    const void* wqkv;        // [L][E+2KV][E]   f32 or f16 rows (TkShape::WT)
    const void* wo;          // [L][E][E]
    const void* w13;         // [L][2H][E]
    const void* w2;          // [L][E][H]
    const void* wcls;        // [V][E]
    float* kc;               // [L][S][KV]
    float* vc;
    const float* rope;       // [hs/2]
    const int* tokpos;       // {token0, pos1, serial} in device memory (graph replay), or null: the three fields below
    int tok_imm, pos_imm, serial_imm;
    unsigned* herr;          // optional sticky error word in HOST memory (direct mode: logits also point at host memory)
    unsigned long long* g_qkv;  // granules [E+2KV]
    unsigned long long* g_xb;   // [E]   attention output
    unsigned long long* g_xa;   // [E]   x after attention residual
    unsigned long long* g_hb;   // [H]
    unsigned long long* g_x;    // [E]   x after FFN residual
    float* logits;           // [V]
    unsigned* err;           // sticky error word (0 = ok)
    const float4* zeros;     // [NCU*TK_WAVES] 1 KB blocks of zeros: what empty ring slots / ragged row ends read
    unsigned long long* trace;  // debug build only: [NCU][TK_TRACE_N] wall-clock stamps of each CU's service wave
    int L, S;
    int nosync;              // debug build only: do not wait for exchange tags (wrong results; measures the pure streaming rate)
    // filled in per workgroup by the kernel: this CU's rows of the QKV and wo matrices (none on an attention CU)
    int q0, qn, o0, on;
    int c0, cn;              // this CU's rows of the classifier
};

template <int E_, int H_, int NH_, int NKV_, int V_, int WT_ = WT_F32>
struct TkShape {
    static constexpr int E = E_, H = H_, NH = NH_, NKV = NKV_, V = V_, WT = WT_;
    static constexpr int HS = E / NH, KV = NKV * HS, KVMUL = NH / NKV, QKV = E + 2 * KV;
    // ---- weight tiles.  A tile is TK_TCOLS (8) lane loads of 16 bytes = 8 "segments" of 1 KB.  f32: one row x 8
    // segments (2048 columns).  f16: a row of 2048 columns is 4 segments, so a tile is RPT = 2 consecutive rows x LPT = 4
    // segments -- the x fragment a lane needs (LPT segments x 16/BW columns) is 32 floats either way.
    static constexpr int BW = (WT == WT_F16) ? 2 : 4;                    // bytes per weight
    static constexpr int VPL = 16 / BW;                                  // weights per 16-byte lane load
    static constexpr int LPR_E = E * BW / 1024, LPR_H = H * BW / 1024;   // 1 KB segments per row, K = E / K = H
    static constexpr int RPT = (WT == WT_F16 && 2 * LPR_E <= TK_TCOLS) ? 2 : 1;   // rows per tile
    static constexpr int LPT = TK_TCOLS / RPT;                           // segments of ONE row in a tile
    // rows per CU and tiles per CU for each phase
    // The NH attention CUs own NO rows of the QKV and wo matrices (the two phases either side of attention): their q poll,
    // K/V rows and attention never queue behind their own weight prefetch, and nobody waits for them to catch up on
    // streaming after attention.  The other NCU_W CUs split those rows as evenly as whole RoPE pairs / tile rows allow.
    static constexpr int NCU_W = TK_NCU - NH;
    static constexpr int QB = (QKV / 2) / NCU_W, QX = (QKV / 2) % NCU_W;      // pairs per CU, CUs with one pair more
    static constexpr int OB = (E / RPT) / NCU_W, OX = (E / RPT) % NCU_W;      // tile-row groups per CU, CUs with one more
    static constexpr int CB = (V / RPT) / TK_NCU, CX = (V / RPT) % TK_NCU;    // classifier: the same over all CUs
    static constexpr int R_Q = 2 * (QB + (QX > 0 ? 1 : 0)), R_O = RPT * (OB + (OX > 0 ? 1 : 0));   // MAX rows per CU
    static constexpr int R_C = RPT * (CB + (CX > 0 ? 1 : 0));
    static constexpr int R_A = 2 * (H / TK_NCU), R_D = E / TK_NCU;
    static constexpr int TPR_H = (LPR_H + LPT - 1) / LPT;                // column parts of a w2 row
    static constexpr int NT_Q = R_Q / RPT, NT_O = R_O / RPT, NT_A = R_A / RPT, NT_D = (R_D / RPT) * TPR_H, NT_C = R_C / RPT;
    static constexpr int SL_Q = (NT_Q + TK_NS - 1) / TK_NS, SL_O = (NT_O + TK_NS - 1) / TK_NS,
                         SL_A = (NT_A + TK_NS - 1) / TK_NS, SL_C = (NT_C + TK_NS - 1) / TK_NS;
    // w2 rows are TPR_H parts wide: streaming wave sw only ever takes column part sw % TPR_H (so its x fragment can
    // live in registers for the whole phase); the part with the fewest waves (TK_NS / TPR_H of them) sets the slot count
    static constexpr int NW_D = TK_NS / TPR_H, SL_D = (R_D / RPT + NW_D - 1) / NW_D;
    static constexpr int SL_LAYER = SL_Q + SL_O + SL_A + SL_D;
    static constexpr int MAXP0 = R_A > R_C ? R_A : R_C, MAXP1 = R_D * TPR_H, MAXP = MAXP0 > MAXP1 ? MAXP0 : MAXP1;   // partial sums per phase
    static_assert(QKV % 2 == 0 && E % TK_NCU == 0 && H % TK_NCU == 0 && V % RPT == 0, "rows must split over CUs");
    static_assert(E * BW % 1024 == 0 && H * BW % 1024 == 0, "rows are whole 1 KB segments");
    static_assert(LPR_E <= LPT && R_Q % RPT == 0 && (R_A / 2) % RPT == 0 && R_D % RPT == 0, "a K = E row is one tile row; row ranges are whole tiles");
    static_assert(NH <= TK_NCU && TK_NCU % NH == 0, "one CU per head");
    static_assert(R_Q <= 64 && R_A / 2 <= 64 && R_O <= 64, "one service lane per output");
    static_assert(HS == 64, "in-kernel attention is written for head_size 64");
    static_assert(TPR_H <= TK_NS, "every column part of a w2 row needs a wave");
};

// LDS carve (bytes): xs (streaming input, up to H floats) | xraw (E) | partial | attention scratch
template <class SH>
struct TkLds {
    static constexpr int XS = 0;
    static constexpr int XRAW = XS + SH::H * 4;
    static constexpr int PART = XRAW + SH::E * 4;
    static constexpr int ATT_Q = PART + (((SH::MAXP + 1) * 4 + 15) / 16) * 16;   // q_h, k_cur, v_cur: 3*HS floats
    static constexpr int ATT_RED = ATT_Q + 3 * SH::HS * 4;                 // [16 waves][HS/4] float4
    static constexpr int ATT_R4 = ATT_RED + TK_WAVES * (256 / SH::HS) * SH::HS * 4;   // [waves][TPW][HS/4] float4
    static constexpr int ROPE = ATT_R4 + 64;                               // cos[HS/2] | sin[HS/2] of pos*freq
    static constexpr int ATT_P = ROPE + SH::HS * 4;                        // [waves][32] softmax weights of the wave's own timesteps
    static constexpr int ATT_S = ATT_P + TK_WAVES * 32 * 4;                // scores [S], then exp(score - max) [S]
};

__device__ __forceinline__ void tk_barrier() {
    // LDS traffic ordered by lgkmcnt; outstanding global LOADS deliberately stay in flight across it
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

__device__ __forceinline__ void tk_publish(unsigned long long* g, unsigned epoch, float v) {
    __hip_atomic_store(g, ((unsigned long long)epoch << 32) | (unsigned long long)__float_as_uint(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}

typedef unsigned tk_v4u __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t tk_rsrc(const void* p, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

// K/V cache rows through a buffer descriptor: a 32-bit byte offset per lane instead of a 64-bit address pair
__device__ __forceinline__ float4 tk_ldkv(__amdgpu_buffer_rsrc_t rs, int voff) {
    const tk_v4u r = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 0, 0);
    return make_float4(__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w));
}

// Service wave: collect N granules (N % 128 == 0) whose tag == epoch; values -> dst (LDS, N floats).
// A single wave pulls only a few GB/s through agent-scope loads when it waits per batch (the price
// list's handoff-payload row), so: 16-byte sc1 loads (two granules each), ALL of a lane's NL loads in
// flight at once, so a pass costs one round trip.
// Returns false on timeout / sticky error.
template <int NL>
__device__ __forceinline__ bool tk_gather_part(__amdgpu_buffer_rsrc_t rs, int first_pair, unsigned epoch, float* dst,
                                               unsigned* err, int lane, bool nowait, unsigned long long* dbg) {
    for (unsigned spin = 0;; ++spin) {
        const unsigned long long tp0 = (TK_DEBUG && dbg) ? wall_clock64() : 0;
        tk_v4u r[NL];
        // NO predicate on the loads: a per-load condition makes hipcc branch around each one and wait
        // vmcnt(0) per element (NL dependent round trips instead of one)
#pragma unroll
        for (int k = 0; k < NL; ++k)
            r[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, (first_pair + lane) * 16, k * WAVE * 16, 16);   // k in the scalar offset: one VGPR address
        bool ok = true;
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            ok = ok & (r[k].y == epoch) & (r[k].w == epoch);
            const int i = 2 * (first_pair + lane + k * WAVE);
            *reinterpret_cast<float2*>(dst + i) = make_float2(__uint_as_float(r[k].x), __uint_as_float(r[k].z));
        }
        if (__all(ok) || nowait) {
            if (TK_DEBUG && dbg && lane == 0) { dbg[0] = spin + 1; dbg[1] = wall_clock64() - tp0; }
            return true;
        }
        if ((spin & 63) == 63) {
            if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
            if (spin > TK_SPIN_LIMIT) {
                if (lane == 0) __hip_atomic_store(err, 0x100u + (epoch & 0xff), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
        __builtin_amdgcn_s_sleep(1);
    }
}
template <int N>
__device__ __forceinline__ bool tk_gather(const unsigned long long* g, unsigned epoch, float* dst, unsigned* err,
                                          int lane, bool nowait = false, unsigned long long* dbg = nullptr) {
    static_assert(N % 128 == 0, "two granules per 16-byte load, 64 lanes");
    constexpr int NL = N / 128;   // 16-byte loads per lane
    const __amdgpu_buffer_rsrc_t rs = tk_rsrc(g, N * 8);
    if constexpr (NL <= 24) {
        return tk_gather_part<NL>(rs, 0, epoch, dst, err, lane, nowait, dbg);
    } else {      // long vectors in two register-sized halves
        constexpr int H0 = NL / 2, H1 = NL - H0;
        const bool a = tk_gather_part<H0>(rs, 0, epoch, dst, err, lane, nowait, dbg);
        const bool b = tk_gather_part<H1>(rs, H0 * WAVE, epoch, dst, err, lane, nowait, dbg ? dbg + 2 : nullptr);
        return a && b;
    }
}

// rmsnorm on one wave, lane <-> 4 consecutive elements per 256: the gains are requested (plain
// loads, static data) BEFORE the exchange is polled, so their HBM latency hides under the gather.
//   xs = x*w/sqrt(mean(x^2)+1e-5)                                         llama2.f90:450-457
template <int E>
struct TkNorm {
    static constexpr int PER = E / (4 * WAVE);
    float4 w[PER];
    __device__ __forceinline__ void prefetch(const float* __restrict__ gains, int lane) {
#pragma unroll
        for (int k = 0; k < PER; ++k) w[k] = reinterpret_cast<const float4*>(gains)[lane + k * WAVE];
    }
    // Stages xs = x*w and returns xn = sqrt(mean(x^2)+1e-5).  The division by xn is linear in the dot
    // product, so it is applied ONCE to each finished row sum (W.(x*w))/xn by the epilogue instead
    // of 2048 times here -- the service wave is the serial section of every phase.
    __device__ __forceinline__ float apply(const float* xraw, float* xs, int lane) const {
        float ss = 0.f;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const float4 x = reinterpret_cast<const float4*>(xraw)[lane + k * WAVE];
            ss = dot4(x, x, ss);
            float4 o;
            o.x = x.x * w[k].x;
            o.y = x.y * w[k].y;
            o.z = x.z * w[k].z;
            o.w = x.w * w[k].w;
            reinterpret_cast<float4*>(xs)[lane + k * WAVE] = o;
        }
        ss = wave_sum(ss);
        return sqrtf(ss / (float)E + 1e-5f);
    }
};

// Every ring slot is exactly TK_TCOLS unconditional loads: a slot with no tile, and the columns past a
// ragged row end, read a 1 KB block of ZEROS (L2-resident) instead of being skipped.  With no control
// flow around the loads hipcc can count them, so consuming the oldest of the TK_NB tiles waits with
// vmcnt(24) and leaves the three younger tiles in flight; a skipped load would force vmcnt(0).
struct TkTile {
    const float4* p;  // first segment of the tile's first row (the zero block when the slot is empty)
    int rstride;      // float4 units between the tile's rows (RPT > 1)
    int ncol;         // real segments per tile row (0..LPT); segments >= ncol read zeros
    int pidx;         // partial index of the tile's first row (MAXP = junk slot); row s of the tile: pidx + s * pstep
    int pstep;
};

template <class SH>
__device__ __forceinline__ void tk_issue(float4 (&b)[TK_TCOLS], const TkTile& t, const float4* zp, int lane) {
#pragma unroll
    for (int j = 0; j < TK_TCOLS; ++j) {
        const int s = j / SH::LPT, jj = j % SH::LPT;                             // compile-time
        const bool real = jj < t.ncol;                                             // wave-uniform select, no branch
        const float4* pj = real ? t.p + s * t.rstride + jj * WAVE : zp;
        // an empty segment is ONE 16-byte access for the whole wave (every lane reads the same zero vector), not a 1 KB
        // sweep of the zero block: it still counts in vmcnt, but costs the CU's memory pipeline one line instead of eight
        b[j] = ldg_nt(pj + (real ? lane : 0));
    }
}
// Every tile of a phase is dotted against the same x fragment (the LPT segments of a row, or of one column part of a
// w2 row): it is read from LDS once per phase into registers, not once per tile -- 7 waves x 8 KB of ds_read per
// slot was ~0.2 us of LDS time in the middle of every slot of the critical path.  32 floats per lane for f32 and f16.
template <class SH>
struct TkX {
    static constexpr int F4 = SH::VPL / 4;          // float4 of x per segment and lane: 1 (f32) or 2 (f16)
    float4 v[SH::LPT * F4];
    // segments seg0 .. seg0+LPT-1 of a vector with nseg segments; segments past the end read as zero
    __device__ __forceinline__ void load(const float4* xs, int seg0, int nseg, int lane) {
#pragma unroll
        for (int j = 0; j < SH::LPT; ++j) {
            const bool in = seg0 + j < nseg;
#pragma unroll
            for (int h = 0; h < F4; ++h) {
                const float4 x = xs[in ? ((seg0 + j) * WAVE + lane) * F4 + h : lane];
                v[j * F4 + h] = in ? x : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }
};
__device__ __forceinline__ float4 tk_h2f_lo(const float4& w) {   // halves 0..3 of a 16-byte vector of 8
    const __half2 a = *reinterpret_cast<const __half2*>(&w.x), b = *reinterpret_cast<const __half2*>(&w.y);
    return make_float4(__low2float(a), __high2float(a), __low2float(b), __high2float(b));
}
__device__ __forceinline__ float4 tk_h2f_hi(const float4& w) {   // halves 4..7
    const __half2 a = *reinterpret_cast<const __half2*>(&w.z), b = *reinterpret_cast<const __half2*>(&w.w);
    return make_float4(__low2float(a), __high2float(a), __low2float(b), __high2float(b));
}
// per-lane partial dots of one tile (one per tile row): four independent FMA chains (x,y,z,w) instead of one long chain
template <class SH>
__device__ __forceinline__ void tk_dot(const float4 (&b)[TK_TCOLS], const TkX<SH>& x, float (&out)[SH::RPT]) {
#pragma unroll
    for (int s = 0; s < SH::RPT; ++s) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int jj = 0; jj < SH::LPT; ++jj) {
            const float4& w = b[s * SH::LPT + jj];
            if constexpr (SH::WT == WT_F16) {
                const float4 lo = tk_h2f_lo(w), hi = tk_h2f_hi(w);
                const float4 &xl = x.v[2 * jj], &xh = x.v[2 * jj + 1];
                acc.x = fmaf(lo.x, xl.x, acc.x); acc.y = fmaf(lo.y, xl.y, acc.y);
                acc.z = fmaf(lo.z, xl.z, acc.z); acc.w = fmaf(lo.w, xl.w, acc.w);
                acc.x = fmaf(hi.x, xh.x, acc.x); acc.y = fmaf(hi.y, xh.y, acc.y);
                acc.z = fmaf(hi.z, xh.z, acc.z); acc.w = fmaf(hi.w, xh.w, acc.w);
            } else {
                acc.x = fmaf(w.x, x.v[jj].x, acc.x);
                acc.y = fmaf(w.y, x.v[jj].y, acc.y);
                acc.z = fmaf(w.z, x.v[jj].z, acc.z);
                acc.w = fmaf(w.w, x.v[jj].w, acc.w);
            }
        }
        out[s] = (acc.x + acc.y) + (acc.z + acc.w);
    }
}
template <class SH>
__device__ __forceinline__ void tk_consume(const float4 (&b)[TK_TCOLS], const TkTile& t, const TkX<SH>& x, float* part, int lane) {
    float v[SH::RPT];
    tk_dot<SH>(b, x, v);
#pragma unroll
    for (int s = 0; s < SH::RPT; ++s) v[s] = wave_sum(v[s]);
    if (lane == 0) {
#pragma unroll
        for (int s = 0; s < SH::RPT; ++s) part[t.pidx + s * t.pstep] = v[s];
    }
}

// Static per-wave schedule: SLP slots per layer (padded to an even count so the 2-deep ring has the
// same parity at every layer start), then the classifier slots.  Slot K (compile-time) belongs to
// one phase; the s-th slot of a phase is tile s*15 + sw of the CU's tile range of that phase
// (null when past the end).  Everything but l, c, sw is a compile-time constant.
template <class SH>
struct TkSched {
    static constexpr int SLP = (SH::SL_LAYER + TK_NB - 1) / TK_NB * TK_NB;   // padded slots per layer
    static constexpr int KQ = 0, KO = KQ + SH::SL_Q, KA = KO + SH::SL_O, KD = KA + SH::SL_A, KP = KD + SH::SL_D;
};

template <class SH>
__device__ __forceinline__ TkTile tk_null(const float4* zp) {
    TkTile t;
    t.p = zp; t.rstride = 0; t.ncol = 0; t.pidx = SH::MAXP; t.pstep = 0;
    return t;
}
// row r of a [rows][K] matrix of SH's weight type
template <class SH, int K>
__device__ __forceinline__ const float4* tk_rowp(const void* mat, long long r) {
    return reinterpret_cast<const float4*>(static_cast<const char*>(mat) + (size_t)r * K * SH::BW);
}

// tile ti of a CU's run-time row range [row0, row0+n) of a K = E matrix: RPT consecutive (= contiguous) rows
template <class SH>
__device__ __forceinline__ TkTile tk_row_tile(const void* mat, long long row0, int ti, int n, const float4* zp) {
    TkTile t;
    const bool live = ti * SH::RPT < n;
    t.ncol = live ? SH::LPR_E : 0;
    t.p = live ? tk_rowp<SH, SH::E>(mat, row0 + ti * SH::RPT) : zp;
    t.rstride = SH::LPR_E * WAVE;
    t.pidx = live ? ti * SH::RPT : SH::MAXP;
    t.pstep = live ? 1 : 0;
    return t;
}

template <class SH, int K>
__device__ __forceinline__ TkTile tk_cls_at(const TokenArgs& a, int c, int sw) {
    if constexpr (K >= SH::SL_C) return tk_null<SH>(a.zeros);
    else if constexpr (SH::CX == 0) return tk_row_tile<SH>(a.wcls, (long long)c * SH::R_C, K * TK_NS + sw, SH::R_C, a.zeros);   // even split: compile-time count
    else return tk_row_tile<SH>(a.wcls, a.c0, K * TK_NS + sw, a.cn, a.zeros);
}

// descriptor of slot K (compile-time) of layer l; K >= SLP looks into layer l+1; past the last
// layer the stream continues with the classifier slots
template <class SH, int K>
__device__ __forceinline__ TkTile tk_at(const TokenArgs& a, int l, int c, int sw) {
    typedef TkSched<SH> SC;
    static_assert(K < 2 * SC::SLP, "lookahead of at most one layer");
    if constexpr (K >= SC::SLP) {
        return tk_at<SH, K - SC::SLP>(a, l + 1, c, sw);
    } else {
        if (l >= a.L) return tk_cls_at<SH, K>(a, c, sw);
        if constexpr (K < SC::KO) {
            return tk_row_tile<SH>(a.wqkv, (long long)l * SH::QKV + a.q0, (K - SC::KQ) * TK_NS + sw, a.qn, a.zeros);
        } else if constexpr (K < SC::KA) {
            return tk_row_tile<SH>(a.wo, (long long)l * SH::E + a.o0, (K - SC::KO) * TK_NS + sw, a.on, a.zeros);
        } else if constexpr (K < SC::KD) {
            // w1|w3: tile 2m is RPT gate rows, tile 2m+1 the RPT up rows of the same hidden units (SwiGLU pairs stay in the CU);
            // partials are laid out (gate, up) per hidden unit
            const int ti = (K - SC::KA) * TK_NS + sw, m = ti >> 1, gu = ti & 1;
            const bool live = ti < SH::NT_A;
            TkTile t;
            t.ncol = live ? SH::LPR_E : 0;
            t.p = live ? tk_rowp<SH, SH::E>(a.w13, (long long)l * 2 * SH::H + gu * SH::H + c * (SH::R_A / 2) + m * SH::RPT) : a.zeros;
            t.rstride = SH::LPR_E * WAVE;
            t.pidx = live ? 2 * m * SH::RPT + gu : SH::MAXP;
            t.pstep = live ? 2 : 0;
            return t;
        } else if constexpr (K < SC::KP) {
            // w2: RPT rows x column part `part` (LPT segments; the last part is ragged)
            constexpr int P = SH::TPR_H;
            const int part = sw % P, nw = (TK_NS - part + P - 1) / P;      // waves that share this column part
            const int rg = (K - SC::KD) * nw + sw / P;                     // group of RPT rows
            const bool live = rg < SH::R_D / SH::RPT;
            TkTile t;
            t.ncol = live ? min(SH::LPT, SH::LPR_H - part * SH::LPT) : 0;
            t.p = live ? tk_rowp<SH, SH::H>(a.w2, (long long)l * SH::E + c * SH::R_D + rg * SH::RPT) + part * SH::LPT * WAVE : a.zeros;
            t.rstride = SH::LPR_H * WAVE;
            t.pidx = live ? rg * SH::RPT * P + part : SH::MAXP;
            t.pstep = live ? P : 0;
            return t;
        } else {
            return tk_null<SH>(a.zeros);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// In-kernel attention for one head on one CU, all 16 waves (llama2.f90:572-598).  q_h, k_pos, v_pos
// arrive through the exchange (LDS); rows t < pos-1 come from the caches written by earlier launches.
// ------------------------------------------------------------------------------------------------
template <class SH>
struct TkAtt {
    static constexpr int HS = SH::HS, LPT = HS / 4, TPW = 64 / LPT, TPB = TK_WAVES * TPW, U = 8, TILE = TPB * U;
    float4 kv[U], vv[U];
    // K and V rows of the first TILE (= 256) timesteps (written by earlier launches) are requested at the START of
    // the layer's attention, long before q exists: by the time q arrives they have crossed the loaded memory system.
    // (The 64 registers are the two ring entries the QKV phase has just consumed and not yet refilled; an attention
    // CU streams no QKV / wo tiles, so nothing else of its own is queued ahead of the q poll.)
    // Only contexts longer than TILE pay exposed round trips.
    __device__ __forceinline__ void prefetch(const TokenArgs& a, int l, int h, int pos, int tid) {
        const int lane = tid & 63, wid = tid >> 6;
        const int g = h / SH::KVMUL, sub = lane % LPT, tl = lane / LPT;
        const __amdgpu_buffer_rsrc_t rk = tk_rsrc(a.kc + (size_t)l * a.S * SH::KV, a.S * SH::KV * 4);
        const __amdgpu_buffer_rsrc_t rv = tk_rsrc(a.vc + (size_t)l * a.S * SH::KV, a.S * SH::KV * 4);
        const int col = (g * HS + sub * 4) * 4;
        const int tb = wid * TPW + tl, tmax = max(pos - 2, 0);
#pragma unroll
        for (int u = 0; u < U; ++u) kv[u] = tk_ldkv(rk, min(u * TPB + tb, tmax) * (SH::KV * 4) + col);
#pragma unroll
        for (int u = 0; u < U; ++u) vv[u] = tk_ldkv(rv, min(u * TPB + tb, tmax) * (SH::KV * 4) + col);
    }
};

// sum over the 16 lanes of a DPP row (= the HS/4 lanes that share a timestep); valid in lane 15 of the row
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_mov<0xB1, 0xf, true>(0.f, v);
    v += dpp_mov<0x4E, 0xf, true>(0.f, v);
    v += dpp_mov<0x114, 0xf, true>(0.f, v);
    v += dpp_mov<0x118, 0xf, true>(0.f, v);
    return v;
}

template <class SH>
__device__ __forceinline__ void tk_attention(const TokenArgs& a, char* lds, int l, int h, int pos, int tid, TkAtt<SH>& pa,
                                             unsigned long long* dbg = nullptr) {
    constexpr int HS = SH::HS, LPT = HS / 4, TPW = 64 / LPT, TPB = TK_WAVES * TPW, U = TkAtt<SH>::U, TILE = TPB * U;
    static_assert(LPT == 16, "one timestep per DPP row");
    float4 (&kv)[U] = pa.kv;
    float4 (&vv)[U] = pa.vv;
    const int lane = tid & 63, wid = tid >> 6;
    const int g = h / SH::KVMUL;
    const float* qs = reinterpret_cast<const float*>(lds + TkLds<SH>::ATT_Q);
    const float4* kcur = reinterpret_cast<const float4*>(qs + HS);
    const float4* vcur = reinterpret_cast<const float4*>(qs + 2 * HS);
    float4* red = reinterpret_cast<float4*>(lds + TkLds<SH>::ATT_RED);   // [waves][TPW][LPT] float4
    float* att = reinterpret_cast<float*>(lds + TkLds<SH>::ATT_S);
    const int sub = lane % LPT, tl = lane / LPT;
    const float4 qv = reinterpret_cast<const float4*>(qs)[sub];
    const float scale = sqrtf((float)HS);
    const __amdgpu_buffer_rsrc_t rk = tk_rsrc(a.kc + (size_t)l * a.S * SH::KV, a.S * SH::KV * 4);
    const __amdgpu_buffer_rsrc_t rv = tk_rsrc(a.vc + (size_t)l * a.S * SH::KV, a.S * SH::KV * 4);
    constexpr int rowb = SH::KV * 4;                 // bytes per cached timestep
    const int col = (g * HS + sub * 4) * 4;
    const int tb = wid * TPW + tl;
    const int npast = pos - 1;  // rows 0..pos-2 live in the cache; row pos-1 is this token's (LDS)
    const int tmax = max(npast - 1, 0);

    float* ex = att + a.S;                                                      // exp(score - max), written per wave
    float* pw = reinterpret_cast<float*>(lds + TkLds<SH>::ATT_P) + wid * 32;    // this wave's 32 weights of a batch
    for (int base = 0; base < pos; base += TILE) {
        if (base > 0) {
#pragma unroll
            for (int u = 0; u < U; ++u) kv[u] = tk_ldkv(rk, min(base + u * TPB + tb, tmax) * rowb + col);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (base + u * TPB < pos) {   // block-uniform: short contexts skip the empty batches
                const int t = base + u * TPB + tb;
                // rows >= pos-1 read a clamped (wrong) row: t = pos-1 is redone from LDS below, later ones are never used
                const float d = row16_sum(dot4(qv, kv[u], 0.f));
                if (sub == LPT - 1 && t < npast) att[t] = d / scale;               // :582
            }
        }
    }
    {   // this token's own key never went through the cache
        const float d = row16_sum(dot4(qv, kcur[sub], 0.f));
        if (wid == 0 && lane == LPT - 1) att[npast] = d / scale;
    }
    if (TK_DEBUG && dbg) dbg[0] = wall_clock64();
    tk_barrier();
    if (TK_DEBUG && dbg) dbg[1] = wall_clock64();
    // every wave folds max and sum over ALL scores itself: no cross-wave reduction, no extra barriers.  exp() is
    // evaluated once per score here (each wave keeps its own copy of what it wrote: same values, benign overlap)
    float m = -INFINITY;
    for (int t = lane; t < pos; t += WAVE) m = fmaxf(m, att[t]);
    m = wave_max(m);
    if (TK_DEBUG && dbg) dbg[3] = wall_clock64();
    float s = 0.f;
    for (int t = lane; t < pos; t += WAVE) {
        const float e = expf(att[t] - m);
        ex[t] = e;
        s += e;
    }
    s = wave_sum(s);
    if (TK_DEBUG && dbg) dbg[4] = wall_clock64();

    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int base = 0; base < pos; base += TILE) {
        if (base > 0) {
#pragma unroll
            for (int u = 0; u < U; ++u) vv[u] = tk_ldkv(rv, min(base + u * TPB + tb, tmax) * rowb + col);
        }
        // xi/sum(xi) (:476) once per timestep: lane i < 32 owns timestep (u = i / TPW, tl = i % TPW) of this wave
        if (lane < 32) {
            const int t = base + (lane / TPW) * TPB + wid * TPW + (lane % TPW);
            pw[lane] = (t < pos) ? ex[t] / s : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (base + u * TPB < pos) {
                const int t = base + u * TPB + tb;
                const float4 v4 = (t == npast) ? vcur[sub] : vv[u];
                const float p = pw[u * TPW + tl];
                acc.x = fmaf(p, v4.x, acc.x);
                acc.y = fmaf(p, v4.y, acc.y);
                acc.z = fmaf(p, v4.z, acc.z);
                acc.w = fmaf(p, v4.w, acc.w);
            }
        }
    }
    red[(wid * TPW + tl) * LPT + sub] = acc;   // the service wave folds the waves*TPW partials per dim
    if (TK_DEBUG && dbg) dbg[2] = wall_clock64();
}

// ------------------------------------------------------------------------------------------------
// SERVICE wave of a CU: everything that depends on other CUs.  Per phase: gather the input vector
// (granule sweep), rmsnorm, barrier A, barrier B, epilogue + publish.
// ------------------------------------------------------------------------------------------------
template <class SH>
__device__ __forceinline__ void tk_service(const TokenArgs& a, char* lds, int c, int lane, int tid) {
    typedef TkLds<SH> LD;
    float* xs = reinterpret_cast<float*>(lds + LD::XS);
    float* xraw = reinterpret_cast<float*>(lds + LD::XRAW);
    const float* part = reinterpret_cast<const float*>(lds + LD::PART);
    const int L = a.L;
    const int tok = a.tokpos ? a.tokpos[0] : a.tok_imm, pos = a.tokpos ? a.tokpos[1] : a.pos_imm;
    const unsigned ebase = (unsigned)(a.tokpos ? a.tokpos[2] : a.serial_imm) * (unsigned)(5 * L + 2);
    constexpr int HPC = TK_NCU / SH::NH;
    // head h runs on CU h*HPC + (its kv group mod HPC): with the dispatcher placing block b on XCD b % 8 the heads
    // of one kv group share an XCD (one L2 copy of their K/V rows) and different groups use different XCDs
    const bool att_cu = (c % HPC) == ((c / HPC / SH::KVMUL) % HPC);
    const int my_head = c / HPC;
    bool ok = true;
    // RoPE angles depend on pos only: cos/sin once per token, not once per layer        :544-548
    float* rope_cs = reinterpret_cast<float*>(lds + LD::ROPE);
    if (lane < SH::HS / 2) {
        const float rval = (float)pos * a.rope[lane];
        rope_cs[lane] = cosf(rval);
        rope_cs[SH::HS / 2 + lane] = sinf(rval);
    }
    unsigned long long* tr = (TK_DEBUG && a.trace) ? a.trace + (size_t)c * TK_TRACE_N : nullptr;
    const bool nosync = TK_DEBUG && a.nosync != 0;
#define TK_STAMP(i) do { if (tr && lane == 0 && l < 64) tr[l * 16 + (i)] = wall_clock64(); } while (0)

    for (int l = 0; l < L; ++l) {
        const unsigned e_q = ebase + 5u * l + 1, e_att = e_q + 1, e_o = e_q + 2, e_a = e_q + 3, e_d = e_q + 4;
        TK_STAMP(0);

        // ---- P0: rmsnorm + QKV + RoPE                                            llama2.f90:527-565
        TkNorm<SH::E> nrm;
        if (!att_cu) nrm.prefetch(a.rms_att + (size_t)l * SH::E, lane);
        float xn_att = 1.f;
        if (!att_cu) {   // an attention CU owns no QKV rows: it goes straight to the q poll
            if (l == 0) {
#pragma unroll 8
                for (int k = 0; k < SH::E / WAVE; ++k) xraw[lane + k * WAVE] = a.emb[(size_t)tok * SH::E + lane + k * WAVE];  // :520
            } else {
                ok = tk_gather<SH::E>(a.g_x, e_q - 1, xraw, a.err, lane, nosync, (tr && l < 32) ? tr + (32 + l) * 16 + 0 : nullptr) && ok;
            }
            TK_STAMP(1);
            xn_att = nrm.apply(xraw, xs, lane);
        }
        tk_barrier();
        TK_STAMP(2);
        tk_barrier();
        TK_STAMP(3);
        if (lane < a.qn) {
            float v0 = part[lane], v1 = part[lane ^ 1];
            v0 = v0 / xn_att;
            v1 = v1 / xn_att;
            const int r = a.q0 + lane;
            float outv = v0;
            if (r < SH::E + SH::KV) {
                // pairs (i,i+1); 1-based odd i -> head_dim = mod(i,hs) = 2j+1 (table index j)   :543-559
                const int i0 = ((r < SH::E) ? r : r - SH::E) & ~1;
                const int jf = (i0 % SH::HS) >> 1;
                const float fcr = rope_cs[jf], fci = rope_cs[SH::HS / 2 + jf];
                // even lane: own = q0, partner = q1; odd lane: own = q1, partner = q0
                outv = (lane & 1) ? (v1 * fci + v0 * fcr) : (v0 * fcr - v1 * fci);
                if (r >= SH::E) a.kc[((size_t)l * a.S + (pos - 1)) * SH::KV + (r - SH::E)] = outv;          // :564
            } else {
                a.vc[((size_t)l * a.S + (pos - 1)) * SH::KV + (r - SH::E - SH::KV)] = outv;                 // :565
            }
            tk_publish(a.g_qkv + r, e_q, outv);
        }
        TK_STAMP(4);
        // ---- P1: attention, one CU per head                                     llama2.f90:572-598
        if (att_cu) {
            float* qs = reinterpret_cast<float*>(lds + LD::ATT_Q);
            const int g = my_head / SH::KVMUL;
            TkAtt<SH> pa;
            pa.prefetch(a, l, my_head, pos, tid);   // K/V rows cross the memory system while q is awaited
            for (unsigned spin = 0;; ++spin) {
                const unsigned long long xq = __hip_atomic_load(a.g_qkv + my_head * SH::HS + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long xk = __hip_atomic_load(a.g_qkv + SH::E + g * SH::HS + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long xv = __hip_atomic_load(a.g_qkv + SH::E + SH::KV + g * SH::HS + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const bool good = (unsigned)(xq >> 32) == e_q && (unsigned)(xk >> 32) == e_q && (unsigned)(xv >> 32) == e_q;
                qs[lane] = __uint_as_float((unsigned)xq);
                qs[SH::HS + lane] = __uint_as_float((unsigned)xk);
                qs[2 * SH::HS + lane] = __uint_as_float((unsigned)xv);
                if (__all(good) || nosync) break;
                if ((spin & 63) == 63) {
                    if (__hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { ok = false; break; }
                    if (spin > TK_SPIN_LIMIT) {
                        if (lane == 0) __hip_atomic_store(a.err, 0x200u + (unsigned)l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ok = false; break;
                    }
                }
                __builtin_amdgcn_s_sleep(1);
            }
            TK_STAMP(5);
            tk_barrier();
            tk_attention<SH>(a, lds, l, my_head, pos, tid, pa, (tr && lane == 0 && l < 22) ? tr + (32 + l) * 16 + 10 : nullptr);
            tk_barrier();
            TK_STAMP(6);
            // fold the waves*TPW partial output vectors: lane = output dim, one conflict-free ds_read_b32 per partial
            const float* redf = reinterpret_cast<const float*>(lds + LD::ATT_RED);
            float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
#pragma unroll
            for (int w = 0; w < TK_WAVES * (256 / SH::HS); w += 4) {
                o0 += redf[(w + 0) * SH::HS + lane];
                o1 += redf[(w + 1) * SH::HS + lane];
                o2 += redf[(w + 2) * SH::HS + lane];
                o3 += redf[(w + 3) * SH::HS + lane];
            }
            const float o = (o0 + o1) + (o2 + o3);
            tk_publish(a.g_xb + my_head * SH::HS + lane, e_att, o);
        }
        // ---- P2: x += wo . xb                                                    llama2.f90:603-605
        if (!att_cu) ok = tk_gather<SH::E>(a.g_xb, e_att, xs, a.err, lane, nosync, (tr && l < 32) ? tr + (32 + l) * 16 + 2 : nullptr) && ok;
        TK_STAMP(7);
        tk_barrier();
        tk_barrier();
        TK_STAMP(8);
        if (lane < a.on) {
            const float v = part[lane];
            const int r = a.o0 + lane;
            tk_publish(a.g_xa + r, e_o, xraw[r] + v);
        }
        // ---- P3: rmsnorm + w1|w3 + SwiGLU                                        llama2.f90:608-616
        nrm.prefetch(a.rms_ffn + (size_t)l * SH::E, lane);
        ok = tk_gather<SH::E>(a.g_xa, e_o, xraw, a.err, lane, nosync, (tr && l < 32) ? tr + (32 + l) * 16 + 4 : nullptr) && ok;
        TK_STAMP(9);
        const float xn_ffn = nrm.apply(xraw, xs, lane);
        tk_barrier();
        TK_STAMP(10);
        tk_barrier();
        TK_STAMP(11);
        if (lane < SH::R_A / 2) {
            float gsum = part[2 * lane], usum = part[2 * lane + 1];
            gsum = gsum / xn_ffn;
            usum = usum / xn_ffn;
            const float hb = gsum * (1.0f / (1.0f + expf(-gsum)));
            tk_publish(a.g_hb + c * (SH::R_A / 2) + lane, e_a, hb * usum);
        }
        // ---- P4: x += w2 . hb                                                    llama2.f90:618-620
        ok = tk_gather<SH::H>(a.g_hb, e_a, xs, a.err, lane, nosync, (tr && l < 32) ? tr + (32 + l) * 16 + 6 : nullptr) && ok;
        TK_STAMP(12);
        tk_barrier();
        TK_STAMP(13);
        tk_barrier();
        TK_STAMP(14);
        if (lane < SH::R_D) {
            float v = 0.f;
#pragma unroll
            for (int p = 0; p < SH::TPR_H; ++p) v += part[lane * SH::TPR_H + p];
            const int r = c * SH::R_D + lane;
            tk_publish(a.g_x + r, e_d, xraw[r] + v);
        }
        TK_STAMP(15);
    }
#undef TK_STAMP
    // ---- final rmsnorm + classifier                                             llama2.f90:627-636
    TkNorm<SH::E> nrmf;
    nrmf.prefetch(a.rms_final, lane);
    ok = tk_gather<SH::E>(a.g_x, ebase + 5u * L, xraw, a.err, lane, nosync) && ok;
    const float xn_fin = nrmf.apply(xraw, xs, lane);
    tk_barrier();
    tk_barrier();
    const int cn = SH::CX ? a.cn : SH::R_C, c0 = SH::CX ? a.c0 : c * SH::R_C;
    for (int j = lane; j < cn; j += WAVE) a.logits[c0 + j] = part[j] / xn_fin;
    if (!ok && lane == 0) {
        atomicOr(a.err, 0x1000u);
        if (a.herr) __hip_atomic_store(a.herr, 0x1000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ------------------------------------------------------------------------------------------------
// STREAMING wave: a static list of 8 KB row tiles, always TK_NB requested ahead (register ring),
// consumed between the phase's two barriers.  Nothing here depends on another CU.
// ------------------------------------------------------------------------------------------------
struct TkRing {
    float4 b[TK_NB][TK_TCOLS];
    TkTile t[TK_NB];
};

// slots K .. K+N-1 (compile-time) of layer l: consume ring entry K % NB, refill it with slot K + NB
template <class SH, int K, int N, bool CLS>
__device__ __forceinline__ void tk_run(TkRing& r, const TokenArgs& a, int l, int c, int sw, const TkX<SH>& x, float* part, int lane) {
    if constexpr (N > 0) {
        constexpr int R = K % TK_NB;
        tk_consume<SH>(r.b[R], r.t[R], x, part, lane);
        if constexpr (CLS) r.t[R] = tk_cls_at<SH, K + TK_NB>(a, c, sw);
        else r.t[R] = tk_at<SH, K + TK_NB>(a, l, c, sw);
        tk_issue<SH>(r.b[R], r.t[R], a.zeros, lane);
        tk_run<SH, K + 1, N - 1, CLS>(r, a, l, c, sw, x, part, lane);
    }
}
// consume only / refill only: a phase's LAST min(NB, slots) slots are dotted first, the partial sums
// handed to the service wave (barrier B), and only THEN refilled.  Issuing a refill can block for
// microseconds when the CU's memory pipeline is full of earlier prefetches; that wait must not sit
// between the dot products and the publish of the phase's result.
// N tiles at once: all per-lane dots first, then the N wave reductions (independent DPP chains the
// scheduler can interleave), then ONE lane-0 block of LDS writes
template <class SH, int K, int N>
__device__ __forceinline__ void tk_eat(const TkRing& r, const TkX<SH>& x, float* part, int lane) {
    if constexpr (N > 0) {
        float v[N][SH::RPT];
#pragma unroll
        for (int i = 0; i < N; ++i) tk_dot<SH>(r.b[(K + i) % TK_NB], x, v[i]);
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int s = 0; s < SH::RPT; ++s) v[i][s] = wave_sum(v[i][s]);
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < N; ++i)
#pragma unroll
                for (int s = 0; s < SH::RPT; ++s) part[r.t[(K + i) % TK_NB].pidx + s * r.t[(K + i) % TK_NB].pstep] = v[i][s];
        }
    }
}
template <class SH, int K, int N, bool CLS>
__device__ __forceinline__ void tk_refill(TkRing& r, const TokenArgs& a, int l, int c, int sw, int lane) {
    if constexpr (N > 0) {
        constexpr int R = K % TK_NB;
        if constexpr (CLS) r.t[R] = tk_cls_at<SH, K + TK_NB>(a, c, sw);
        else r.t[R] = tk_at<SH, K + TK_NB>(a, l, c, sw);
        tk_issue<SH>(r.b[R], r.t[R], a.zeros, lane);
        tk_refill<SH, K + 1, N - 1, CLS>(r, a, l, c, sw, lane);
    }
}
// one phase of S slots starting at slot K0: [barrier A] early slots (consume+refill), late slots
// (consume), [barrier B], late refills
template <class SH, int K0, int S, bool CLS, bool WIDE = false>
__device__ __forceinline__ void tk_phase(TkRing& r, const TokenArgs& a, int l, int c, int sw, const float4* xs4,
                                         float* part, int lane) {
    constexpr int LATE = S < TK_NB ? S : TK_NB, EARLY = S - LATE;
    tk_barrier();
    TkX<SH> x;
    if constexpr (WIDE) x.load(xs4, (sw % SH::TPR_H) * SH::LPT, SH::LPR_H, lane);
    else x.load(xs4, 0, SH::LPR_E, lane);
    tk_run<SH, K0, EARLY, CLS>(r, a, l, c, sw, x, part, lane);
    tk_eat<SH, K0 + EARLY, LATE>(r, x, part, lane);
    tk_barrier();
    tk_refill<SH, K0 + EARLY, LATE, CLS>(r, a, l, c, sw, lane);
}

template <class SH, int K>
__device__ __forceinline__ void tk_prime(TkRing& r, const TokenArgs& a, int c, int sw, int lane) {
    if constexpr (K < TK_NB) {
        r.t[K] = tk_at<SH, K>(a, 0, c, sw);
        tk_issue<SH>(r.b[K], r.t[K], a.zeros, lane);
        tk_prime<SH, K + 1>(r, a, c, sw, lane);
    }
}

template <class SH>
__device__ __forceinline__ void tk_stream(const TokenArgs& a, char* lds, int c, int sw, int lane, int tid) {
    typedef TkLds<SH> LD;
    typedef TkSched<SH> SC;
    const float4* xs4 = reinterpret_cast<const float4*>(lds + LD::XS);
    float* part = reinterpret_cast<float*>(lds + LD::PART);
    const int L = a.L;
    const int pos = a.tokpos ? a.tokpos[1] : a.pos_imm;
    constexpr int HPC = TK_NCU / SH::NH;
    // head h runs on CU h*HPC + (its kv group mod HPC): with the dispatcher placing block b on XCD b % 8 the heads
    // of one kv group share an XCD (one L2 copy of their K/V rows) and different groups use different XCDs
    const bool att_cu = (c % HPC) == ((c / HPC / SH::KVMUL) % HPC);
    const int my_head = c / HPC;

    TkRing r;
    tk_prime<SH, 0>(r, a, c, sw, lane);

    for (int l = 0; l < L; ++l) {
        {   // QKV phase; on the attention CUs the refills wait until attention has issued ITS loads,
            // which would otherwise queue behind 100+ KB of prefetch in this CU's memory pipeline
            constexpr int LATE = SH::SL_Q < TK_NB ? SH::SL_Q : TK_NB, EARLY = SH::SL_Q - LATE;
            tk_barrier();
            TkX<SH> x;
            x.load(xs4, 0, SH::LPR_E, lane);
            tk_run<SH, SC::KQ, EARLY, false>(r, a, l, c, sw, x, part, lane);
            tk_eat<SH, SC::KQ + EARLY, LATE>(r, x, part, lane);
            tk_barrier();
            if (att_cu) {
                TkAtt<SH> pa;
                pa.prefetch(a, l, my_head, pos, tid);   // before the wait for q: K/V latency overlaps it
                tk_barrier();
                tk_attention<SH>(a, lds, l, my_head, pos, tid, pa);
                tk_barrier();
            }
            tk_refill<SH, SC::KQ + EARLY, LATE, false>(r, a, l, c, sw, lane);
        }
        tk_phase<SH, SC::KO, SH::SL_O, false>(r, a, l, c, sw, xs4, part, lane);
        tk_phase<SH, SC::KA, SH::SL_A, false>(r, a, l, c, sw, xs4, part, lane);
        tk_phase<SH, SC::KD, SC::SLP - SC::KD, false, true>(r, a, l, c, sw, xs4, part, lane);   // w2 slots + padding
    }
    // classifier: the ring index is 0 again (SLP is a multiple of TK_NB); refills run off the stream's end
    tk_phase<SH, 0, SH::SL_C, true>(r, a, L, c, sw, xs4, part, lane);
}

template <class SH>
__global__ __launch_bounds__(TK_THREADS, 2) void token_kernel(TokenArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = blockIdx.x;
    // every wave reads its OWN zero block: one shared block would make 1800 waves hammer one HBM channel
    a.zeros += (size_t)(c * TK_WAVES + wid) * WAVE;
    {
        constexpr int HPC = TK_NCU / SH::NH;
        const int blk = c / HPC, ap = (blk / SH::KVMUL) % HPC;       // attention CU of this block of HPC CUs (see tk_service)
        const bool att_cu = (c % HPC) == ap;
        const int n = c - blk - ((c % HPC) > ap ? 1 : 0);            // rank among the CUs that own QKV / wo rows
        a.qn = att_cu ? 0 : 2 * (SH::QB + (n < SH::QX ? 1 : 0));
        a.q0 = 2 * (n * SH::QB + min(n, SH::QX));
        a.on = att_cu ? 0 : SH::RPT * (SH::OB + (n < SH::OX ? 1 : 0));
        a.o0 = SH::RPT * (n * SH::OB + min(n, SH::OX));
        a.cn = SH::RPT * (SH::CB + (c < SH::CX ? 1 : 0));
        a.c0 = SH::RPT * (c * SH::CB + min(c, SH::CX));
    }
    if (wid == TK_NS) { __builtin_amdgcn_s_setprio(3); tk_service<SH>(a, lds, c, lane, tid); }
    else tk_stream<SH>(a, lds, c, wid, lane, tid);
}

typedef TkShape<2048, 5632, 32, 4, 32000> TkTinyLlama;   // /root/reference/llama2.f90:102-108
typedef TkShape<256, 768, 4, 2, 1024> TkSmall;           // tests/golden/tk-small*.npz: pinned to the real reference
typedef TkShape<2048, 5632, 32, 4, 32000, WT_F16> TkTinyLlamaF16;   // BASELINE.json configs[2]: the same model, f16 matrices
typedef TkShape<512, 1536, 8, 2, 1024, WT_F16> TkSmallF16;          // parity shape for the f16 tiles (tests: tk-small16)

}  // namespace llmk
