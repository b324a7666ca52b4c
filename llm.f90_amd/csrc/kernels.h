// gfx950 (MI355X, CDNA4) device kernels for the llm.f90 single-token forward pass.
// Each kernel cites the reference lines it replaces (/root/reference/llama2.f90).
//
// Shape of the problem: decode is one GEMV after another over weights that are read exactly once
// per token, 0.5 flop/B -> HBM-bound.  So: every wave streams whole weight rows with 16-byte
// non-temporal loads straight into VGPRs (no LDS round trip for a stream nobody shares), the
// 8..22 KB activation vector is staged once per block in LDS, partial sums are reduced with
// wave64 cross-lane ops, and everything elementwise (rmsnorm, RoPE, KV write, SwiGLU, residual)
// is fused into the prologue/epilogue of the GEMV that produces or consumes it.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

namespace llmk {

constexpr int WAVE = 64;
constexpr int GEMV_THREADS = 256;  // 4 waves
constexpr int GEMV_WAVES = GEMV_THREADS / WAVE;
constexpr int GEMV_SU = 8;         // vectors of x per thread requested at once while a block stages it (K <= 8192: all of them)

enum WeightType { WT_F32 = 0, WT_F16 = 1, WT_Q4_0 = 2, WT_Q6_K = 14 };   // (q6_K: classifier rows only, q6k.h)
enum Epilogue {
    EPI_STORE = 0,    // y[r] = W[r]·x                           classifier, llama2.f90:634-636
    EPI_RESID = 1,    // y[r] += W[r]·x                          wo :603-605, w2 :618-620
    EPI_ROPE_KV = 2,  // qkv rows -> RoPE -> q / key_cache / value_cache   :529-565
    EPI_SWIGLU = 3,   // hb[r] = silu(W[r]·x) * (W[r+H]·x)       :610-616
};

struct GemvArgs {
    const void* W;        // this layer's matrix, row-major [rows][K] in the weight type
    const float* x;       // input vector [K] (global)
    const float* norm_w;  // rmsnorm gain [K] when NORM
    float* y;             // STORE/RESID: output [rows]; ROPE_KV: q [E]; SWIGLU: hb [H]
    int rows;             // number of matrix rows
    int K;                // row length (in_features)
    // ROPE_KV
    const float* rope_freqs;  // [hs/2]
    const int* tokpos;        // device {token0, pos1}
    float* kc;                // key_cache   base of this layer [S][KV]
    float* vc;                // value_cache base of this layer [S][KV]
    int E, KV, hs;
    // SWIGLU
    int H;
    float eps;            // rmsnorm epsilon (1e-5 unless the host opted into the file's, llmk_set_rms_eps)
    // q4_0 only (device layout of llmk_upload): a row is its K/2 nibble bytes followed by its K/32 f16 block scales,
    // rows are row_stride bytes apart (16-byte aligned)
    int row_stride;
};

// Wave64 reductions on the DPP crossbar (quad_perm / row_shr / row_bcast), result broadcast through
// v_readlane: ~10 VALU ops.  The __shfl_xor ladder compiles to six dependent ds_bpermute_b32 (an
// LDS-crossbar round trip each, ~700 cycles per reduction), which showed up as the per-tile cost of
// the streaming waves.
template <int CTRL, int ROW_MASK, bool BOUND>
__device__ __forceinline__ float dpp_mov(float old, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v),
                                                                  CTRL, ROW_MASK, 0xf, BOUND));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_mov<0xB1, 0xf, true>(0.f, v);    // quad_perm:[1,0,3,2]
    v += dpp_mov<0x4E, 0xf, true>(0.f, v);    // quad_perm:[2,3,0,1]
    v += dpp_mov<0x114, 0xf, true>(0.f, v);   // row_shr:4
    v += dpp_mov<0x118, 0xf, true>(0.f, v);   // row_shr:8   -> lanes 12..15 of each row hold the row sum
    v += dpp_mov<0x142, 0xa, true>(0.f, v);   // row_bcast:15 into rows 1,3
    v += dpp_mov<0x143, 0xc, true>(0.f, v);   // row_bcast:31 into rows 2,3 -> lane 63 holds the total
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
    constexpr float NI = -__builtin_huge_valf();
    v = fmaxf(v, dpp_mov<0xB1, 0xf, false>(NI, v));
    v = fmaxf(v, dpp_mov<0x4E, 0xf, false>(NI, v));
    v = fmaxf(v, dpp_mov<0x114, 0xf, false>(NI, v));
    v = fmaxf(v, dpp_mov<0x118, 0xf, false>(NI, v));
    v = fmaxf(v, dpp_mov<0x142, 0xa, false>(NI, v));
    v = fmaxf(v, dpp_mov<0x143, 0xc, false>(NI, v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// 16-byte non-temporal global load (weights are read once per token: keep them out of L2's way).
__device__ __forceinline__ float4 ldg_nt(const float4* p) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint4 ldg_nt(const uint4* p) {
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    v4u v = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}

__device__ __forceinline__ float dot4(const float4& a, const float4& b, float acc) {
    acc = fmaf(a.x, b.x, acc);
    acc = fmaf(a.y, b.y, acc);
    acc = fmaf(a.z, b.z, acc);
    acc = fmaf(a.w, b.w, acc);
    return acc;
}
__device__ __forceinline__ float2 h2f(unsigned u) {
    __half2 h = *reinterpret_cast<__half2*>(&u);
    return __half22float2(h);
}
// 8 f16 weights (one uint4) against 8 f32 activations (two float4)
__device__ __forceinline__ float dot8h(const uint4& w, const float4& x0, const float4& x1, float acc) {
    float2 a = h2f(w.x), b = h2f(w.y), c = h2f(w.z), d = h2f(w.w);
    acc = fmaf(a.x, x0.x, acc);
    acc = fmaf(a.y, x0.y, acc);
    acc = fmaf(b.x, x0.z, acc);
    acc = fmaf(b.y, x0.w, acc);
    acc = fmaf(c.x, x1.x, acc);
    acc = fmaf(c.y, x1.y, acc);
    acc = fmaf(d.x, x1.z, acc);
    acc = fmaf(d.y, x1.w, acc);
    return acc;
}

// ------------------------------------------------------------------------------------------------
// Weight-type traits: how one lane pulls "one vector" (16 B) of a row and dots it against x in LDS.
//   VE = weights per 16-byte vector.  Row r, vector v  <->  weights [v*VE, (v+1)*VE).
// x in LDS: f32/f16 natural order (float4 xs[K/4]).
// q4_0 (device layout, re-packed by llmk_upload): per row K/2 nibble bytes, then the row's K/32 f16 scales
// (row stride = 9K/16 rounded up to 16 bytes); one vector = the 16 nibble bytes of one 32-weight block, low nibbles = elements
// 0..15, high nibbles = elements 16..31 (ggml block_q4_0).  x is staged TRANSPOSED for it:
// xs[m*nblk + b] = x[32b+4m .. 32b+4m+3], so the 8 float4 reads a lane needs are lane-contiguous.
// ------------------------------------------------------------------------------------------------
template <int WT> struct WTraits;
template <> struct WTraits<WT_F32> {
    static constexpr int VE = 4;
    typedef float4 vec;
    static __device__ __forceinline__ size_t row_bytes(int K) { return (size_t)K * 4; }
};
template <> struct WTraits<WT_F16> {
    static constexpr int VE = 8;
    typedef uint4 vec;
    static __device__ __forceinline__ size_t row_bytes(int K) { return (size_t)K * 2; }
};
template <> struct WTraits<WT_Q4_0> {
    static constexpr int VE = 32;
    typedef uint4 vec;
    static __device__ __forceinline__ size_t row_bytes(int K) { return (size_t)K / 2; }
};

template <int WT>
__device__ __forceinline__ float vdot(const typename WTraits<WT>::vec& w, const float4* xs, int v, int nvec,
                                      float acc);
template <>
__device__ __forceinline__ float vdot<WT_F32>(const float4& w, const float4* xs, int v, int, float acc) {
    return dot4(w, xs[v], acc);
}
template <>
__device__ __forceinline__ float vdot<WT_F16>(const uint4& w, const float4* xs, int v, int, float acc) {
    return dot8h(w, xs[2 * v], xs[2 * v + 1], acc);
}
// q4_0: returns the UNSCALED integer-weighted sum for this block; caller multiplies by d.
__device__ __forceinline__ float q4_block_dot(const uint4& w, const float4* xs, int b, int nblk) {
    const unsigned u[4] = {w.x, w.y, w.z, w.w};
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {  // dword i holds bytes 4i..4i+3: lo nibbles = elems 4i..4i+3, hi = 16+4i..
        const float4 xl = xs[i * nblk + b];
        const float4 xh = xs[(4 + i) * nblk + b];
        const unsigned q = u[i];
        acc = fmaf((float)((int)(q & 0xF) - 8), xl.x, acc);
        acc = fmaf((float)((int)((q >> 8) & 0xF) - 8), xl.y, acc);
        acc = fmaf((float)((int)((q >> 16) & 0xF) - 8), xl.z, acc);
        acc = fmaf((float)((int)((q >> 24) & 0xF) - 8), xl.w, acc);
        acc = fmaf((float)((int)((q >> 4) & 0xF) - 8), xh.x, acc);
        acc = fmaf((float)((int)((q >> 12) & 0xF) - 8), xh.y, acc);
        acc = fmaf((float)((int)((q >> 20) & 0xF) - 8), xh.z, acc);
        acc = fmaf((float)((int)((q >> 28) & 0xF) - 8), xh.w, acc);
    }
    return acc;
}

// A register tile: N vector-columns of ROWS rows per lane; all ROWS*N loads are issued before any use.
template <int WT, int ROWS, int N>
struct Tile {
    typename WTraits<WT>::vec r[ROWS][N];
    __half d[ROWS][N];
    __device__ __forceinline__ void load(const typename WTraits<WT>::vec* const (&w)[ROWS],
                                         const __half* const (&sc)[ROWS], int v0) {
#pragma unroll
        for (int j = 0; j < N; ++j)
#pragma unroll
            for (int i = 0; i < ROWS; ++i) {
                r[i][j] = ldg_nt(w[i] + v0 + j * WAVE);
                if constexpr (WT == WT_Q4_0) d[i][j] = sc[i][v0 + j * WAVE];
            }
    }
    __device__ __forceinline__ void fma(const float4* xs, int v0, int nvec, float (&acc)[ROWS]) const {
#pragma unroll
        for (int j = 0; j < N; ++j)
#pragma unroll
            for (int i = 0; i < ROWS; ++i) {
                if constexpr (WT == WT_Q4_0)
                    acc[i] = fmaf(__half2float(d[i][j]), q4_block_dot(r[i][j], xs, v0 + j * WAVE, nvec), acc[i]);
                else
                    acc[i] = vdot<WT>(r[i][j], xs, v0 + j * WAVE, nvec, acc[i]);
            }
    }
};
template <int WT, int ROWS, int N>
__device__ __forceinline__ void tile_step(const typename WTraits<WT>::vec* const (&w)[ROWS],
                                          const __half* const (&sc)[ROWS], const float4* xs, int v0, int nvec,
                                          float (&acc)[ROWS]) {
    Tile<WT, ROWS, N> t;
    t.load(w, sc, v0);
    t.fma(xs, v0, nvec, acc);
}

template <int EPI>
__device__ __forceinline__ void gemv_epilogue2(const GemvArgs& a, int g, int r0, int r1, float a0, float a1) {
    if (EPI == EPI_STORE) {
        a.y[r0] = a0;
        a.y[r1] = a1;
    } else if (EPI == EPI_RESID) {
        a.y[r0] += a0;
        a.y[r1] += a1;
    } else if (EPI == EPI_SWIGLU) {
        // hb = hb*(1/(1+exp(-hb))); hb = hb*hb2        llama2.f90:615-616
        float hb = a0 * (1.0f / (1.0f + expf(-a0)));
        a.y[g] = hb * a1;
    } else {  // EPI_ROPE_KV
        const int pos = a.tokpos[1];  // 1-based, as llama2.f90:546 uses it
        const int E = a.E, KV = a.KV;
        if (r0 < E + KV) {
            const int i0 = (r0 < E) ? r0 : r0 - E;  // 0-based even index inside q or k
            // reference: 1-based odd i, head_dim = mod(i,hs) = 2j+1, freq = 1/10000**(head_dim/hs)
            const float freq = a.rope_freqs[(i0 % a.hs) >> 1];
            const float rval = (float)pos * freq;
            const float fcr = cosf(rval), fci = sinf(rval);
            const float o0 = a0 * fcr - a1 * fci;
            const float o1 = a0 * fci + a1 * fcr;
            float* dst = (r0 < E) ? a.y + i0 : a.kc + (size_t)(pos - 1) * KV + i0;
            dst[0] = o0;
            dst[1] = o1;
        } else {
            float* dst = a.vc + (size_t)(pos - 1) * KV + (r0 - E - KV);
            dst[0] = a0;
            dst[1] = a1;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The GEMV.  One block = 4 waves, ONE JOB PER WAVE (no loop over jobs): a job is
//   ROWS=2, STORE/RESID/ROPE_KV: rows (2g, 2g+1)  -- a RoPE pair is (i, i+1), llama2.f90:549-552
//   ROWS=2, SWIGLU:              rows (g, g+H)    -- gate row and its up row, llama2.f90:613-616
//   ROWS=1, STORE/RESID:         row g            -- long rows (w2, K = hidden_dim)
// so that ALL of a kernel's weight bytes are requested from HBM the moment its waves start: a
// 20-90 MB GEMV is a latency problem as much as a bandwidth one (one kernel = 3-15 us of
// streaming), and a wave that loops pays the ~2 us loaded HBM latency once per round.
// The first NCH vector-columns per row (the whole row for the hot shapes) are requested BEFORE
// the block stages x: weights do not depend on activations, so HBM latency overlaps the
// x load / rmsnorm / LDS write instead of following it.
// NORM fuses rmsnorm (llama2.f90:450-457: x*w/sqrt(dot(x,x)/n+1e-5)) into the x staging; every
// block recomputes the 8 KB reduction from L2 instead of paying a kernel boundary for it.
// ------------------------------------------------------------------------------------------------
template <int WT, int EPI, bool NORM, int ROWS, int NCH>
__global__ __launch_bounds__(GEMV_THREADS) void gemv_kernel(GemvArgs a) {
    static_assert(ROWS == 2 || EPI == EPI_STORE || EPI == EPI_RESID, "row pairs required");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* red = reinterpret_cast<float*>(smem_raw);         // [4] (16 B)
    float4* xs = reinterpret_cast<float4*>(smem_raw + 16);   // x, K floats
    typedef typename WTraits<WT>::vec wvec;
    constexpr int VE = WTraits<WT>::VE;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int K = a.K;
    const int nx4 = K >> 2;        // float4s of x
    const int nvec = K / VE;       // weight vectors per row
    const int ncol = nvec / WAVE;  // full wave-wide vector columns
    const int njobs = (EPI == EPI_SWIGLU) ? a.H : (a.rows / ROWS);
    const size_t rb = WTraits<WT>::row_bytes(K);
    const char* Wb = reinterpret_cast<const char*>(a.W);
    const __half* Sb = nullptr;   // (the generic kernel is not instantiated for q4_0: gemv_q4_kernel below)

    const int g = blockIdx.x * GEMV_WAVES + wid;
    const bool active = g < njobs;                 // wave-uniform
    int rows[ROWS];
    const wvec* w[ROWS];
    const __half* sc[ROWS];
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
        rows[i] = (ROWS == 1) ? g : (EPI == EPI_SWIGLU ? g + i * a.H : 2 * g + i);
        const int r = active ? rows[i] : 0;
        w[i] = reinterpret_cast<const wvec*>(Wb + (size_t)r * rb);
        sc[i] = (WT == WT_Q4_0) ? Sb + (size_t)r * nvec : nullptr;
    }
    const bool pf = active && (ncol >= NCH);
    Tile<WT, ROWS, NCH> pre;

    // ---- stage x (optionally rmsnorm'ed) into LDS -------------------------------------------
    {
        const float4* xg = reinterpret_cast<const float4*>(a.x);
        float ss = 0.f;
        // x's loads go out first, GEMV_SU per thread at once, then the weight tile; x is staged while the weights stream
        // (round 4: one load per trip was a dependent L2 round trip per trip, the first behind the weight tile -- see gemv_q4_kernel)
        float4 v[GEMV_SU];
#define G_X_REQUEST(I0_) _Pragma("unroll") for (int u = 0; u < GEMV_SU; ++u) v[u] = xg[min((I0_) + u * GEMV_THREADS, nx4 - 1)];
#define G_X_STAGE(I0_)                                                          \
        _Pragma("unroll") for (int u = 0; u < GEMV_SU; ++u) {                    \
            const int i_ = (I0_) + u * GEMV_THREADS;                             \
            if (i_ < nx4) {                                                      \
                if (NORM) ss = dot4(v[u], v[u], ss);                             \
                if (WT == WT_Q4_0) xs[(i_ & 7) * nvec + (i_ >> 3)] = v[u];       \
                else xs[i_] = v[u];                                              \
            }                                                                    \
        }
        G_X_REQUEST(tid)
        if (pf) pre.load(w, sc, lane);
        G_X_STAGE(tid)
        for (int i0 = tid + GEMV_THREADS * GEMV_SU; i0 < nx4; i0 += GEMV_THREADS * GEMV_SU) {
            G_X_REQUEST(i0)
            G_X_STAGE(i0)
        }
#undef G_X_REQUEST
#undef G_X_STAGE
        if (NORM) {
            ss = wave_sum(ss);
            if (lane == 0) red[wid] = ss;
            __syncthreads();
            ss = red[0] + red[1] + red[2] + red[3];
            const float xn = sqrtf(ss / (float)K + a.eps);
            const float4* wg = reinterpret_cast<const float4*>(a.norm_w);
            for (int i0 = tid; i0 < nx4; i0 += GEMV_THREADS * GEMV_SU) {
                float4 nw[GEMV_SU];
#pragma unroll
                for (int u = 0; u < GEMV_SU; ++u) nw[u] = wg[min(i0 + u * GEMV_THREADS, nx4 - 1)];
#pragma unroll
                for (int u = 0; u < GEMV_SU; ++u) {
                    const int i = i0 + u * GEMV_THREADS;
                    if (i < nx4) {
                        const int li = (WT == WT_Q4_0) ? ((i & 7) * nvec + (i >> 3)) : i;
                        float4 xv = xs[li];
                        xv.x = xv.x * nw[u].x / xn;
                        xv.y = xv.y * nw[u].y / xn;
                        xv.z = xv.z * nw[u].z / xn;
                        xv.w = xv.w * nw[u].w / xn;
                        xs[li] = xv;
                    }
                }
            }
        }
        __syncthreads();
    }
    if (!active) return;

    float acc[ROWS];
#pragma unroll
    for (int i = 0; i < ROWS; ++i) acc[i] = 0.f;
    int v = lane, rem = ncol;
    if (pf) {
        pre.fma(xs, v, nvec, acc);
        v += NCH * WAVE;
        rem -= NCH;
    }
    while (rem >= NCH) { tile_step<WT, ROWS, NCH>(w, sc, xs, v, nvec, acc); v += NCH * WAVE; rem -= NCH; }
    if (NCH > 8 && rem >= 8) { tile_step<WT, ROWS, 8>(w, sc, xs, v, nvec, acc); v += 8 * WAVE; rem -= 8; }
    if (NCH > 4 && rem >= 4) { tile_step<WT, ROWS, 4>(w, sc, xs, v, nvec, acc); v += 4 * WAVE; rem -= 4; }
    if (NCH > 2 && rem >= 2) { tile_step<WT, ROWS, 2>(w, sc, xs, v, nvec, acc); v += 2 * WAVE; rem -= 2; }
    if (NCH > 1 && rem >= 1) { tile_step<WT, ROWS, 1>(w, sc, xs, v, nvec, acc); v += WAVE; }
    if (v < nvec) tile_step<WT, ROWS, 1>(w, sc, xs, v, nvec, acc);  // ragged tail (small shapes)
#pragma unroll
    for (int i = 0; i < ROWS; ++i) acc[i] = wave_sum(acc[i]);
    if (lane == 0) {
        if constexpr (ROWS == 2) {
            gemv_epilogue2<EPI>(a, g, rows[0], rows[1], acc[0], acc[1]);
        } else {
            if (EPI == EPI_RESID) a.y[g] += acc[0]; else a.y[g] = acc[0];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// q4_0 GEMV (ggml block_q4_0: 32 weights = f16 d + 16 nibble bytes; device layout: nibble plane + scale
// plane, see llmk_upload).  18 B per 32 weights makes this kernel ALU/LDS-heavy rather than purely
// HBM-bound, so it is organised around reuse:
//   * a wave owns NP row pairs (8 rows): the 32 activations of a block are read from LDS ONCE (8
//     ds_read_b128, transposed staging -> lane-contiguous) and used for all 8 rows;
//   * nibbles are widened with v_cvt_f32_ubyteN on (q & 0x0F0F0F0F) / ((q >> 4) & 0x0F0F0F0F): no per-weight
//     shift/mask/subtract; the "-8" of (nibble-8)*d is applied per block through the staged block sums
//     of x:  sum_i (n_i-8) d x_i = d (sum_i n_i x_i - 8 sum_i x_i);
//   * x is staged once per 32 rows (4 waves x 8) instead of once per 8.
// Same pairing rules and epilogues as gemv_kernel.
// ------------------------------------------------------------------------------------------------
// byte N of a dword as float: ONE v_cvt_f32_ubyteN.  Written as asm because the optimiser folds
// `((q & 0x0F0F0F0F) >> 8) & 0xFF` into `(q >> 8) & 0x0F`, which no longer matches the byte-convert pattern and costs
// a shift + an and + a convert per weight (measured: 3.2 VALU ops per weight instead of 1.75).
template <int N>
__device__ __forceinline__ float cvt_ubyte(unsigned v) {
    float f;
    if constexpr (N == 0) asm("v_cvt_f32_ubyte0 %0, %1" : "=v"(f) : "v"(v));
    else if constexpr (N == 1) asm("v_cvt_f32_ubyte1 %0, %1" : "=v"(f) : "v"(v));
    else if constexpr (N == 2) asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(f) : "v"(v));
    else asm("v_cvt_f32_ubyte3 %0, %1" : "=v"(f) : "v"(v));
    return f;
}
// One dword = bytes 4i..4i+3 of a block: low nibbles are elements 4i.., high nibbles elements 16+4i...
// lo += sum n_lo x ;  hi16 += sum (16 n_hi) x   -- the high nibbles are converted in place (q & 0xF0F0F0F0 = 16 n),
// the exact factor 1/16 is applied once per block by the caller.
// Round 3: no conversion -- the nibbles are used where they lie as f16 SUBNORMALS (0x000n = n * 2^-24, 0x00n0 = 16 n * 2^-24)
// through v_fma_mix_f32 (f16 source half selected by op_sel, f32 multiplicand and accumulator): 1 shift + 4 ands + 8
// fma_mix per dword instead of 2 ands + 8 half-rate v_cvt_f32_ubyteN + 8 fmas; chains in element order, every partial sum
// is the old one times 2^-24 exactly, so q4_block_fold's rescale gives bit-identical results (probes/q4_mix_probe.hip).
__device__ __forceinline__ void q4_dword_dot(unsigned q, const float4& xl, const float4& xh, float& lo, float& hi16) {
    unsigned l0, h0, l1, h1, s;
    asm("v_and_b32 %[l0], 0x000f000f, %[q]\n\t"
        "v_and_b32 %[h0], 0x00f000f0, %[q]\n\t"
        "v_lshrrev_b32 %[s], 8, %[q]\n\t"
        "v_and_b32 %[l1], 0x000f000f, %[s]\n\t"
        "v_fma_mix_f32 %[lo], %[l0], %[a0], %[lo] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %[hi], %[h0], %[b0], %[hi] op_sel_hi:[1,0,0]\n\t"
        "v_and_b32 %[h1], 0x00f000f0, %[s]\n\t"
        "v_fma_mix_f32 %[lo], %[l1], %[a1], %[lo] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %[hi], %[h1], %[b1], %[hi] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %[lo], %[l0], %[a2], %[lo] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %[hi], %[h0], %[b2], %[hi] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %[lo], %[l1], %[a3], %[lo] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %[hi], %[h1], %[b3], %[hi] op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : [lo] "+v"(lo), [hi] "+v"(hi16), [l0] "=&v"(l0), [h0] "=&v"(h0), [l1] "=&v"(l1), [h1] "=&v"(h1), [s] "=&v"(s)
        : [q] "v"(q), [a0] "v"(xl.x), [a1] "v"(xl.y), [a2] "v"(xl.z), [a3] "v"(xl.w), [b0] "v"(xh.x), [b1] "v"(xh.y),
          [b2] "v"(xh.z), [b3] "v"(xh.w));
}
// the two chains of a block (tl: low nibbles, th: 16 x high nibbles) -> sum n x, on the old recipe's scale
__device__ __forceinline__ float q4_block_fold(float tl, float th) {
    const float t = fmaf(th, 0.0625f, tl);
    return t * 16777216.0f;       // exact (a power of two)
}

template <int EPI, bool NORM, int NP, int KS = 1>
__global__ __launch_bounds__(GEMV_THREADS) void gemv_q4_kernel(GemvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* red = reinterpret_cast<float*>(smem_raw);          // [4]
    float4* xs = reinterpret_cast<float4*>(smem_raw + 16);    // [8][xp] float4 (transposed x), row pitch xp = nblk + 1
    constexpr int NR = 2 * NP;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int K = a.K, nx4 = K >> 2, nblk = K >> 5;
    // +1 float4 of pitch: the 8 lanes that stage one block's 8 vectors write 8 different rows of the transposed
    // array; at a pitch of nblk*16 bytes (a multiple of the bank width) they all hit the same banks (rocprofv3:
    // SQ_LDS_BANK_CONFLICT 5.6x SQ_ACTIVE_INST_LDS on the 7B shapes)
    const int xp = nblk + 1;
    float* xsum = reinterpret_cast<float*>(xs + 8 * xp);      // [nblk] block sums of the staged x
    const int npairs = (EPI == EPI_SWIGLU) ? a.H : (a.rows >> 1);
    constexpr int NGRP = GEMV_WAVES / KS;                     // row groups per block
    const int ksl = wid % KS;                                 // this wave's column slice
    const int g0 = (blockIdx.x * NGRP + wid / KS) * NP;       // first pair of this wave
    const int cb = nblk / KS, cb0 = ksl * cb;                 // its block columns [cb0, cb0 + cb)
    const char* Wb = reinterpret_cast<const char*>(a.W);
    const size_t RS = (size_t)a.row_stride;
    const int soff = K >> 1;                                  // a row's scales follow its K/2 nibble bytes

    int rows[NR];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int g = min(g0 + j, npairs - 1);                // clamped: past-the-end pairs recompute the last one
        rows[2 * j] = (EPI == EPI_SWIGLU) ? g : 2 * g;
        rows[2 * j + 1] = (EPI == EPI_SWIGLU) ? g + a.H : 2 * g + 1;
    }
    uint4 wq[NR];
    __half wd[NR];
    // ---- stage x: transposed float4 groups + per-block sums ------------------------------------
    float xn = 1.f;
    {
        const float4* xg = reinterpret_cast<const float4*>(a.x);
        const float4* wg = reinterpret_cast<const float4*>(a.norm_w);
        float ss = 0.f;
        // ONE pass: rmsnorm (llama2.f90:450-457) stages x*w -- the gains do not wait for the sum of squares, because the
        // division by sqrt(mean(x^2)+eps) is linear in the dot product and is applied once to each finished row sum
        // (round 3: the separate multiply pass over LDS and its barrier are gone; same products, same sums)
        // Round 4 (found in the ISA): written as one load per trip, hipcc waits for each -- `s_waitcnt vmcnt(0)` inside the loop:
        // K = 8192 was eight dependent L2 round trips, and vmcnt retires in order, so the first of them also waited for the
        // weight column requested ahead of it (an HBM round trip): ~5 us of the 6-12 us a 70B-rank GEMV took.  Now GEMV_SU
        // vectors of x (and of the gains) per thread are requested at once, AHEAD of the weights (x comes from L2 and is
        // needed first), and staged while the weights stream.  Same elements, same order per thread: the same sums.
        float4 v[GEMV_SU], nw[GEMV_SU];
        // (macros, not lambdas: captured by reference the two arrays went to scratch in the instantiations without NORM)
#define Q4_X_REQUEST(I0_)                                                       \
        _Pragma("unroll") for (int u = 0; u < GEMV_SU; ++u) {                    \
            const int i_ = min((I0_) + u * GEMV_THREADS, nx4 - 1);               \
            v[u] = xg[i_];                                                       \
            if (NORM) nw[u] = wg[i_];                                            \
        }
#define Q4_X_STAGE(I0_)                                                         \
        _Pragma("unroll") for (int u = 0; u < GEMV_SU; ++u) {                    \
            const int i_ = (I0_) + u * GEMV_THREADS;                             \
            if (i_ < nx4) {                                                      \
                if (NORM) {                                                      \
                    ss = dot4(v[u], v[u], ss);                                   \
                    v[u].x = v[u].x * nw[u].x;                                   \
                    v[u].y = v[u].y * nw[u].y;                                   \
                    v[u].z = v[u].z * nw[u].z;                                   \
                    v[u].w = v[u].w * nw[u].w;                                   \
                }                                                                \
                xs[(i_ & 7) * xp + (i_ >> 3)] = v[u];                            \
            }                                                                    \
        }
        Q4_X_REQUEST(tid)
        {   // first block column of all NR rows: requested before x is staged, behind x's own loads
            const int b = min(cb0 + lane, nblk - 1);
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const char* rp = Wb + (size_t)rows[i] * RS;
                wq[i] = ldg_nt(reinterpret_cast<const uint4*>(rp) + b);
                wd[i] = reinterpret_cast<const __half*>(rp + soff)[b];
            }
        }
        Q4_X_STAGE(tid)
        for (int i0 = tid + GEMV_THREADS * GEMV_SU; i0 < nx4; i0 += GEMV_THREADS * GEMV_SU) {   // K > 8192 (Llama-2-7B's w2: 11008)
            Q4_X_REQUEST(i0)
            Q4_X_STAGE(i0)
        }
#undef Q4_X_REQUEST
#undef Q4_X_STAGE
        if (NORM) {
            ss = wave_sum(ss);
            if (lane == 0) red[wid] = ss;
        }
        __syncthreads();
        if (NORM) {
            ss = red[0] + red[1] + red[2] + red[3];
            xn = sqrtf(ss / (float)K + a.eps);
        }
        // (block sums folded into the pass above with 7 shuffles per vector: measured slower, QKV 6.0 -> 7.0 us)
        for (int b = tid; b < nblk; b += GEMV_THREADS) {
            float t = 0.f;
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const float4 v = xs[m * xp + b];
                t += (v.x + v.y) + (v.z + v.w);
            }
            xsum[b] = t;
        }
        __syncthreads();
    }

    float acc[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) acc[i] = 0.f;
    for (int b0 = cb0; b0 < cb0 + cb; b0 += WAVE) {
        const int bl = b0 + lane;
        const bool live = bl < cb0 + cb;
        const int b = live ? bl : nblk - 1;
        // next column's loads go out before this column's math
        uint4 nq[NR];
        __half nd[NR];
        const bool more = b0 + WAVE < cb0 + cb;   // wave-uniform
        if (more) {
            const int nb = min(bl + WAVE, nblk - 1);
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const char* rp = Wb + (size_t)rows[i] * RS;
                nq[i] = ldg_nt(reinterpret_cast<const uint4*>(rp) + nb);
                nd[i] = reinterpret_cast<const __half*>(rp + soff)[nb];
            }
        }
        float4 xl[4], xh[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            xl[m] = xs[m * xp + b];
            xh[m] = xs[(4 + m) * xp + b];
        }
        const float xs8 = 8.0f * xsum[b];
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            float tl = 0.f, th = 0.f;
            q4_dword_dot(wq[i].x, xl[0], xh[0], tl, th);
            q4_dword_dot(wq[i].y, xl[1], xh[1], tl, th);
            q4_dword_dot(wq[i].z, xl[2], xh[2], tl, th);
            q4_dword_dot(wq[i].w, xl[3], xh[3], tl, th);
            const float t = q4_block_fold(tl, th);
            const float d = live ? __half2float(wd[i]) : 0.f;
            acc[i] = fmaf(d, t - xs8, acc[i]);
        }
        if (more) {
#pragma unroll
            for (int i = 0; i < NR; ++i) { wq[i] = nq[i]; wd[i] = nd[i]; }
        }
    }
#pragma unroll
    for (int i = 0; i < NR; ++i) acc[i] = wave_sum(acc[i]);
    if constexpr (KS > 1) {   // the row's KS column-slice sums, added in slice order by the group's first wave
        float* ksum = xsum;   // (the staged block sums are dead once every wave has left the loop)
        __syncthreads();
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < NR; ++i) ksum[wid * NR + i] = acc[i];
        }
        __syncthreads();
        if (ksl != 0) return;
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            float t = ksum[wid * NR + i];
#pragma unroll
            for (int k = 1; k < KS; ++k) t += ksum[(wid + k) * NR + i];
            acc[i] = t;
        }
    }
    if (NORM) {
#pragma unroll
        for (int i = 0; i < NR; ++i) acc[i] = acc[i] / xn;
    }
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < NP; ++j)
            if (g0 + j < npairs) gemv_epilogue2<EPI>(a, g0 + j, rows[2 * j], rows[2 * j + 1], acc[2 * j], acc[2 * j + 1]);
    }
}

// ------------------------------------------------------------------------------------------------
// Attention for one query head per block (llama2.f90:572-598 + softmax :468-478).
// GQA: head h reads kv head h/kv_mul (the intended semantics of the slice at :581/:591).
// Decode attention is a latency problem, not a bandwidth one (pos*512 B per head): every K (and
// V) vector a thread needs for a 16-deep batch is requested before the first one is used, with
// the timestep CLAMPED to pos-1 instead of predicated (a predicate would make hipcc branch around
// each load and drain vmcnt per element).  HS/4 lanes share one timestep (16-byte reads, one
// contiguous head row per timestep); the same lane mapping serves QK^T and PV.
// ------------------------------------------------------------------------------------------------
constexpr int ATT_U = 16;  // timestep batches in flight per thread

template <int HS>
__global__ __launch_bounds__(256) void attn_kernel(const float* __restrict__ q, const float* __restrict__ kc,
                                                   const float* __restrict__ vc, float* __restrict__ xb,
                                                   const int* __restrict__ tokpos, int KV, int kv_mul,
                                                   int pos0 = 0, int tstride = 0) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float4* red = reinterpret_cast<float4*>(smem_raw);            // [4 waves][HS/4] float4  (<= 2 KB)
    float* red4 = reinterpret_cast<float*>(smem_raw) + 512;       // [4]
    float* att = reinterpret_cast<float*>(smem_raw) + 516;        // [pos] (capacity S)
    constexpr int LPT = HS / 4;         // lanes per timestep
    constexpr int TPW = 64 / LPT;       // timesteps per wave-instruction
    constexpr int TPB = 4 * TPW;        // timesteps per block-instruction
    constexpr int TILE = TPB * ATT_U;   // timesteps per batch

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int h = blockIdx.x, g = h / kv_mul;
    // decode: one query at tokpos[1].  Prefill (pos0 > 0): blockIdx.y-th prompt position of the batch, its own q / xb rows
    const int pos = pos0 > 0 ? pos0 + (int)blockIdx.y : tokpos[1];
    q += (size_t)blockIdx.y * tstride;
    xb += (size_t)blockIdx.y * tstride;
    const int sub = lane % LPT, tl = lane / LPT;
    const float4 qv = reinterpret_cast<const float4*>(q + (size_t)h * HS)[sub];
    const float scale = sqrtf((float)HS);
    const float4* kg = reinterpret_cast<const float4*>(kc + (size_t)g * HS) + sub;
    const float4* vg = reinterpret_cast<const float4*>(vc + (size_t)g * HS) + sub;
    const int kv4 = KV >> 2;            // float4 stride between timesteps
    const int tb = wid * TPW + tl;      // this thread's timestep within a block-instruction

    // ---- scores: att[t] = q.k_t / sqrt(hs)                                       :578-583
    for (int base = 0; base < pos; base += TILE) {
        float4 kv[ATT_U];
#pragma unroll
        for (int u = 0; u < ATT_U; ++u) {
            const int t = min(base + u * TPB + tb, pos - 1);
            kv[u] = kg[(size_t)t * kv4];
        }
#pragma unroll
        for (int u = 0; u < ATT_U; ++u) {
            const int t = base + u * TPB + tb;
            float d = dot4(qv, kv[u], 0.f);
#pragma unroll
            for (int o = LPT / 2; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
            if (sub == 0 && t < pos) att[t] = d / scale;
        }
    }
    // first V batch is requested before the softmax: it does not depend on the scores
    float4 vv[ATT_U];
#pragma unroll
    for (int u = 0; u < ATT_U; ++u) vv[u] = vg[(size_t)min(u * TPB + tb, pos - 1) * kv4];
    __syncthreads();

    // ---- softmax over t < pos                                                    :468-478
    float m = -INFINITY;
    for (int t = tid; t < pos; t += 256) m = fmaxf(m, att[t]);
    m = wave_max(m);
    if (lane == 0) red4[wid] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red4[0], red4[1]), fmaxf(red4[2], red4[3]));
    __syncthreads();
    float s = 0.f;
    for (int t = tid; t < pos; t += 256) {
        const float e = expf(att[t] - m);
        att[t] = e;
        s += e;
    }
    s = wave_sum(s);
    if (lane == 0) red4[wid] = s;
    __syncthreads();
    s = red4[0] + red4[1] + red4[2] + red4[3];
    for (int t = tid; t < pos; t += 256) att[t] = att[t] / s;  // p(:s) = xi/sum(xi)   :476
    __syncthreads();

    // ---- xb_h = sum_t p_t * v_t                                                  :589-596
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int base = 0; base < pos; base += TILE) {
        if (base > 0) {
#pragma unroll
            for (int u = 0; u < ATT_U; ++u) vv[u] = vg[(size_t)min(base + u * TPB + tb, pos - 1) * kv4];
        }
#pragma unroll
        for (int u = 0; u < ATT_U; ++u) {
            const int t = base + u * TPB + tb;
            const float p = (t < pos) ? att[t] : 0.f;
            acc.x = fmaf(p, vv[u].x, acc.x);
            acc.y = fmaf(p, vv[u].y, acc.y);
            acc.z = fmaf(p, vv[u].z, acc.z);
            acc.w = fmaf(p, vv[u].w, acc.w);
        }
    }
#pragma unroll
    for (int o = LPT; o < 64; o <<= 1) {  // fold the TPW timestep groups of the wave
        acc.x += __shfl_xor(acc.x, o, 64);
        acc.y += __shfl_xor(acc.y, o, 64);
        acc.z += __shfl_xor(acc.z, o, 64);
        acc.w += __shfl_xor(acc.w, o, 64);
    }
    if (tl == 0) red[wid * LPT + sub] = acc;
    __syncthreads();
    if (tid < LPT) {
        float4 a = red[tid], b = red[LPT + tid], c = red[2 * LPT + tid], d = red[3 * LPT + tid];
        float4 o;
        o.x = (a.x + b.x) + (c.x + d.x);
        o.y = (a.y + b.y) + (c.y + d.y);
        o.z = (a.z + b.z) + (c.z + d.z);
        o.w = (a.w + b.w) + (c.w + d.w);
        reinterpret_cast<float4*>(xb + (size_t)h * HS)[tid] = o;
    }
}

// x = token_embedding_table(:,token)   llama2.f90:520
__global__ void embed_kernel(const float* __restrict__ table, const int* __restrict__ tokpos, float* __restrict__ x,
                             int E) {
    const int tok = tokpos[0];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < E; i += gridDim.x * blockDim.x)
        x[i] = table[(size_t)tok * E + i];
}

// token = maxloc(logits,DIM=1)   llama2.f90:388 -- first maximum wins; writes the 1-based id.
__global__ __launch_bounds__(1024) void argmax_kernel(const float* __restrict__ logits, int n, int* __restrict__ out) {
    __shared__ float bv[16];
    __shared__ int bi[16];
    float best = -INFINITY;
    int idx = 0x7fffffff;
    for (int i = threadIdx.x; i < n; i += 1024) {
        const float v = logits[i];
        if (v > best) { best = v; idx = i; }  // ascending i per thread: keeps the first maximum
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(idx, o, 64);
        if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) { bv[wid] = best; bi[wid] = idx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w)
            if (bv[w] > best || (bv[w] == best && bi[w] < idx)) { best = bv[w]; idx = bi[w]; }
        out[0] = (unsigned)idx < (unsigned)n ? idx + 1 : 0;   // no finite maximum (all NaN / -inf): 0 is no token -- the next call rejects it (LLMK_E_ARG)
    }
}

// q4_0 re-pack on upload: ggml blocks {f16 d; u8 qs[16]} (18 B, 2-byte aligned), bpr per row -> device rows of
// row_stride bytes: bpr 16-byte aligned nibble vectors (so the GEMV issues dwordx4 loads), then the row's bpr f16 scales.
__global__ void q4_repack_kernel(const uint8_t* __restrict__ src, char* __restrict__ dst, size_t nblocks, int bpr,
                                 size_t row_stride) {
    for (size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x; b < nblocks; b += (size_t)gridDim.x * blockDim.x) {
        const uint16_t* p = reinterpret_cast<const uint16_t*>(src + b * 18);
        const size_t row = b / bpr;
        const int j = (int)(b - row * bpr);
        char* rp = dst + row * row_stride;
        reinterpret_cast<uint16_t*>(rp + (size_t)bpr * 16)[j] = p[0];
        uint4 v;
        v.x = (unsigned)p[1] | ((unsigned)p[2] << 16);
        v.y = (unsigned)p[3] | ((unsigned)p[4] << 16);
        v.z = (unsigned)p[5] | ((unsigned)p[6] << 16);
        v.w = (unsigned)p[7] | ((unsigned)p[8] << 16);
        reinterpret_cast<uint4*>(rp)[j] = v;
    }
}

}  // namespace llmk
