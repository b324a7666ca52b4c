// Batched prompt prefill for gfx950: T prompt positions go through each
// layer TOGETHER, so a layer's weights cross HBM once per batch instead of once per token
// (/root/reference/llama2.f90:376-402 feeds the prompt one token at a time through `transformer`, :480-640,
// and discards the logits of every position but the last).  Same arithmetic per position as the
// decode path (rmsnorm :450-457, RoPE :543-559 with the 1-based pos, GQA attention :572-598, SwiGLU
// :615-616); only the order of the dot-product partial sums differs.
//
// The weight GEMMs are the one place on this path where the matrix cores are the right tool: Y[T][rows] =
// X[T][K] . W[rows][K]^T has 2*T flop per weight word.  Default since round 3: pf_gemm_h_kernel (further down) --
// v_mfma_f32_16x16x32_f16 with every f32 operand fed as two f16 pieces, exact products, 16x the products per clock.
// pf_gemm_kernel (v_mfma_f32_16x16x4_f32, full f32 operands, 256 flop/clk/CU) is the path of LLMK_PF_F32_MFMA=1 and of
// prompts whose activations exceed the f16 range.  Parity bar either way: 1e-4 relative on logits.
//
// pf_gemm: 4 waves per block; wave w owns weight rows strip*64+16w .. +15 of the block's units, all four share the
// units' activations through LDS (up to 128 tokens x 64 columns per step, double-buffered).
// Lane l holds W[row + l%16][k + 4*(l/16) .. +3] (16 bytes; the 4 lanes of a row cover one 64-byte line) and, per
// 16-token group, X[t0 + l%16][same columns]: MFMA step j multiplies component j of both -- k is a summation
// index, so any lane->k assignment is valid as long as A and B agree.  Weight loads run two 64-column steps ahead of the
// MFMAs.  Each block writes one partial tile P[slot][t][row] per strip it touches; the epilogue kernel adds a strip's
// partials in slot order (deterministic) and applies the fused tail (RoPE + KV write / residual / SwiGLU).
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "kernels.h"

namespace llmk {

constexpr int PF_TMAX = 128;         // prompt positions per pass (8 MFMA token groups)
constexpr int PF_WAVES = 4;

typedef float pf_v4f __attribute__((ext_vector_type(4)));

// Work of one GEMM = strips (64 weight rows) x nk (64-column steps) UNITS, strip-major.  Block b owns the U consecutive
// units [b*U, (b+1)*U): every block multiplies the same number of 64x64 weight tiles whatever rows/K are, and the grid is
// the number of workgroups the chip holds at once (1 or 2 per CU, pinned by the LDS request) -- with a strips x slices
// grid the 528 / 880 blocks of the w1|w3 GEMM left 16 / 112 CUs with one block more than the others and the kernel
// waited 1.5x the median block for them (profiles/README.md, round 2).  A block whose range crosses a strip boundary
// writes one partial tile per strip it touches: partial `slot` of strip s comes from block floor(s*nk/U) + slot.
struct PfGemmArgs {
    const void* W;       // [rows][K]  f32 or f16 (template parameter WT)
    const float* X;      // [T][K]   (activations, L2-resident)
    float* P;            // [slots][Tp][rows] partial sums, Tp = 16*NG
    int rows, K, T;
    int nk;              // 64-column steps per strip = K / PF_KSTEP
    int U;               // units per block
    int total;           // strips * nk   (strip = 64 rows, or 128 with two row groups per wave)
    int RS;              // q4_0: bytes between rows (K/2 nibble bytes, then the row's f16 scales: llmk.hip q4_row_stride)
#ifdef LLMK_PF_TRACE
    unsigned long long* trace;   // [grid][20] wall-clock stamps: entry, prologue done, end of steps 0..14; [17] exit, [18],[19] shader clock at entry / exit
#endif
};

constexpr int PF_KSTEP = 64;                 // columns per pipeline step (4 MFMA chunks of 16)
constexpr int PF_LDW = PF_KSTEP + 4;         // LDS row pitch in floats (+16 bytes: the 16 tokens of a read spread over banks)

__host__ __device__ inline int pf_first_block(int strip, int nk, int U) { return strip * nk / U; }
__host__ __device__ inline int pf_nslots(int strip, int nk, int U) { return ((strip + 1) * nk - 1) / U - strip * nk / U + 1; }

// The block's partial tile goes to P through LDS.  In the accumulators (D layout of 16x16x4: lane l, register v <-> weight
// row 4*(l/16)+v, token l%16) a wave's store instruction would touch 16 positions x 64 bytes; written to LDS as
// [position][strip row] and read back row-major, every store instruction writes whole 512-byte (256 with one row group)
// runs of P[slot][position][rows].  All CUs flush at about the same time: scattered, a workgroup's 64 KB took 2.2 us.
constexpr int PF_TPAD = 4;     // floats of padding per position in the transpose buffer
// f16 weights: a lane's 16 bytes are 8 columns, so a chunk is 32 columns (8 MFMA steps after the exact half -> float
// conversion) and a 64-column step has two chunks; the activation side is f32 either way.
//
// NR = 16-row groups per wave.  NR 1: strips of 64 rows, weights requested TWO steps ahead (ring of three register
// stages).  NR 2: strips of 128 rows (rows % 128 == 0 required: no ragged strip), every activation fragment read from LDS
// feeds two MFMAs -- a step is twice the matrix work for the same LDS reads, barrier and staging, so the fixed cost per
// step weighs half; one step (>= 3.4 us of MFMAs) is enough lead for the weights, which keeps the ring at two stages and
// the kernel at two waves per SIMD.
template <int NG, int WT, int NR>
__global__ __launch_bounds__(PF_WAVES * WAVE) void pf_gemm_kernel(PfGemmArgs a) {
    constexpr int NW = PF_WAVES;
    constexpr int TP = NG * 16;
    constexpr bool Q4 = WT == WT_Q4_0;
    constexpr int BW = (WT == WT_F16) ? 2 : 4;          // bytes per weight (f32 / f16)
    constexpr int CPL = Q4 ? 16 : 16 / BW;                // columns per lane load: 4 / 8 / 16 (half a q4_0 block)
    constexpr int CW = 4 * CPL;                           // columns per chunk (4 lane groups): 16 / 32 / 64
    constexpr int NJ = PF_KSTEP / CW;                     // chunks per step: 4 / 2 / 1
    constexpr int SR = 16 * NR * NW;                      // rows per strip
    constexpr int NT = NW * WAVE;                         // threads
    extern __shared__ __attribute__((aligned(16))) char pf_smem[];
    float* xs = reinterpret_cast<float*>(pf_smem);        // [2][TP][PF_LDW]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int u0 = blockIdx.x * a.U, nsteps = min(a.U, a.total - u0);
    if (nsteps <= 0) return;                          // whole block: no barrier is pending
    const int li = lane & 15, lk = (lane >> 4) * CPL;
    // q4_0: a 64-column step is two 32-weight blocks; lane group q = lane/16 takes columns 16q .. 16q+15 = the low (q even)
    // or high (q odd) nibbles of all 16 bytes of block q/2 -- both groups of a block load the same 16 bytes
    const size_t rowb = Q4 ? (size_t)a.RS : (size_t)a.K * BW;
    const char* wbase = static_cast<const char*>(a.W) + (Q4 ? (size_t)(lane >> 5) * 16 : (size_t)lk * BW);
    constexpr int STEPB = Q4 ? 32 : PF_KSTEP * BW;        // weight bytes per row per step
    const bool hi_nib = (lane >> 4) & 1;
    const unsigned nib_mask = hi_nib ? 0xF0F0F0F0u : 0x0F0F0F0Fu;
    // cursors over the block's units: weights run ahead of the MFMAs, activations one step
    int cs = u0 / a.nk, ck = u0 % a.nk;                   // strip / column step being multiplied
    int ws = cs, wk = ck, wi = 0, xk = ck, xi = 0;
#define wkl wk
    const char* wp = wbase + (size_t)min(ws * SR + wid * (16 * NR) + li, a.rows - 1) * rowb + (size_t)wk * STEPB;
    bool active = cs * SR + wid * (16 * NR) < a.rows;     // ragged last strip (NR 1): idle waves still take the barriers
    // activation staging: thread -> (token, 16-byte column group) of the TP x 64 tile, TP*16/256 vectors per thread
    constexpr int XN = TP * (PF_KSTEP / 4), XV = (XN + NT - 1) / NT;   // XN % NT = 0 or NT/2 (never with 4 waves)
    const bool xlast = (XV - 1) * NT + tid < XN;
    const float* xg[XV];
    int xo[XV];
#pragma unroll
    for (int i = 0; i < XV; ++i) {
        const int idx = min(tid + i * NT, XN - 1), t = idx / (PF_KSTEP / 4), c4 = idx % (PF_KSTEP / 4);
        xg[i] = a.X + (size_t)min(t, a.T - 1) * a.K + c4 * 4;        // pad tokens re-read the last row
        xo[i] = t * PF_LDW + c4 * 4;
    }

    pf_v4f acc[NR][NG];
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int g = 0; g < NG; ++g) acc[r][g] = (pf_v4f){0.f, 0.f, 0.f, 0.f};

    // The register stages have names and the loop is written out once per stage.  Every load is unconditional (the
    // cursors stop on the block's last unit) and the cursors advance by selects, not branches: hipcc's waitcnt pass
    // merges the pending-load state of both arms of a branch, and with conditional steps it concluded that the stage
    // about to be multiplied might be the newest load -- s_waitcnt vmcnt(0) at the top of every step, no weights in
    // flight across a step at all (round-2 ISA reading).
    float4 w0[NR * NJ], w1[NR * NJ], w2[NR == 1 ? NJ : 1];
    float w0d[NR], w1d[NR], w2d[1];                      // q4_0: the block scale that belongs to the stage
    pf_v4f xr[XV];
#define PF_WLOAD_PART(W_, LO_, HI_)                                                                                  \
    do {                                                                                                             \
        _Pragma("unroll") for (int q = (LO_); q < (HI_); ++q) {                                                      \
            W_[q] = ldg_nt(reinterpret_cast<const float4*>(wp + (size_t)(q / NJ) * 16 * rowb + (Q4 ? 0 : (q % NJ) * CW * BW))); \
            if constexpr (Q4) {                                                                                      \
                const char* rp_ = wp + (size_t)q * 16 * rowb - (size_t)(lane >> 5) * 16 - (size_t)wkl * 32;           \
                W_##d[q] = __half2float(reinterpret_cast<const __half*>(rp_ + (a.K >> 1))[2 * wkl + (lane >> 5)]);   \
            }                                                                                                        \
        }                                                                                                            \
    } while (0)
#define PF_WNEXT()                                                                                                   \
    do {                                                                                                             \
        const int adv_ = wi < nsteps - 1 ? 1 : 0;                                                                    \
        wi += adv_;                                                                                                  \
        wk += adv_;                                                                                                  \
        const bool wrap_ = wk == a.nk;                                                                               \
        wk = wrap_ ? 0 : wk;                                                                                         \
        ws += wrap_ ? 1 : 0;                                                                                         \
        const char* nxt_ = wbase + (size_t)min(ws * SR + wid * (16 * NR) + li, a.rows - 1) * rowb;                   \
        wp = wrap_ ? nxt_ : wp + adv_ * STEPB;                                                                       \
    } while (0)
#define PF_WLOAD(W_)                                                                                                 \
    do {                                                                                                             \
        PF_WLOAD_PART(W_, 0, NR * NJ);                                                                               \
        PF_WNEXT();                                                                                                  \
    } while (0)
#define PF_XLOAD_PART(LO_, HI_)                                                                                      \
    do {                                                                                                             \
        _Pragma("unroll") for (int i = (LO_); i < (HI_); ++i) xr[i] = *reinterpret_cast<const pf_v4f*>(xg[i] + xk * PF_KSTEP); \
    } while (0)
#define PF_XNEXT()                                                                                                   \
    do {                                                                                                             \
        const int adv_ = xi < nsteps - 1 ? 1 : 0;                                                                    \
        xi += adv_;                                                                                                  \
        xk += adv_;                                                                                                  \
        xk = xk == a.nk ? 0 : xk;                                                                                    \
    } while (0)
#define PF_XLOAD()                                                                                                   \
    do {                                                                                                             \
        PF_XLOAD_PART(0, XV);                                                                                        \
        PF_XNEXT();                                                                                                  \
    } while (0)
#define PF_XSTORE(BUF_)                                                                                              \
    do {                                                                                                             \
        _Pragma("unroll") for (int i = 0; i < XV; ++i)                                                               \
            if (i < XV - 1 || XN % NT == 0 || xlast) *reinterpret_cast<pf_v4f*>(xs + (BUF_) * TP * PF_LDW + xo[i]) = xr[i]; \
    } while (0)
    // a quarter of a step: 16 columns, one float4 of activations per token group (f16: half of a 32-column chunk)
    auto piece = [&](const float4 (&wc)[NR * NJ], const float (&wd)[NR == 1 ? 1 : NR], int buf, int pc) {
        const float* xb = xs + buf * TP * PF_LDW;
        const int j = pc / (CPL / 4), h = pc % (CPL / 4);
        float4 w[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            if constexpr (Q4) {
                // (n - 8) * d exactly as the decode path computes it: n * d - 8 d in f32; the high nibbles are read as 16 n
                const unsigned qd = (h == 0 ? __float_as_uint(wc[r].x) : h == 1 ? __float_as_uint(wc[r].y) : h == 2 ? __float_as_uint(wc[r].z) : __float_as_uint(wc[r].w)) & nib_mask;
                const float d = hi_nib ? wd[r] * 0.0625f : wd[r], m8 = -8.0f * wd[r];
                w[r] = make_float4(fmaf(cvt_ubyte<0>(qd), d, m8), fmaf(cvt_ubyte<1>(qd), d, m8), fmaf(cvt_ubyte<2>(qd), d, m8), fmaf(cvt_ubyte<3>(qd), d, m8));
            } else if constexpr (WT == WT_F16) {
                const __half2 p0 = *reinterpret_cast<const __half2*>(h == 0 ? &wc[r * NJ + j].x : &wc[r * NJ + j].z);
                const __half2 p1 = *reinterpret_cast<const __half2*>(h == 0 ? &wc[r * NJ + j].y : &wc[r * NJ + j].w);
                w[r] = make_float4(__low2float(p0), __high2float(p0), __low2float(p1), __high2float(p1));
            } else {
                w[r] = wc[r * NJ + j];
            }
        }
        float4 x[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) x[g] = *reinterpret_cast<const float4*>(xb + (g * 16 + li) * PF_LDW + j * CW + lk + 4 * h);
        // component-major: the NR*NG accumulators are independent chains the matrix core can interleave
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int g = 0; g < NG; ++g) acc[r][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[r].x, x[g].x, acc[r][g], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int g = 0; g < NG; ++g) acc[r][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[r].y, x[g].y, acc[r][g], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int g = 0; g < NG; ++g) acc[r][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[r].z, x[g].z, acc[r][g], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int g = 0; g < NG; ++g) acc[r][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[r].w, x[g].w, acc[r][g], 0, 0, 0);
    };
#ifdef LLMK_PF_TRACE
    unsigned long long* tr = a.trace + (size_t)blockIdx.x * 20;
    if (tid == 0) { tr[0] = wall_clock64(); tr[18] = __builtin_readcyclecounter(); }
#define PF_STAMP(I_) if (tid == 0 && (I_) < 17) tr[I_] = wall_clock64()
#else
#define PF_STAMP(I_)
#endif
    PF_WLOAD(w0);
    if constexpr (NR == 1) PF_WLOAD(w1);
    PF_XLOAD();
    PF_XSTORE(0);
    PF_XLOAD();
    __syncthreads();
    PF_STAMP(1);
    // step s: publish the activations of s+1 (requested a whole step ago: nothing to wait for; the buffer they go to
    // was last read in step s-1, which every wave has left), request the activations of s+2 and the weights the ring has
    // room for (in this order: vmcnt retires in order and the activations are needed first), multiply step s; at the end
    // of a strip (or of the block's range) write the partial tile and start the next strip.  The step count is padded to a multiple
    // of the ring depth; a padding step re-requests the last unit and multiplies nothing.
#define PF_STEP(S_, CUR_, NXT_)                                                                                      \
    {                                                                                                                \
        const bool mul_ = active && (S_) < nsteps;                                                                   \
        PF_XSTORE(((S_) + 1) & 1);                                                                                   \
        if (mul_) piece(CUR_, CUR_##d, (S_) & 1, 0);                                                                          \
        PF_XLOAD_PART(0, XV / 2);                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        if (mul_) piece(CUR_, CUR_##d, (S_) & 1, 1);                                                                          \
        PF_XLOAD_PART(XV / 2, XV);                                                                                   \
        PF_XNEXT();                                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        if (mul_) piece(CUR_, CUR_##d, (S_) & 1, 2);                                                                          \
        PF_WLOAD_PART(NXT_, 0, NR * NJ / 2);                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        if (mul_) piece(CUR_, CUR_##d, (S_) & 1, 3);                                                                          \
        PF_WLOAD_PART(NXT_, NR * NJ / 2, NR * NJ);                                                                   \
        PF_WNEXT();                                                                                                  \
        if ((S_) < nsteps && (++ck == a.nk || (S_) == nsteps - 1)) {                                                 \
            const int slot = (int)blockIdx.x - pf_first_block(cs, a.nk, a.U);                                        \
            float* tb_ = xs + 2 * TP * PF_LDW;                      /* [TP][SR + PF_TPAD] */                           \
            _Pragma("unroll") for (int r = 0; r < NR; ++r)                                                           \
                _Pragma("unroll") for (int g = 0; g < NG; ++g) {                                                     \
                    *reinterpret_cast<pf_v4f*>(tb_ + (g * 16 + li) * (SR + PF_TPAD) + wid * (16 * NR) + r * 16 + (lane >> 4) * 4) = acc[r][g]; \
                    acc[r][g] = (pf_v4f){0.f, 0.f, 0.f, 0.f};                                                        \
                }                                                                                                    \
            __syncthreads();                                                                                         \
            float* dst_ = a.P + (size_t)slot * TP * a.rows + cs * SR;                                                \
            _Pragma("unroll") for (int k = 0; k < TP * SR / 4 / NT; ++k) {                                           \
                const int idx_ = tid + k * NT, t_ = idx_ / (SR / 4), c_ = idx_ % (SR / 4) * 4;                       \
                const pf_v4f v_ = *reinterpret_cast<const pf_v4f*>(tb_ + t_ * (SR + PF_TPAD) + c_);                  \
                if (cs * SR + c_ < a.rows) *reinterpret_cast<pf_v4f*>(dst_ + (size_t)t_ * a.rows + c_) = v_;         \
            }                                                                                                        \
            ck = 0;                                                                                                  \
            ++cs;                                                                                                    \
            active = cs * SR + wid * (16 * NR) < a.rows;                                                             \
        }                                                                                                            \
        __syncthreads();                                                                                             \
        PF_STAMP(2 + (S_));                                                                                          \
    }
    if constexpr (NR == 1) {
        for (int s = 0; s < nsteps; s += 3) {
            PF_STEP(s, w0, w2)
            PF_STEP(s + 1, w1, w0)
            PF_STEP(s + 2, w2, w1)
        }
    } else {
        for (int s = 0; s < nsteps; s += 2) {
            PF_STEP(s, w0, w1)
            PF_STEP(s + 1, w1, w0)
        }
    }
#ifdef LLMK_PF_TRACE
    if (tid == 0) { tr[17] = wall_clock64(); tr[19] = __builtin_readcyclecounter(); }     // shader clock = (19 - 18) / (17 - 0)
#endif
#undef PF_STEP
#undef PF_STAMP
#undef PF_WLOAD
#undef PF_WLOAD_PART
#undef PF_WNEXT
#undef PF_XLOAD
#undef PF_XLOAD_PART
#undef PF_XNEXT
#undef wkl
#undef PF_XSTORE
}

// ---- f16 weights on the f16 matrix instruction (round 3) ------------------------------------------------------------------
// v_mfma_f32_16x16x32_f16 does 16x16x32 in 16 clocks where v_mfma_f32_16x16x4_f32 needs 8 x 32: an f16 weight matrix is
// ALREADY the A operand (a lane's 16 bytes = 8 consecutive columns of its row: no conversion), and the f32 activation is
// fed as TWO f16 pieces, x = hi + lo with hi = f16(x), lo = f16(x - hi): two instructions per 32 columns instead of eight,
// every product exact in f32 (11 x 11 significand bits), |x - hi - lo| <= 2^-20 |x| (2^-24 absolute where lo is an f16
// subnormal, |x| < 2^-3: the instruction honours subnormal inputs, csrc/probes/mfma_f16_denorm_probe.hip) -- the f32
// accumulation order is the matrix core's either way.  Activations of 65504 and above do not fit hi: the staging raises
// `flag` and llmk_prefill redoes the call on the f32 instruction (pf_gemm_kernel<.., WT_F16, ..>).
// Same units / strips / partial tiles as pf_gemm_kernel; a step (64 columns) is 4 x NR x NG instructions = 1,024 matrix
// clocks at 128 positions with two row groups; the weights run ST - 1 steps ahead (ring of register stages; depth: see ST in the kernel).
typedef _Float16 pf_v8h __attribute__((ext_vector_type(8)));
typedef _Float16 pf_v4h __attribute__((ext_vector_type(4)));
// LDS image of a step's activations: [hi, lo][position][64 halfs], 128-byte rows, the 16-byte slot s of position t stored at
// slot s ^ ((t >> 1) & 7).  ds_read_b128 is served in four NON-contiguous 16-lane groups ({0-3, 12-15, 20-27}, ...:
// MI355X_MICROARCH.md, LDS): a group holds all 16 positions of a fragment, half of them with the neighbouring column slot,
// and with this swizzle its 16 addresses fall into 16 different 4-bank slots (a padded pitch of 144 bytes left 7 of the 16
// pairs on one slot: two LDS cycles per group, and the LDS, not the matrix core, set the step time).
constexpr int PF_HP = PF_KSTEP;

template <int Q, int N, class F>
__device__ __forceinline__ void pf_prologue(F& wload) {
    if constexpr (Q < N) { wload(std::integral_constant<int, Q>()); pf_prologue<Q + 1, N>(wload); }
}
// N steps written out over a ring of R stages: step s+Q multiplies stage Q % R and refills the stage step s+Q-1 used.  The
// steps nest (a step past the block's last unit skips everything behind it: forward branches only, no join at a step's top
// for hipcc's waitcnt pass to merge pending loads at).
template <int Q, int N, int R, class F>
__device__ __forceinline__ void pf_ring(F& step, int s, int nsteps) {      // (N and the trip stride are even: Q & 1 is the step's parity)
    if constexpr (Q < N) {
        if (s + Q < nsteps) {
            step(s + Q, std::integral_constant<int, Q % R>(), std::integral_constant<int, (Q + R - 1) % R>(), std::integral_constant<int, Q & 1>());
            pf_ring<Q + 1, N, R>(step, s, nsteps);
        }
    }
}
// q4_0 weights (WT_Q4_0) on the same instruction: a 32-column chunk is one block; lane (row li, k group kg) takes the low
// (kg < 2) or high nibbles of bytes 8 (kg & 1) .. + 7 of the block = its elements 16 (kg >> 1) + 8 (kg & 1) .. + 7 = the 8
// consecutive columns of slot kg (the activation side is the f16 kernel's), turns them into the reference's f32 weights
// (n - 8) d exactly (n - 8 as f16: byte n under 0x64 is the half 1024 + n, minus 1032; times the row's f16 block scale through
// v_fma_mix_f32) and feeds each as two f16 pieces: see the step.
// NWV = waves per workgroup: 4 (one per SIMD), or 8 with NR = 1 -- the strip of the 4-wave NR = 2 form (same units, same partial tiles)
// on two waves per SIMD; used for q4_0 weights (llmk.hip pf_gemm_launch)
template <int NG, int NR, int WT = WT_F16, int NWV = PF_WAVES>
__global__ __launch_bounds__(NWV * WAVE) void pf_gemm_h_kernel(PfGemmArgs a, unsigned* __restrict__ flag, unsigned* __restrict__ lowcnt) {
    constexpr int NW = NWV, TP = NG * 16, NJ = 2, SR = 16 * NR * NW, NT = NW * WAVE;
    constexpr bool Q4 = WT == WT_Q4_0, F32W = WT == WT_F32;
    // ring depth of the weight stages.  Round 5: SHALLOWER is faster -- six f16 stages left ~100 values parked in accumulation registers and
    // moved in and out inside the step (256 VGPRs + 170 AGPRs for 64 accumulators; three stages: 254 + 92), and what they bought in lead they lost
    // again (f16: six -> three stages +5.6 % prompt rate, f32 four / three -> two +1.8 %, q4_0 on eight waves
    // four -> two +2.3 % at 7B; profiles/r05_prefill_ring_depth.txt)
#ifndef LLMK_PF_ST_Q8
#define LLMK_PF_ST_Q8 2
#endif
#ifndef LLMK_PF_ST_Q4
#define LLMK_PF_ST_Q4 2
#endif
#ifndef LLMK_PF_ST_F32
#define LLMK_PF_ST_F32 2
#endif
#ifndef LLMK_PF_ST_F16
#define LLMK_PF_ST_F16 3
#endif
    constexpr int ST = Q4 ? (NWV == 8 ? LLMK_PF_ST_Q8 : LLMK_PF_ST_Q4) : F32W ? LLMK_PF_ST_F32 : LLMK_PF_ST_F16;
    extern __shared__ __attribute__((aligned(16))) char pf_smem[];
    _Float16* xh = reinterpret_cast<_Float16*>(pf_smem);                                  // [2 buffers][hi, lo][TP][PF_HP]
    float* tb = reinterpret_cast<float*>(pf_smem + (size_t)4 * TP * PF_HP * sizeof(_Float16));   // [TP][SR + PF_TPAD]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int u0 = blockIdx.x * a.U, nsteps = min(a.U, a.total - u0);
    if (nsteps <= 0) return;
    const int li = lane & 15, kg = lane >> 4;
    const size_t rowb = Q4 ? (size_t)a.RS : (size_t)a.K * (F32W ? 4 : 2);
    const char* wbase = static_cast<const char*>(a.W) + (Q4 ? (size_t)(kg & 1) * 8 : (size_t)kg * (F32W ? 32 : 16));
    constexpr int STEPB = Q4 ? 32 : PF_KSTEP * (F32W ? 4 : 2);           // weight bytes per row per step
    int cs = u0 / a.nk, ck = u0 % a.nk;
    int ws = cs, wk = ck, wi = 0, xk = ck, xi = 0;
    const char* wp = wbase + (size_t)min(ws * SR + wid * (16 * NR) + li, a.rows - 1) * rowb + (size_t)wk * STEPB;
    constexpr int XV = TP * (PF_KSTEP / 4) / NT;          // float4 per thread per step (= NG)
    static_assert(XV * NT == TP * (PF_KSTEP / 4), "whole vectors per thread");
    int xg[XV];               // element offsets into X (T * K < 2^31)
    int xo[XV];
#pragma unroll
    for (int i = 0; i < XV; ++i) {
        const int idx = tid + i * NT, t = idx / (PF_KSTEP / 4), c4 = idx % (PF_KSTEP / 4);
        xg[i] = min(t, a.T - 1) * a.K + c4 * 4;
        xo[i] = t * PF_HP + (((c4 >> 1) ^ ((t >> 1) & 7)) << 3) + (c4 & 1) * 4;
    }
    pf_v4f acc[NR][NG];
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int g = 0; g < NG; ++g) acc[r][g] = (pf_v4f){0.f, 0.f, 0.f, 0.f};
    // a stage: f16 -- 16 bytes per chunk and row group
    typedef unsigned pf_v2u __attribute__((ext_vector_type(2)));
    float4 w[Q4 ? 1 : ST][Q4 ? 1 : NR * NJ * (F32W ? 2 : 1)];     // f32: a lane's 8 columns of a chunk are two 16-byte loads
    pf_v2u wq[Q4 ? ST : 1][Q4 ? NR * NJ : 1];        // q4_0: the lane's 8 nibble bytes per block and row group
    unsigned wd[Q4 ? ST : 1][Q4 ? NR : 1];          //       the row's two block scales (f16 pair)
    // activations of two steps in flight: requested a whole step before they are published, published in the shadow of the
    // matrix instructions
    constexpr bool XR2 = true;
    pf_v4f xr[XR2 ? 2 : 1][XV];
#ifdef LLMK_PF_TRACE
    unsigned long long* tr = a.trace + (size_t)blockIdx.x * 20;
    if (tid == 0) { tr[0] = wall_clock64(); tr[18] = __builtin_readcyclecounter(); }
#define PF_STAMP(I_) if (tid == 0 && (I_) < 17) tr[I_] = wall_clock64()
#else
#define PF_STAMP(I_)
#endif

    auto wload = [&](auto stage) {
        constexpr int Q = decltype(stage)::value;
#pragma unroll
        for (int q = 0; q < NR * NJ; ++q) {
            if constexpr (Q4) {
                const char* rp = wp + (size_t)(q / NJ) * 16 * rowb;
                wq[Q][q] = __builtin_nontemporal_load(reinterpret_cast<const pf_v2u*>(rp + (q % NJ) * 16));
                if (q % NJ == 0)        // the scales of blocks 2 wk and 2 wk + 1 of this row: K/2 nibble bytes in, 2 bytes per block
                    wd[Q][q / NJ] = __builtin_nontemporal_load(reinterpret_cast<const unsigned*>(rp - (size_t)(kg & 1) * 8 - (size_t)wk * 32 + (a.K >> 1) + (size_t)wk * 4));
            } else if constexpr (F32W) {
                w[Q][2 * q] = ldg_nt(reinterpret_cast<const float4*>(wp + (size_t)(q / NJ) * 16 * rowb + (q % NJ) * 128));
                w[Q][2 * q + 1] = ldg_nt(reinterpret_cast<const float4*>(wp + (size_t)(q / NJ) * 16 * rowb + (q % NJ) * 128 + 16));
            } else {
                w[Q][q] = ldg_nt(reinterpret_cast<const float4*>(wp + (size_t)(q / NJ) * 16 * rowb + (q % NJ) * 64));
            }
        }
        const int adv = wi < nsteps - 1 ? 1 : 0;
        wi += adv;
        wk += adv;
        const bool wrap = wk == a.nk;
        wk = wrap ? 0 : wk;
        ws += wrap ? 1 : 0;
        const char* nxt = wbase + (size_t)min(ws * SR + wid * (16 * NR) + li, a.rows - 1) * rowb;
        wp = wrap ? nxt : wp + adv * STEPB;
    };
    auto xload = [&](auto par) {
        constexpr int B_ = decltype(par)::value;
#pragma unroll
        for (int i = 0; i < XV; ++i) xr[B_][i] = *reinterpret_cast<const pf_v4f*>(a.X + xg[i] + xk * PF_KSTEP);
        const int adv = xi < nsteps - 1 ? 1 : 0;
        xi += adv;
        xk += adv;
        xk = xk == a.nk ? 0 : xk;
    };
    // vectors [LO, HI) of a step's activations: split and published.  Five VALU operations per PAIR of elements (the split
    // of a step's tile was 256 of a wave's ~500 VALU operations per step, against 64 matrix instructions that cover 190):
    // hi = both halves in one v_cvt_pkrtz (toward zero: the remainder then has up to 11 significant bits and the same sign),
    // x - hi through v_fma_mix_f32 with the f16 half as a source (no conversion back), lo = v_cvt_pkrtz of the two remainders
    // (|x - hi - lo| <= 2^-20 |x|), and the running max |x| for the range check.
    // rmax[i]: running max |x| of the thread's i-th vector slot -- four columns of ONE position (row t of the tile) -- over every
    // step of the block: the range check above 65504 and, round 4, the check at the LOW end (below)
    float rmax[XV], wmax = 0.f;          // (wmax: the f32 weights' own range check)
#pragma unroll
    for (int i = 0; i < XV; ++i) rmax[i] = 0.f;
    auto xstore = [&](auto par, int buf, int lo_i, int hi_i) {
        constexpr int B_ = decltype(par)::value;
        _Float16* hi = xh + (size_t)(buf * 2) * TP * PF_HP;
        _Float16* lo = hi + (size_t)TP * PF_HP;
#pragma unroll
        for (int i = 0; i < XV; ++i) {
            if (i < lo_i || i >= hi_i) continue;
            typedef __fp16 pf_h2 __attribute__((ext_vector_type(2)));
            union { pf_h2 h2[2]; unsigned u[2]; pf_v4h v; } H, L;
#pragma unroll
            for (int e = 0; e < 4; e += 2) {
                const float x0 = xr[B_][i][e], x1 = xr[B_][i][e + 1];
                H.h2[e / 2] = __builtin_amdgcn_cvt_pkrtz(x0, x1);
                float d0, d1;
                asm("v_fma_mix_f32 %0, %2, -1.0, %3 op_sel_hi:[1,0,0]\n\t"
                    "v_fma_mix_f32 %1, %2, -1.0, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                    : "=&v"(d0), "=&v"(d1) : "v"(H.u[e / 2]), "v"(x0), "v"(x1));
                L.h2[e / 2] = __builtin_amdgcn_cvt_pkrtz(d0, d1);
                rmax[i] = fmaxf(fmaxf(rmax[i], fabsf(x0)), fabsf(x1));
            }
            *reinterpret_cast<pf_v4h*>(hi + xo[i]) = H.v;
            *reinterpret_cast<pf_v4h*>(lo + xo[i]) = L.v;
        }
    };
    // A step = four pieces (32-column chunk j, hi or lo): A = the stage's 16 bytes as they came from memory, B = 8 halfs of
    // each position's row.  The B fragments of piece p+1 are requested before the instructions of piece p are issued, and a
    // quarter of the NEXT step's activations is split and published between them (VALU and LDS writes issue in the shadow
    // of the matrix instructions; the two LDS buffers alternate).
    auto bread = [&](pf_v8h (&B)[NG], int buf, int p) {
        const _Float16* src = xh + (size_t)(buf * 2 + (p & 1)) * TP * PF_HP + li * PF_HP + ((((p >> 1) * 4 + kg) ^ ((li >> 1) & 7)) << 3);
#pragma unroll
        for (int g = 0; g < NG; ++g) B[g] = *reinterpret_cast<const pf_v8h*>(src + g * 16 * PF_HP);
    };
    auto step = [&](int S, auto cur, auto nxt, auto par) {
        constexpr int CUR = decltype(cur)::value, PAR = decltype(par)::value;      // PAR = S & 1
        const int buf = S & 1;
        if constexpr (XR2) {
            xload(std::integral_constant<int, PAR>());          // X(S+2) -> xr[S&1] (published during step S-1: free)
        } else {
            xstore(std::integral_constant<int, 0>(), buf ^ 1, 0, XV);
            xload(std::integral_constant<int, 0>());
        }
        pf_v8h B0[NG], B1[NG];
        bread(B0, buf, 0);
        if constexpr (Q4 || F32W) {
            // f32 weights: w = wh + wl, two f16 pieces (|w - wh - wl| <= 2^-20 |w|, as for the activations; a weight of
            // magnitude >= 65504 raises the same flag), three instructions per chunk: wh.xhi + wl.xhi + wh.xlo.
            // q4_0: the weight (n - 8) d is an exact f32 (4 x 11 significand bits) and splits EXACTLY into two f16 pieces,
            // wh + wl (15 bits: 11 + the rest; the tail of a weight below 2^-9 that falls under the f16 subnormal step,
            // 2^-24, is dropped) -- so a chunk is three instructions into the SAME accumulators, wh.xhi + wl.xhi + wh.xlo
            // (wl.xlo is 2^-22 of the term), with the reference's own weight values and no arithmetic on the accumulators
            // outside the matrix core: they can live in accumulation registers, and two row groups per wave fit.  (First
            // version, round 3: n - 8 as the A operand and the block scale applied to each block's 16x16 sum by the VALU --
            // 64 accumulators + the sums in architectural VGPRs, one row group per wave at 128 positions.)
            // Per pair of weights: two v_fma_mix_f32 (f16 n - 8 times f16 d, exact), v_cvt_pkrtz (wh), two v_fma_mix_f32
            // (w - wh), v_cvt_pkrtz (wl).
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                pf_v8h Wh[NR], Wl[NR];
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    typedef _Float16 pf_h2 __attribute__((ext_vector_type(2)));
                    typedef __fp16 pf_g2 __attribute__((ext_vector_type(2)));
                    union { unsigned u[4]; pf_h2 h[4]; } t;
                    unsigned dp = 0;
                    if constexpr (Q4) {
                        const unsigned sh = (kg >> 1) * 4;
                        dp = wd[CUR][r];
#pragma unroll
                        for (int d = 0; d < 2; ++d) {
                            const unsigned q = ((d ? wq[CUR][r * NJ + j].y : wq[CUR][r * NJ + j].x) >> sh) & 0x0F0F0F0Fu;
                            t.u[2 * d] = __builtin_amdgcn_perm(0x64646464u, q, 0x04010400u);        // halves 0x64nn = 1024 + n of bytes 0, 1
                            t.u[2 * d + 1] = __builtin_amdgcn_perm(0x64646464u, q, 0x04030402u);    // ... of bytes 2, 3
                        }
                    }
                    union { unsigned u[4]; pf_g2 g[4]; pf_v8h v; } H, L;
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        float w0, w1, r0, r1;
                        if constexpr (F32W) {
                            const float4& wv = w[CUR][2 * (r * NJ + j) + d / 2];
                            w0 = (d & 1) ? wv.z : wv.x;
                            w1 = (d & 1) ? wv.w : wv.y;
                            wmax = fmaxf(fmaxf(wmax, fabsf(w0)), fabsf(w1));
                        } else {
                        t.h[d] = t.h[d] - (pf_h2){(_Float16)1032.0f, (_Float16)1032.0f};       // n - 8
                        if (j == 0)
                            asm("v_fma_mix_f32 %0, %2, %3, 0 op_sel_hi:[1,1,0]\n\t"
                                "v_fma_mix_f32 %1, %2, %3, 0 op_sel:[1,0,0] op_sel_hi:[1,1,0]" : "=&v"(w0), "=&v"(w1) : "v"(t.u[d]), "v"(dp));
                        else
                            asm("v_fma_mix_f32 %0, %2, %3, 0 op_sel:[0,1,0] op_sel_hi:[1,1,0]\n\t"
                                "v_fma_mix_f32 %1, %2, %3, 0 op_sel:[1,1,0] op_sel_hi:[1,1,0]" : "=&v"(w0), "=&v"(w1) : "v"(t.u[d]), "v"(dp));
                        }
                        H.g[d] = __builtin_amdgcn_cvt_pkrtz(w0, w1);
                        asm("v_fma_mix_f32 %0, %2, -1.0, %3 op_sel_hi:[1,0,0]\n\t"
                            "v_fma_mix_f32 %1, %2, -1.0, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=&v"(r0), "=&v"(r1) : "v"(H.u[d]), "v"(w0), "v"(w1));
                        L.g[d] = __builtin_amdgcn_cvt_pkrtz(r0, r1);
                    }
                    Wh[r] = H.v;
                    Wl[r] = L.v;
                }
#pragma unroll
                for (int pp = 0; pp < 2; ++pp) {
                    const int p = 2 * j + pp;
                    pf_v8h (&Bc)[NG] = (p & 1) ? B1 : B0;
                    pf_v8h (&Bn)[NG] = (p & 1) ? B0 : B1;
                    if (p < 3) bread(Bn, buf, p + 1);
                    if constexpr (XR2) xstore(std::integral_constant<int, PAR ^ 1>(), buf ^ 1, p * XV / 4, (p + 1) * XV / 4);     // X(S+1) from xr[(S+1)&1]
#pragma unroll
                    for (int r = 0; r < NR; ++r)
#pragma unroll
                        for (int g = 0; g < NG; ++g) acc[r][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wh[r], Bc[g], acc[r][g], 0, 0, 0);
                    if (pp == 0) {
#pragma unroll
                        for (int r = 0; r < NR; ++r)
#pragma unroll
                            for (int g = 0; g < NG; ++g) acc[r][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wl[r], Bc[g], acc[r][g], 0, 0, 0);
                    }
                }
            }
        } else {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                pf_v8h (&Bc)[NG] = (p & 1) ? B1 : B0;
                pf_v8h (&Bn)[NG] = (p & 1) ? B0 : B1;
                if (p < 3) bread(Bn, buf, p + 1);
                xstore(std::integral_constant<int, PAR ^ 1>(), buf ^ 1, p * XV / 4, (p + 1) * XV / 4);     // X(S+1) from xr[(S+1)&1]
                pf_v8h A[NR];
#pragma unroll
                for (int r = 0; r < NR; ++r) A[r] = *reinterpret_cast<const pf_v8h*>(&w[CUR][r * NJ + (p >> 1)]);
#pragma unroll
                for (int r = 0; r < NR; ++r)
#pragma unroll
                    for (int g = 0; g < NG; ++g) acc[r][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[r], Bc[g], acc[r][g], 0, 0, 0);
            }
        }
        wload(nxt);
        if (++ck == a.nk || S == nsteps - 1) {
            const int slot = (int)blockIdx.x - pf_first_block(cs, a.nk, a.U);
#pragma unroll
            for (int r = 0; r < NR; ++r)
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    *reinterpret_cast<pf_v4f*>(tb + (g * 16 + li) * (SR + PF_TPAD) + wid * (16 * NR) + r * 16 + (lane >> 4) * 4) = acc[r][g];
                    acc[r][g] = (pf_v4f){0.f, 0.f, 0.f, 0.f};
                }
            __syncthreads();
            float* dst = a.P + (size_t)slot * TP * a.rows + cs * SR;
#pragma unroll
            for (int k = 0; k < TP * SR / 4 / NT; ++k) {
                const int idx = tid + k * NT, t = idx / (SR / 4), c = idx % (SR / 4) * 4;
                const pf_v4f v = *reinterpret_cast<const pf_v4f*>(tb + t * (SR + PF_TPAD) + c);
                if (cs * SR + c < a.rows) *reinterpret_cast<pf_v4f*>(dst + (size_t)t * a.rows + c) = v;
            }
            ck = 0;
            ++cs;
        }
        __syncthreads();
        PF_STAMP(2 + S);
    };
    // prologue: weights of steps 0 .. ST-2; activations of step 0 published, of step 1 in xr[1] (step 0 requests step 2 into xr[0])
    pf_prologue<0, ST - 1>(wload);
    xload(std::integral_constant<int, 0>());
    xstore(std::integral_constant<int, 0>(), 0, 0, XV);
    xload(std::integral_constant<int, XR2 ? 1 : 0>());
    __syncthreads();
    PF_STAMP(1);
    // two ring cycles per trip: the loop's back edge costs a drain of the weight prefetch (hipcc's waitcnt pass merges the
    // loop's two entries conservatively: s_waitcnt vmcnt(0) in the first step of a trip), and this path's GEMMs have <= 12 units per block
    for (int s = 0; s < nsteps; s += 2 * ST) pf_ring<0, 2 * ST, ST>(step, s, nsteps);
    // Range checks of the two-piece split (advisor, round 3).  High end: an activation >= 65504 fits neither piece.  LOW end:
    // below 2^-3 the lo piece is an f16 subnormal and the pair's error is 2^-24 ABSOLUTE, not 2^-20 relative -- harmless for
    // the small elements of an O(1) row, but a position whose WHOLE row is small (attention output or SwiGLU output of a model
    // with tiny values there) would carry 2^-24 / |row| into every product.  The 16 lanes that stage one position's 64 columns
    // fold their running maxima (DPP row).  A workgroup sees only ITS column steps of a row (advisor, round 4: a row whose large
    // elements lie in another workgroup's steps must not count), so it only VOTES: a position whose largest element over this
    // block's steps is below 2^-7 adds one to lowcnt[t] -- and one to lowcnt[PF_TMAX + t] unless the slice is all zero.  The
    // epilogue kernel that follows every GEMM (pf_low_check) raises bit 1 of the flag for a position ALL workgroups voted
    // for, not all of them with zeros, and clears the counters.  With activations of ordinary size no atomic is issued at all.
    float amax = wmax;
#pragma unroll
    for (int i = 0; i < XV; ++i) {
        amax = fmaxf(amax, rmax[i]);
        float r = rmax[i];
        r = fmaxf(r, dpp_mov<0xB1, 0xf, false>(0.f, r));     // quad_perm [1,0,3,2]
        r = fmaxf(r, dpp_mov<0x4E, 0xf, false>(0.f, r));     // quad_perm [2,3,0,1]
        r = fmaxf(r, dpp_mov<0x141, 0xf, false>(0.f, r));    // row_half_mirror
        r = fmaxf(r, dpp_mov<0x140, 0xf, false>(0.f, r));    // row_mirror: every lane of the row holds the row's maximum
        const int t = (tid + i * NT) / (PF_KSTEP / 4);
        if (r < 0.0078125f && (lane & 15) == 0 && t < a.T) {
            atomicAdd(lowcnt + t, 1u);
            if (r > 0.f) atomicAdd(lowcnt + PF_TMAX + t, 1u);
        }
    }
    if (__any(!(amax < 65504.0f)) && lane == 0) atomicOr(flag, 1u);     // (NaN counts)
#ifdef LLMK_PF_TRACE
    if (tid == 0) { tr[17] = wall_clock64(); tr[19] = __builtin_readcyclecounter(); }
#endif
#undef PF_STAMP
}

// x[t] = token_embedding_table(:, token_t)                                              llama2.f90:520
__global__ void pf_embed_kernel(const float* __restrict__ table, const int* __restrict__ tokens0, float* __restrict__ X, int E) {
    const int t = blockIdx.y;
    const float* src = table + (size_t)tokens0[t] * E;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < E; i += gridDim.x * blockDim.x) X[(size_t)t * E + i] = src[i];
}

// xs[t] = x[t]*w ; xn[t] = sqrt(dot(x,x)/E + 1e-5): the division is applied to the finished row sums    :450-457
__global__ __launch_bounds__(256) void pf_norm_kernel(const float* __restrict__ X, const float* __restrict__ w,
                                                      float* __restrict__ Xs, float* __restrict__ xn, int E, float eps) {
    __shared__ float red[4];
    const int t = blockIdx.x, tid = threadIdx.x;
    const float* x = X + (size_t)t * E;
    float ss = 0.f;
    for (int i = tid; i < E; i += 256) {
        const float v = x[i];
        ss = fmaf(v, v, ss);
        Xs[(size_t)t * E + i] = v * w[i];
    }
    ss = wave_sum(ss);
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    __syncthreads();
    if (tid == 0) xn[t] = sqrtf(((red[0] + red[1]) + (red[2] + red[3])) / (float)E + eps);
}

// ---- batched causal attention on the matrix cores                                     llama2.f90:572-598 ------------
// One workgroup per (query head, 16 prompt positions); its 8 waves take the 16-row tiles of the KV cache round robin, each
// with its own running softmax (max m, sum l, unnormalised output O), and the workgroup merges the eight partial results
// at the end:  out = sum_w O_w e^(m_w - M) / sum_w l_w e^(m_w - M).  The token-by-token kernel (kernels.h attn_kernel) is
// launched per (head, position) and re-reads the cache for every position: 16x the L2 traffic of this one.
//
// Per tile and wave, with v_mfma_f32_16x16x4_f32 (D[i][j] += sum_k A[i][k] B[k][j]; lane l gives A[l%16][l/16] and
// B[l/16][l%16], and receives D[4*(l/16)+v][l%16] in register v):
//   S^T = K Q^T : A = K tile (i = cache row), B = Q (j = query), HS/4 steps; lane (q = l%16, g = l/16) holds head
//         dimensions 16g' .. of both (the contraction index can be any permutation as long as A and B agree), and ends up
//         with S[q][row 4g+v], v = 0..3 -- a query's 16 scores sit in 4 lanes x 4 registers: max and sum need two
//         cross-lane steps (xor 16, 32);
//   O += P V   : A = P (i = query): lane (q, g) feeds P[q][row 4g+v] in step v -- exactly the registers it holds, no
//         transpose; B = V: lane (c = l%16, g) feeds V[row 4g+v][NT*c + t] for output tile t (NT = HS/16 adjacent floats:
//         one vector load per row), and receives O[query 4g+r][NT*c + t] in register r.
constexpr int PF_ATT_WAVES = 8;

template <int HS>
__global__ __launch_bounds__(PF_ATT_WAVES * WAVE) void pf_attn_kernel(const float* __restrict__ Q, const float* __restrict__ kc,
                                                                      const float* __restrict__ vc, float* __restrict__ out,
                                                                      int KV, int kv_mul, int pos0, int T, int E) {
    constexpr int DG = HS / 4, NT = HS / 16, NW = PF_ATT_WAVES;
    extern __shared__ __attribute__((aligned(16))) char pf_smem[];
    float* so = reinterpret_cast<float*>(pf_smem);   // [NW][16][HS]
    float* sm = so + NW * 16 * HS;                    // [NW][16]
    float* sl = sm + NW * 16;                         // [NW][16]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, li = lane & 15, g = lane >> 4;
    const int h = blockIdx.x, kvh = h / kv_mul, q0 = blockIdx.y * 16;
    const int tq = q0 + li;                           // this lane's query (token index of the batch); row r is visible iff r < pos0 + tq
    const int nrows = pos0 + min(q0 + 15, T - 1);     // rows the tile's last query sees: 0 .. nrows-1
    const float scale = sqrtf((float)HS);
    float qf[DG];
    {
        const float* qp = Q + (size_t)min(tq, T - 1) * E + (size_t)h * HS + g * DG;
#pragma unroll
        for (int m = 0; m < DG; m += 4) {
            const float4 v = *reinterpret_cast<const float4*>(qp + m);
            qf[m] = v.x; qf[m + 1] = v.y; qf[m + 2] = v.z; qf[m + 3] = v.w;
        }
    }
    float m_run = -INFINITY, l_run = 0.f;
    pf_v4f O[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) O[t] = (pf_v4f){0.f, 0.f, 0.f, 0.f};
    const float* kb = kc + (size_t)kvh * HS + g * DG;
    const float* vb = vc + (size_t)kvh * HS + NT * li;
    const int ntile = (nrows + 15) >> 4;
    // K / V fragments of a tile are requested one tile ahead (unconditional loads of a clamped tile index): a tile's two
    // dozen MFMAs and its softmax take far less than an L2 round trip
    float kn[DG], vn[4][NT];
#define PF_ATT_LOAD(R0_)                                                                                  \
    do {                                                                                                  \
        const float* kp_ = kb + (size_t)min((R0_) + li, nrows - 1) * KV;                                  \
        _Pragma("unroll") for (int m = 0; m < DG; m += 4) {                                               \
            const float4 v_ = *reinterpret_cast<const float4*>(kp_ + m);                                  \
            kn[m] = v_.x; kn[m + 1] = v_.y; kn[m + 2] = v_.z; kn[m + 3] = v_.w;                           \
        }                                                                                                 \
        _Pragma("unroll") for (int v = 0; v < 4; ++v) {                                                   \
            const float* vp_ = vb + (size_t)min((R0_) + 4 * g + v, nrows - 1) * KV;                       \
            if constexpr (NT % 4 == 0) {                                                                  \
                _Pragma("unroll") for (int t = 0; t < NT; t += 4) {                                       \
                    const float4 x_ = *reinterpret_cast<const float4*>(vp_ + t);                          \
                    vn[v][t] = x_.x; vn[v][t + 1] = x_.y; vn[v][t + 2] = x_.z; vn[v][t + 3] = x_.w;       \
                }                                                                                         \
            } else if constexpr (NT == 2) {                                                               \
                const float2 x_ = *reinterpret_cast<const float2*>(vp_);                                  \
                vn[v][0] = x_.x; vn[v][1] = x_.y;                                                         \
            } else {                                                                                      \
                vn[v][0] = vp_[0];                                                                        \
            }                                                                                             \
        }                                                                                                 \
    } while (0)
    PF_ATT_LOAD(min(wid, ntile - 1) << 4);
    for (int kt = wid; kt < ntile; kt += NW) {
        const int r0 = kt << 4;
        float kf[DG], vf[4][NT];
#pragma unroll
        for (int m = 0; m < DG; ++m) kf[m] = kn[m];
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
            for (int t = 0; t < NT; ++t) vf[v][t] = vn[v][t];
        PF_ATT_LOAD(min(kt + NW, ntile - 1) << 4);
        pf_v4f acc = (pf_v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < DG; ++m) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[m], qf[m], acc, 0, 0, 0);
        float sc[4], mx = -INFINITY;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const bool vis = r0 + 4 * g + v < pos0 + tq && tq < T;
            sc[v] = vis ? acc[v] / scale : -INFINITY;
            mx = fmaxf(mx, sc[v]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const bool any = m_new != -INFINITY;                                   // false: nothing of this query seen so far
        const float alpha = !any ? 1.f : (m_run == -INFINITY ? 0.f : expf(m_run - m_new));
        float p[4], rs = 0.f;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            p[v] = (any && sc[v] != -INFINITY) ? expf(sc[v] - m_new) : 0.f;
            rs += p[v];
        }
        rs += __shfl_xor(rs, 16, 64);
        rs += __shfl_xor(rs, 32, 64);
        l_run = l_run * alpha + rs;
        m_run = m_new;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float ar = __shfl(alpha, 4 * g + r, 64);                     // O rows are queries 4g+r
#pragma unroll
            for (int t = 0; t < NT; ++t) O[t][r] *= ar;
        }
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int v = 0; v < 4; ++v) O[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(p[v], vf[v][t], O[t], 0, 0, 0);
    }
#undef PF_ATT_LOAD
    if (g == 0) {
        sm[wid * 16 + li] = m_run;
        sl[wid * 16 + li] = l_run;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int t = 0; t < NT; ++t) so[(size_t)(wid * 16 + 4 * g + r) * HS + NT * li + t] = O[t][r];
    __syncthreads();
    for (int idx = tid; idx < 16 * HS; idx += NW * WAVE) {
        const int q = idx / HS, d = idx % HS;
        if (q0 + q >= T) continue;
        float M = -INFINITY;
#pragma unroll
        for (int w = 0; w < NW; ++w) M = fmaxf(M, sm[w * 16 + q]);
        float L = 0.f, val = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const float mw = sm[w * 16 + q];
            const float e = mw == -INFINITY ? 0.f : expf(mw - M);
            L += sl[w * 16 + q] * e;
            val += so[(size_t)(w * 16 + q) * HS + d] * e;
        }
        out[(size_t)(q0 + q) * E + (size_t)h * HS + d] = val / L;
    }
}

struct PfEpiArgs {
    const float* P;      // [slots][Tp][rows]
    const float* xn;     // [T] (QKV, SWIGLU) or null
    float* out;          // QKV: Q [T][E];  RESID: X [T][rows] (+=);  SWIGLU: HB [T][H]
    float* kc;           // QKV: this layer's caches [S][KV]
    float* vc;
    const float* rope;   // [hs/2]
    int rows, Tp, T, pos0;       // pos0: 1-based position of token 0
    int U, nk, sh;               // the GEMM's units per block / steps per strip (PfGemmArgs): the strip r >> sh has pf_nslots partials
    int E, KV, hs, H;
    unsigned* lowcnt;            // [2][PF_TMAX] votes of the GEMM's workgroups (pf_gemm_h_kernel), null on the f32-instruction path
    unsigned* flag;
    int gemm_blocks;             // workgroups of the GEMM this epilogue follows
};
// the first workgroup of the epilogue that follows a GEMM on the f16 instruction: a position every workgroup found small in its
// column steps -- and not all of them zero -- is small as a whole (bit 1 of the flag: llmk_prefill redoes this call on the f32
// instruction); the votes are cleared for the next GEMM of this lane
__device__ __forceinline__ void pf_low_check(const PfEpiArgs& a, bool first_block) {
    if (!a.lowcnt || !first_block) return;
    for (int t = threadIdx.x; t < a.T; t += blockDim.x) {
        const unsigned n = a.lowcnt[t], nz = a.lowcnt[PF_TMAX + t];
        if (n | nz) { a.lowcnt[t] = 0u; a.lowcnt[PF_TMAX + t] = 0u; }
        if (n == (unsigned)a.gemm_blocks && nz > 0u) atomicOr(a.flag, 2u);
    }
}

// partials of (t, r) added in slot order; four loads in flight at a time (the trip count is a run-time value: a plain
// loop would wait for each load before asking for the next)
__device__ __forceinline__ float pf_sum(const PfEpiArgs& a, int t, int r) {
    const int n = pf_nslots(r >> a.sh, a.nk, a.U);
    const size_t pitch = (size_t)a.Tp * a.rows;
    const float* p = a.P + (size_t)t * a.rows + r;
    float s = 0.f;
    int ks = 0;
    for (; ks + 4 <= n; ks += 4, p += 4 * pitch) {
        const float v0 = p[0], v1 = p[pitch], v2 = p[2 * pitch], v3 = p[3 * pitch];
        s += v0; s += v1; s += v2; s += v3;
    }
    for (; ks < n; ++ks, p += pitch) s += *p;
    return s;
}

// the same for four consecutive rows (r % 4 == 0: one strip, one slot count; 16-byte loads): each row's partials still add in
// slot order
__device__ __forceinline__ float4 pf_sum4(const PfEpiArgs& a, int t, int r) {
    const int n = pf_nslots(r >> a.sh, a.nk, a.U);
    const size_t pitch = (size_t)a.Tp * a.rows;
    const float* p = a.P + (size_t)t * a.rows + r;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    auto add = [&](const float4& v) { s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; };
    int ks = 0;
    for (; ks + 4 <= n; ks += 4, p += 4 * pitch) {
        const float4 v0 = *reinterpret_cast<const float4*>(p), v1 = *reinterpret_cast<const float4*>(p + pitch),
                     v2 = *reinterpret_cast<const float4*>(p + 2 * pitch), v3 = *reinterpret_cast<const float4*>(p + 3 * pitch);
        add(v0); add(v1); add(v2); add(v3);
    }
    for (; ks < n; ++ks, p += pitch) add(*reinterpret_cast<const float4*>(p));
    return s;
}

// one thread per (token, RoPE pair / V pair)                                          llama2.f90:543-565
__device__ __forceinline__ void pf_epi_qkv_pair(const PfEpiArgs& a, int t, int r0, float a0, float a1) {
    const int pos = a.pos0 + t;
    if (r0 < a.E + a.KV) {
        const int i0 = (r0 < a.E) ? r0 : r0 - a.E;
        const float rval = (float)pos * a.rope[(i0 % a.hs) >> 1];
        const float fcr = cosf(rval), fci = sinf(rval);
        float* dst = (r0 < a.E) ? a.out + (size_t)t * a.E + i0 : a.kc + (size_t)(pos - 1) * a.KV + i0;
        dst[0] = a0 * fcr - a1 * fci;
        dst[1] = a0 * fci + a1 * fcr;
    } else {
        float* dst = a.vc + (size_t)(pos - 1) * a.KV + (r0 - a.E - a.KV);
        dst[0] = a0;
        dst[1] = a1;
    }
}
__global__ void pf_epi_qkv_kernel(PfEpiArgs a) {
    // a thread takes four rows = two pairs (16-byte loads of the partial tiles); E, KV and the head size are multiples of 4,
    // so both pairs lie in the same part (q, k or v)
    const int p4 = blockIdx.x * blockDim.x + threadIdx.x, t = blockIdx.y;
    pf_low_check(a, blockIdx.x == 0 && blockIdx.y == 0);
    if (p4 >= a.rows / 4) return;
    const float4 s4 = pf_sum4(a, t, 4 * p4);
    pf_epi_qkv_pair(a, t, 4 * p4, s4.x / a.xn[t], s4.y / a.xn[t]);
    pf_epi_qkv_pair(a, t, 4 * p4 + 2, s4.z / a.xn[t], s4.w / a.xn[t]);
}
// x[t] += sum (:603-605, :618-620), then the NEXT rmsnorm's products from the finished row (:450-457): xs[t] = x[t]*w and
// xn[t] = sqrt(dot(x,x)/E + eps).  One workgroup per position; w == nullptr (after the last layer): residual only.
__global__ __launch_bounds__(1024) void pf_epi_resid_norm_kernel(PfEpiArgs a, const float* __restrict__ w, float* __restrict__ Xs,
                                                                 float* __restrict__ xn, float eps) {
    __shared__ float red[16];
    const int t = blockIdx.x, tid = threadIdx.x;
    pf_low_check(a, blockIdx.x == 0);
    float ss = 0.f;
    for (int r = 4 * tid; r < a.rows; r += 4096) {       // rows % 64 == 0 on this path (llmk_prefill)
        const float4 x = *reinterpret_cast<const float4*>(a.out + (size_t)t * a.rows + r), d = pf_sum4(a, t, r);
        const float4 v = make_float4(x.x + d.x, x.y + d.y, x.z + d.z, x.w + d.w);
        *reinterpret_cast<float4*>(a.out + (size_t)t * a.rows + r) = v;
        if (w) {
            ss = fmaf(v.x, v.x, ss); ss = fmaf(v.y, v.y, ss); ss = fmaf(v.z, v.z, ss); ss = fmaf(v.w, v.w, ss);
            const float4 g = *reinterpret_cast<const float4*>(w + r);
            *reinterpret_cast<float4*>(Xs + (size_t)t * a.rows + r) = make_float4(v.x * g.x, v.y * g.y, v.z * g.z, v.w * g.w);
        }
    }
    if (!w) return;
    ss = wave_sum(ss);
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    __syncthreads();
    if (tid == 0) {
        float tot = 0.f;
        for (int i = 0; i < 16; ++i) tot += red[i];
        xn[t] = sqrtf(tot / (float)a.rows + eps);
    }
}
// hb = silu(gate) * up                                                                  :613-616
__global__ void pf_epi_swiglu_kernel(PfEpiArgs a) {
    const int g = 4 * (blockIdx.x * blockDim.x + threadIdx.x), t = blockIdx.y;      // four hidden units per thread (H % 64 == 0)
    pf_low_check(a, blockIdx.x == 0 && blockIdx.y == 0);
    if (g >= a.H) return;
    const float xn = a.xn[t];
    const float4 gs = pf_sum4(a, t, g), us = pf_sum4(a, t, g + a.H);
    auto unit = [&](float gsum, float usum) {
        const float gate = gsum / xn, up = usum / xn;
        return gate * (1.0f / (1.0f + expf(-gate))) * up;
    };
    *reinterpret_cast<float4*>(a.out + (size_t)t * a.H + g) = make_float4(unit(gs.x, us.x), unit(gs.y, us.y), unit(gs.z, us.z), unit(gs.w, us.w));
}

}  // namespace llmk
