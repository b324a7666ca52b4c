// q6_K classifier rows on the device (round 6).
//
// What this is for: stock llama.cpp q4_0 files keep `output.weight` in q6_K.  The reference stops on any tensor type >= 2
// (/root/reference/read_ggml.f90:633-635, :682-684); until round 5 this repo's loader dequantised the tensor on the host to f32
// and the context then left the persistent kernel (a 524 MB f32 classifier at 7B instead of 107 MB, five launches per layer).
// Now the raw super-blocks are uploaded and dotted here: the classifier GEMV of llama2.f90:634-636 on q6_K rows, in the persistent
// q4_0 kernels' classifier phase (token_kernel.h tk_q6_phase) and as a kernel of its own for every other path (gemv_q6k_kernel).
//
// Format (ggml's public block_q6_K; third-party knowledge, not citable in /root/reference -- SURVEY.md section 8c): a super-block
// of 256 weights = ql[128] (low 4 bits) | qh[64] (high 2 bits) | int8 scales[16] (one per 16 weights) | f16 d, 210 bytes;
//     weight(128 n + 32 k + l) = d * scales[8 n + 2 k + l / 16] * (q - 32),   n = 0..1, k = 0..3, l = 0..31,
//     q = nibble k of ql (k < 2: low nibble of ql[64 n + 32 (k & 1) + l], k >= 2: high nibble) | (qh[32 n + l] >> 2 k & 3) << 4.
// d (11 significant bits) x scale (7) x (q - 32) (6) is exact in f32 in any order: the weight IS the f32 value the host-side
// dequantisation (host/gguf_loader.f90 q6k_weight, tools/gguf.py dequantize_q6_K) hands the oracle.
//
// Device row (q6k_repack_kernel; a permutation of the file's 16-bit words, so the upload's word-sum verification still holds).
// A QUAD q = 4 sb + 2 n + h is what one lane dots: super-block sb, half n, l = 16 h .. 16 h + 15 -- 64 weights, the elements
// 256 sb + 128 n + 32 k + 16 h + i (k = 0..3, i = 0..15).  Q = K / 64 quads per row, nsb = K / 256 super-blocks:
//     [0, 16 Q)         A[q] = ql[64 n + 16 h ..+16)         low nibbles: k = 0, high nibbles: k = 2
//     [16 Q, 32 Q)      B[q] = ql[64 n + 32 + 16 h ..+16)    low nibbles: k = 1, high nibbles: k = 3
//     [32 Q, 48 Q)      C[q] = qh[32 n + 16 h ..+16)         bits 2 k, 2 k + 1 of byte i: the high bits of (k, i)
//     [48 Q, 48 Q + 16 nsb)   the 16 scales of each super-block as in the file; quad (n, h) uses bytes 8 n + h + 2 k
//     [.., + 2 nsb)     d per super-block
// 210 K / 256 bytes, rows q6k_row_stride(K) apart (rounded up to 16).  A wave reads one row with three coalesced 16-byte loads,
// one 8-byte and one 2-byte load per lane (K = 4096: 64 quads = 64 lanes).
//
// The dot: (q as a byte) & 0x00ff00ff leaves two 16-bit halves that ARE the f16 subnormals q 2^-24 (q < 64), and v_fma_mix_f32
// takes an f16 source half with an f32 multiplicand and accumulator (kernels.h q4_dword_dot: the same trick on nibbles): no
// conversion instructions.  Per quad: sum_k sc_k (2^24 sum_i mix(q_ki) x_ki - 32 sum_i x_ki), times d.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace llmk {

constexpr int Q6K_BLOCK_BYTES = 210, Q6K_WEIGHTS = 256;
__host__ __device__ constexpr size_t q6k_row_stride(int K) { return ((size_t)K / Q6K_WEIGHTS * Q6K_BLOCK_BYTES + 15) / 16 * 16; }

struct Q6Quad {
    uint4 a, b, c;          // ql low / high halves and qh of the quad
    uint2 s;                // the 8 scales of the quad's (super-block, half)
    unsigned short d;       // f16 bits of the super-block's d
};
// the five loads of quad qd (clamped by the caller) of the row at `row`
__device__ __forceinline__ void q6k_load(Q6Quad& w, const char* __restrict__ row, int qd, int Q) {
    const uint4* p = reinterpret_cast<const uint4*>(row);
    w.a = ldg_nt(p + qd);
    w.b = ldg_nt(p + Q + qd);
    w.c = ldg_nt(p + 2 * Q + qd);
    const char* sp = row + (size_t)48 * Q;
    w.s = *reinterpret_cast<const uint2*>(sp + (qd >> 1) * 8);
    w.d = *reinterpret_cast<const unsigned short*>(sp + 4 * Q + (qd >> 2) * 2);       // (16 nsb = 4 Q bytes of scales)
}
// first element of quad qd in the row: x_k[i] = x[q6k_base(qd) + 32 k + i]
__device__ __forceinline__ int q6k_base(int qd) { return (qd >> 2) * 256 + ((qd >> 1) & 1) * 128 + (qd & 1) * 16; }

__device__ __forceinline__ float q6k_mix_lo(unsigned h, float x, float acc) {
    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(acc) : "v"(h), "v"(x));
    return acc;
}
__device__ __forceinline__ float q6k_mix_hi(unsigned h, float x, float acc) {
    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(acc) : "v"(h), "v"(x));
    return acc;
}
// one dword of 6-bit values (bytes i .. i + 3 of group k) against x[i .. i + 3]
__device__ __forceinline__ float q6k_dword(unsigned q, const float* x, float acc) {
    const unsigned e = q & 0x00ff00ffu, o = (q >> 8) & 0x00ff00ffu;      // bytes 0, 2 | bytes 1, 3 as f16 subnormals
    acc = q6k_mix_lo(e, x[0], acc);
    acc = q6k_mix_lo(o, x[1], acc);
    acc = q6k_mix_hi(e, x[2], acc);
    acc = q6k_mix_hi(o, x[3], acc);
    return acc;
}
// x: the quad's 64 activations in (k, i) order; sx32[k] = 32 * sum_i x[16 k + i]; h = qd & 1.  Returns the quad's share of W[r] . x
__device__ __forceinline__ float q6k_quad_dot(const Q6Quad& w, const float (&x)[64], const float (&sx32)[4], int h) {
    const unsigned A[4] = {w.a.x, w.a.y, w.a.z, w.a.w}, B[4] = {w.b.x, w.b.y, w.b.z, w.b.w}, C[4] = {w.c.x, w.c.y, w.c.z, w.c.w};
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned q0 = (A[j] & 0x0f0f0f0fu) | ((C[j] << 4) & 0x30303030u);
        const unsigned q1 = (B[j] & 0x0f0f0f0fu) | ((C[j] << 2) & 0x30303030u);
        const unsigned q2 = ((A[j] >> 4) & 0x0f0f0f0fu) | (C[j] & 0x30303030u);
        const unsigned q3 = ((B[j] >> 4) & 0x0f0f0f0fu) | ((C[j] >> 2) & 0x30303030u);
        acc[0] = q6k_dword(q0, x + 4 * j, acc[0]);
        acc[1] = q6k_dword(q1, x + 16 + 4 * j, acc[1]);
        acc[2] = q6k_dword(q2, x + 32 + 4 * j, acc[2]);
        acc[3] = q6k_dword(q3, x + 48 + 4 * j, acc[3]);
    }
    // scales: bytes h, h + 2 of the low word (k = 0, 1), of the high word (k = 2, 3); int8
    const unsigned s0 = h ? (w.s.x >> 8) : w.s.x, s1 = h ? (w.s.y >> 8) : w.s.y;
    const float sc0 = (float)(int)(signed char)(s0 & 0xffu), sc1 = (float)(int)(signed char)((s0 >> 16) & 0xffu);
    const float sc2 = (float)(int)(signed char)(s1 & 0xffu), sc3 = (float)(int)(signed char)((s1 >> 16) & 0xffu);
    float t = sc0 * fmaf(acc[0], 16777216.0f, -sx32[0]);
    t = fmaf(sc1, fmaf(acc[1], 16777216.0f, -sx32[1]), t);
    t = fmaf(sc2, fmaf(acc[2], 16777216.0f, -sx32[2]), t);
    t = fmaf(sc3, fmaf(acc[3], 16777216.0f, -sx32[3]), t);
    return __half2float(__ushort_as_half(w.d)) * t;
}
// the quad's activations out of a natural-order f32 vector in LDS (or zeros for a lane without a quad), and 32 x the group sums
__device__ __forceinline__ void q6k_load_x(float (&x)[64], float (&sx32)[4], const float* xs, int qd, bool live) {
    const float4* p = reinterpret_cast<const float4*>(xs + q6k_base(qd));
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float4 v = p[8 * k + i];
            if (!live) v = make_float4(0.f, 0.f, 0.f, 0.f);
            x[16 * k + 4 * i + 0] = v.x; x[16 * k + 4 * i + 1] = v.y; x[16 * k + 4 * i + 2] = v.z; x[16 * k + 4 * i + 3] = v.w;
            s += (v.x + v.y) + (v.z + v.w);
        }
        sx32[k] = 32.0f * s;
    }
}

// ------------------------------------------------------------------------------------------------
// The classifier GEMV on q6_K rows as a kernel of its own (multi-kernel path, tensor-parallel ranks, the last position of a
// prefill): y[r] = W[r] . rmsnorm(x)      llama2.f90:627-636.  One block = 4 waves x Q6K_RPW rows each, all of a wave's rows
// requested before x is staged (weights do not depend on activations); x * gains in LDS in natural order, the division by
// sqrt(mean(x^2) + eps) applied once per row sum (as gemv_q4_kernel does).
// ------------------------------------------------------------------------------------------------
constexpr int Q6K_RPW = 4;
template <bool NORM>
__global__ __launch_bounds__(GEMV_THREADS) void gemv_q6k_kernel(GemvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* red = reinterpret_cast<float*>(smem_raw);          // [4]
    float* xs = reinterpret_cast<float*>(smem_raw + 16);      // x (* gains), K floats
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int K = a.K, nx4 = K >> 2, Q = K >> 6;
    const int r0 = (blockIdx.x * GEMV_WAVES + wid) * Q6K_RPW;
    const char* Wb = reinterpret_cast<const char*>(a.W);
    const size_t RS = (size_t)a.row_stride;
    Q6Quad w[Q6K_RPW];
    const int qd0 = min(lane, Q - 1);
#pragma unroll
    for (int i = 0; i < Q6K_RPW; ++i) q6k_load(w[i], Wb + (size_t)min(r0 + i, a.rows - 1) * RS, qd0, Q);
    float xn = 1.f;
    {
        const float4* xg = reinterpret_cast<const float4*>(a.x);
        const float4* wg = reinterpret_cast<const float4*>(a.norm_w);
        float ss = 0.f;
        for (int i0 = tid; i0 < nx4; i0 += GEMV_THREADS * 4) {
            float4 v[4], nw[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = min(i0 + u * GEMV_THREADS, nx4 - 1);
                v[u] = xg[i];
                if (NORM) nw[u] = wg[i];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * GEMV_THREADS;
                if (i < nx4) {
                    if (NORM) {
                        ss = dot4(v[u], v[u], ss);
                        v[u].x *= nw[u].x; v[u].y *= nw[u].y; v[u].z *= nw[u].z; v[u].w *= nw[u].w;
                    }
                    reinterpret_cast<float4*>(xs)[i] = v[u];
                }
            }
        }
        if (NORM) {
            ss = wave_sum(ss);
            if (lane == 0) red[wid] = ss;
        }
        __syncthreads();
        if (NORM) xn = sqrtf(((red[0] + red[1]) + (red[2] + red[3])) / (float)K + a.eps);
    }
    float acc[Q6K_RPW];
#pragma unroll
    for (int i = 0; i < Q6K_RPW; ++i) acc[i] = 0.f;
    for (int q0 = 0; q0 < Q; q0 += WAVE) {            // K <= 4096: one trip
        const int qd = q0 + lane;
        const bool live = qd < Q;
        const int qc = live ? qd : Q - 1;
        if (q0 > 0) {
#pragma unroll
            for (int i = 0; i < Q6K_RPW; ++i) q6k_load(w[i], Wb + (size_t)min(r0 + i, a.rows - 1) * RS, qc, Q);
        }
        float x[64], sx32[4];
        q6k_load_x(x, sx32, xs, qc, live);
#pragma unroll
        for (int i = 0; i < Q6K_RPW; ++i) acc[i] += q6k_quad_dot(w[i], x, sx32, qc & 1);
    }
#pragma unroll
    for (int i = 0; i < Q6K_RPW; ++i) acc[i] = wave_sum(acc[i]);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < Q6K_RPW; ++i)
            if (r0 + i < a.rows) a.y[r0 + i] = NORM ? acc[i] / xn : acc[i];
    }
}

// upload: ggml super-blocks (210 bytes, 2-byte aligned) of `nsb` per row -> device rows (see the top of this file).  One thread
// per (row, super-block); every move is a whole 16-bit word.
__global__ void q6k_repack_kernel(const uint8_t* __restrict__ src, char* __restrict__ dst, size_t nblocks, int nsb, size_t row_stride) {
    for (size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x; b < nblocks; b += (size_t)gridDim.x * blockDim.x) {
        const uint16_t* p = reinterpret_cast<const uint16_t*>(src + b * Q6K_BLOCK_BYTES);
        const size_t row = b / nsb;
        const int sb = (int)(b - row * nsb), Q = 4 * nsb;
        uint16_t* rp = reinterpret_cast<uint16_t*>(dst + row * row_stride);
        for (int n = 0; n < 2; ++n)
            for (int h = 0; h < 2; ++h) {
                const int q = 4 * sb + 2 * n + h;
                for (int i = 0; i < 8; ++i) {
                    rp[(size_t)q * 8 + i] = p[(64 * n + 16 * h) / 2 + i];                         // A: ql[64 n + 16 h ..]
                    rp[(size_t)(Q + q) * 8 + i] = p[(64 * n + 32 + 16 * h) / 2 + i];              // B: ql[64 n + 32 + 16 h ..]
                    rp[(size_t)(2 * Q + q) * 8 + i] = p[64 + (32 * n + 16 * h) / 2 + i];          // C: qh[32 n + 16 h ..]
                }
            }
        for (int i = 0; i < 8; ++i) rp[(size_t)24 * Q + sb * 8 + i] = p[96 + i];                  // scales[16]
        rp[(size_t)24 * Q + 8 * nsb + sb] = p[104];                                               // d
    }
}

}  // namespace llmk
