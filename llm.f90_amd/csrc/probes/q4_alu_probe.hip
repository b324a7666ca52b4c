// Hardware probe (not product code): what does the q4_0 nibble -> f32 dot cost per dword (8 weights) on gfx950?
// The 7B persistent kernel spends ~1.8 us per 8 KB tile between its phase barriers (tests/host_tools/tk_trace.py) --
// three times the weight stream's share -- so the ALU recipe matters.  Variants, all on register-resident data:
//   0  v_and x2 + 8 v_cvt_f32_ubyteN + 8 v_fmac_f32            (token_kernel.h tk_q4_dword today)
//   1  v_and x2 + 4 v_cvt_pk_f32_fp8 + 8 v_fmac_f32            (byte 0x0n read as OCP e4m3 is exactly n * 2^-9)
//   2  v_and x2 + 4 v_cvt_pk_f32_fp8 + 4 v_pk_fma_f32
//   3  3 v_lshrrev + 4 v_and_or (0x4300 | n = bf16 128+n) + 1 v_mfma_f32_16x16x32_bf16
//   4  8 v_fmac_f32 only      5  8 v_cvt_f32_ubyte only      6  4 v_cvt_pk_f32_fp8 only
//   7  the MFMA recipe as it would run in the kernel: per 16-byte load (4 dwords) 3 shifts + 4 v_and_or (0x6400 | n =
//      f16 1024+n) + 4 v_pk_add_f16 (-1032 -> n-8, exact) per dword, 4 chained v_mfma_f32_16x16x32_f16 into one of 8
//      independent accumulators, then 4 v_fma_mix_f32 (block scales)
//   8  integer recipe: x held as six signed 4-bit digits per element (block fixed point), weights as signed nibbles
//      (n ^ 8 = n - 8 in two's complement): per dword 1 v_xor + 6 v_dot8c_i32_i4, per block (4 dwords) 4 v_lshl_add +
//      2 v_cvt_f32_i32 + 3 f32 ops.  No per-weight conversion at all.
//   9  6 v_dot8c_i32_i4 only
//  10  int8 matrix-core recipe: per 16-byte block load (4 dwords = 32 weights of one row) 4 v_and + 4 (v_lshrrev + v_and) unpack
//      the nibbles to bytes, 4 chained v_mfma_i32_16x16x32_i8 against a block-diagonal B (16 columns = 4 blocks x 3 int8 digits
//      of x + 1 spare: every block lands in its own output columns, so the per-(row, block) scales stay separable), then
//      4 v_cvt_f32_i32 + 4 v_fma (scales).  0.6 VALU operations per weight instead of 2.25.
//  11  the same idea on v_mfma_i32_4x4x4_16B_i8 (16 independent 4x4x4 products per instruction: MFMA-block = q4_0 block, A = 3
//      int8 digits of x (+1 spare row), B = 4 weight ROWS x 4 nibbles) -- fits the token kernel's 4-row tiles: lane (blk, row)
//      loads its row's 16-byte block, 8 MFMAs per load, then 3 v_cvt_f32_i32 + 2 v_fma (digits) + 2 v_fma (scale, 8 sum x)
// Prints cycles (s_memtime) per dword per wave for 1 and 2 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 q4_alu_probe.hip -o q4_alu_probe && ./q4_alu_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef short bf8 __attribute__((ext_vector_type(8)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

constexpr int ITERS = 2000;

template <int V>
__global__ __launch_bounds__(512) void probe(const unsigned* __restrict__ qin, const float* __restrict__ xin, float* out,
                                             unsigned long long* cyc) {
    const int tid = threadIdx.x;
    unsigned q[4];
    float x[8];
    for (int i = 0; i < 4; ++i) q[i] = qin[tid * 4 + i];
    for (int i = 0; i < 8; ++i) x[i] = xin[tid * 8 + i];
    float lo = 0.f, hi = 0.f;
    f2 plo = {0.f, 0.f}, phi = {0.f, 0.f};
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    bf8 bx;
    for (int i = 0; i < 8; ++i) bx[i] = (short)(__float_as_uint(x[i]) >> 16);
    h8 hx;
    for (int i = 0; i < 8; ++i) hx[i] = (_Float16)x[i];
    f4 accs[8];
    for (int i = 0; i < 8; ++i) accs[i] = (f4){0.f, 0.f, 0.f, 0.f};
    f4 outv = {0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    if constexpr (V == 8 || V == 9) {
        int xd[4][6];
        for (int i = 0; i < 4; ++i) for (int p = 0; p < 6; ++p) xd[i][p] = (int)q[(i + p) & 3] * (p + 3) + i;
        float facc = 0.f;
#pragma unroll 1
        for (int it = 0; it < ITERS; ++it) {
            int I[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int dw = 0; dw < 4; ++dw) {
                const int qq = (int)((q[dw] + it) ^ (V == 8 ? 0x88888888u : 0u));
#pragma unroll
                for (int p = 0; p < 6; ++p) I[p] = __builtin_amdgcn_sdot8(qq, xd[dw][p], I[p], false);
            }
            if constexpr (V == 8) {
                const int lo_ = I[0] + (I[1] << 4) + (I[2] << 8), hi_ = I[3] + (I[4] << 4) + (I[5] << 8);
                const float f = fmaf((float)hi_, 4096.0f, (float)lo_);
                facc = fmaf(f, x[0] * x[1], facc);
            } else {
                facc += (float)(I[0] ^ I[1] ^ I[2] ^ I[3] ^ I[4] ^ I[5]);
            }
        }
        acc[0] = facc;
    } else if constexpr (V == 10) {
        typedef int i4v __attribute__((ext_vector_type(4)));
        long bx[4];
        for (int m = 0; m < 4; ++m) bx[m] = ((long)__float_as_uint(x[2 * m]) << 32) | (long)__float_as_uint(x[2 * m + 1]);
#pragma unroll 1
        for (int it = 0; it < ITERS / 2; ++it) {
#pragma unroll
            for (int ld = 0; ld < 8; ++ld) {
                i4v d = {0, 0, 0, 0};
                const unsigned M = 0x0f0f0f0fu;
                const unsigned a0 = q[0] + it + ld, a1 = q[1] + it + ld, a2 = q[2] + it + ld, a3 = q[3] + it + ld;
                const long A0 = ((long)(a1 & M) << 32) | (long)(a0 & M), A1 = ((long)(a3 & M) << 32) | (long)(a2 & M);
                const long A2 = ((long)((a1 >> 4) & M) << 32) | (long)((a0 >> 4) & M), A3 = ((long)((a3 >> 4) & M) << 32) | (long)((a2 >> 4) & M);
                d = __builtin_amdgcn_mfma_i32_16x16x32_i8(A0, bx[0], d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_i32_16x16x32_i8(A1, bx[1], d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_i32_16x16x32_i8(A2, bx[2], d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_i32_16x16x32_i8(A3, bx[3], d, 0, 0, 0);
                const h2 s01 = __builtin_bit_cast(h2, q[0]), s23 = __builtin_bit_cast(h2, q[1]);
                outv[0] = fmaf((float)s01[0], (float)d[0], outv[0]); outv[1] = fmaf((float)s01[1], (float)d[1], outv[1]);
                outv[2] = fmaf((float)s23[0], (float)d[2], outv[2]); outv[3] = fmaf((float)s23[1], (float)d[3], outv[3]);
            }
        }
        acc = outv;
    } else if constexpr (V == 11) {
        typedef int i4v __attribute__((ext_vector_type(4)));
        int xd[8];
        for (int m = 0; m < 8; ++m) xd[m] = (int)__float_as_uint(x[m]);
        float racc = 0.f;
#pragma unroll 1
        for (int it = 0; it < ITERS / 2; ++it) {
#pragma unroll
            for (int ld = 0; ld < 8; ++ld) {
                i4v d = {0, 0, 0, 0};
                const unsigned M = 0x0f0f0f0fu;
                const unsigned a0 = q[0] + it + ld, a1 = q[1] + it + ld, a2 = q[2] + it + ld, a3 = q[3] + it + ld;
                d = __builtin_amdgcn_mfma_i32_4x4x4i8(xd[0], (int)(a0 & M), d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_i32_4x4x4i8(xd[1], (int)(a1 & M), d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_i32_4x4x4i8(xd[2], (int)(a2 & M), d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_i32_4x4x4i8(xd[3], (int)(a3 & M), d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_i32_4x4x4i8(xd[4], (int)((a0 >> 4) & M), d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_i32_4x4x4i8(xd[5], (int)((a1 >> 4) & M), d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_i32_4x4x4i8(xd[6], (int)((a2 >> 4) & M), d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_i32_4x4x4i8(xd[7], (int)((a3 >> 4) & M), d, 0, 0, 0);
                const float v = fmaf((float)d[2], 65536.f, fmaf((float)d[1], 256.f, (float)d[0]));
                const h2 s01 = __builtin_bit_cast(h2, q[0]);
                racc = fmaf((float)s01[0], fmaf(v, x[0], -x[1]), racc);
            }
        }
        acc[0] = racc;
    } else if constexpr (V == 7) {
#pragma unroll 1
        for (int it = 0; it < ITERS / 2; ++it) {
#pragma unroll
            for (int ld = 0; ld < 8; ++ld) {           // 8 lane-loads of a tile: 8 independent accumulators
                f4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int dw = 0; dw < 4; ++dw) {
                    const unsigned qq = q[dw] + it + ld;
                    unsigned w[4];
                    w[0] = (qq & 0x000f000fu) | 0x64006400u;
                    w[1] = ((qq >> 4) & 0x000f000fu) | 0x64006400u;
                    w[2] = ((qq >> 8) & 0x000f000fu) | 0x64006400u;
                    w[3] = ((qq >> 12) & 0x000f000fu) | 0x64006400u;
                    h8 a;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        h2 p = __builtin_bit_cast(h2, w[k]);
                        p = p - (h2){(_Float16)1032.f, (_Float16)1032.f};
                        a[2 * k] = p[0]; a[2 * k + 1] = p[1];
                    }
                    d = __builtin_amdgcn_mfma_f32_16x16x32_f16(hx, a, d, 0, 0, 0);
                }
                const h2 s01 = __builtin_bit_cast(h2, q[0]), s23 = __builtin_bit_cast(h2, q[1]);
                outv[0] = fmaf((float)s01[0], d[0], outv[0]); outv[1] = fmaf((float)s01[1], d[1], outv[1]);
                outv[2] = fmaf((float)s23[0], d[2], outv[2]); outv[3] = fmaf((float)s23[1], d[3], outv[3]);
            }
        }
        acc = outv;
    } else
#pragma unroll 1
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const unsigned qq = q[d] + it;
            if constexpr (V == 0) {
                unsigned l, h; float t0_, t1_;
                asm volatile("v_and_b32 %[l], 0x0f0f0f0f, %[q]\n\tv_and_b32 %[h], 0xf0f0f0f0, %[q]\n\t"
                    "v_cvt_f32_ubyte0 %[t0], %[l]\n\tv_cvt_f32_ubyte0 %[t1], %[h]\n\tv_fmac_f32 %[lo], %[t0], %[a0]\n\tv_fmac_f32 %[hi], %[t1], %[b0]\n\t"
                    "v_cvt_f32_ubyte1 %[t0], %[l]\n\tv_cvt_f32_ubyte1 %[t1], %[h]\n\tv_fmac_f32 %[lo], %[t0], %[a1]\n\tv_fmac_f32 %[hi], %[t1], %[b1]\n\t"
                    "v_cvt_f32_ubyte2 %[t0], %[l]\n\tv_cvt_f32_ubyte2 %[t1], %[h]\n\tv_fmac_f32 %[lo], %[t0], %[a2]\n\tv_fmac_f32 %[hi], %[t1], %[b2]\n\t"
                    "v_cvt_f32_ubyte3 %[t0], %[l]\n\tv_cvt_f32_ubyte3 %[t1], %[h]\n\tv_fmac_f32 %[lo], %[t0], %[a3]\n\tv_fmac_f32 %[hi], %[t1], %[b3]"
                    : [lo] "+v"(lo), [hi] "+v"(hi), [l] "=&v"(l), [h] "=&v"(h), [t0] "=&v"(t0_), [t1] "=&v"(t1_)
                    : [q] "v"(qq), [a0] "v"(x[0]), [a1] "v"(x[1]), [a2] "v"(x[2]), [a3] "v"(x[3]), [b0] "v"(x[4]), [b1] "v"(x[5]), [b2] "v"(x[6]), [b3] "v"(x[7]));
            } else if constexpr (V == 1) {
                const unsigned l = qq & 0x0f0f0f0fu, h = (qq >> 4) & 0x0f0f0f0fu;
                const f2 a = __builtin_amdgcn_cvt_pk_f32_fp8(l, false), b = __builtin_amdgcn_cvt_pk_f32_fp8(l, true);
                const f2 c = __builtin_amdgcn_cvt_pk_f32_fp8(h, false), e = __builtin_amdgcn_cvt_pk_f32_fp8(h, true);
                lo = fmaf(a[0], x[0], lo); hi = fmaf(c[0], x[4], hi);
                lo = fmaf(a[1], x[1], lo); hi = fmaf(c[1], x[5], hi);
                lo = fmaf(b[0], x[2], lo); hi = fmaf(e[0], x[6], hi);
                lo = fmaf(b[1], x[3], lo); hi = fmaf(e[1], x[7], hi);
            } else if constexpr (V == 2) {
                const unsigned l = qq & 0x0f0f0f0fu, h = (qq >> 4) & 0x0f0f0f0fu;
                const f2 a = __builtin_amdgcn_cvt_pk_f32_fp8(l, false), b = __builtin_amdgcn_cvt_pk_f32_fp8(l, true);
                const f2 c = __builtin_amdgcn_cvt_pk_f32_fp8(h, false), e = __builtin_amdgcn_cvt_pk_f32_fp8(h, true);
                const f2 x01 = {x[0], x[1]}, x23 = {x[2], x[3]}, x45 = {x[4], x[5]}, x67 = {x[6], x[7]};
                plo = __builtin_elementwise_fma(a, x01, plo); phi = __builtin_elementwise_fma(c, x45, phi);
                plo = __builtin_elementwise_fma(b, x23, plo); phi = __builtin_elementwise_fma(e, x67, phi);
            } else if constexpr (V == 3) {
                unsigned w0, w1, w2, w3;
                asm volatile("v_and_or_b32 %0, %4, %5, %6\n\tv_lshrrev_b32 %1, 4, %4\n\tv_lshrrev_b32 %2, 8, %4\n\tv_lshrrev_b32 %3, 12, %4\n\t"
                             "v_and_or_b32 %1, %1, %5, %6\n\tv_and_or_b32 %2, %2, %5, %6\n\tv_and_or_b32 %3, %3, %5, %6"
                             : "=&v"(w0), "=&v"(w1), "=&v"(w2), "=&v"(w3) : "v"(qq), "v"(0x000f000fu), "v"(0x43004300u));
                bf8 a;
                a[0] = (short)w0; a[1] = (short)(w0 >> 16); a[2] = (short)w1; a[3] = (short)(w1 >> 16);
                a[4] = (short)w2; a[5] = (short)(w2 >> 16); a[6] = (short)w3; a[7] = (short)(w3 >> 16);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bx, acc, 0, 0, 0);
            } else if constexpr (V == 4) {
                const float f = __uint_as_float(qq);
                lo = fmaf(f, x[0], lo); hi = fmaf(f, x[4], hi); lo = fmaf(f, x[1], lo); hi = fmaf(f, x[5], hi);
                lo = fmaf(f, x[2], lo); hi = fmaf(f, x[6], hi); lo = fmaf(f, x[3], lo); hi = fmaf(f, x[7], hi);
            } else if constexpr (V == 5) {
                float t;
                asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(t) : "v"(qq)); lo += 0.f * t;
                asm volatile("v_cvt_f32_ubyte1 %0, %1\n\tv_cvt_f32_ubyte2 %0, %1\n\tv_cvt_f32_ubyte3 %0, %1\n\tv_cvt_f32_ubyte0 %0, %1\n\t"
                             "v_cvt_f32_ubyte1 %0, %1\n\tv_cvt_f32_ubyte2 %0, %1\n\tv_cvt_f32_ubyte3 %0, %1" : "=v"(t) : "v"(qq));
                hi = t;
            } else {
                f2 t;
                asm volatile("v_cvt_pk_f32_fp8 %0, %1\n\tv_cvt_pk_f32_fp8 %0, %1\n\tv_cvt_pk_f32_fp8 %0, %1\n\tv_cvt_pk_f32_fp8 %0, %1" : "=v"(t) : "v"(qq));
                hi = t[0];
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + tid] = lo + hi + plo[0] + plo[1] + phi[0] + phi[1] + acc[0] + acc[1] + acc[2] + acc[3];
    if ((tid & 63) == 0) cyc[(blockIdx.x * blockDim.x + tid) >> 6] = t1 - t0;
}

template <int V>
void run(int threads, const unsigned* q, const float* x, float* out, unsigned long long* cyc) {
    const int blocks = 256;
    hipLaunchKernelGGL(probe<V>, dim3(blocks), dim3(threads), 0, 0, q, x, out, cyc);
    hipLaunchKernelGGL(probe<V>, dim3(blocks), dim3(threads), 0, 0, q, x, out, cyc);
    hipDeviceSynchronize();
    const int nw = blocks * threads / 64;
    unsigned long long* h = (unsigned long long*)malloc(nw * 8);
    hipMemcpy(h, cyc, nw * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < nw; ++i) s += (double)h[i];
    // __builtin_readcyclecounter = s_memtime = shader cycles
    const double ndw = (V == 7 || V == 10 || V == 11) ? (ITERS / 2) * 32.0 : ITERS * 4.0;
    printf("{\"variant\": %d, \"waves_per_simd\": %d, \"cycles_per_dword\": %.4f}\n", V, threads / 256, s / nw / ndw);
    free(h);
}

int main() {
    unsigned* q; float *x, *out; unsigned long long* cyc;
    hipMalloc(&q, 512 * 16); hipMalloc(&x, 512 * 32); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    hipMemset(q, 0x5a, 512 * 16); hipMemset(x, 0x3c, 512 * 32);
    for (int th : {256, 512}) {
        run<0>(th, q, x, out, cyc); run<1>(th, q, x, out, cyc); run<2>(th, q, x, out, cyc); run<3>(th, q, x, out, cyc);
        run<4>(th, q, x, out, cyc); run<5>(th, q, x, out, cyc); run<6>(th, q, x, out, cyc); run<7>(th, q, x, out, cyc); run<8>(th, q, x, out, cyc); run<9>(th, q, x, out, cyc); run<10>(th, q, x, out, cyc); run<11>(th, q, x, out, cyc);
    }
    return 0;
}
