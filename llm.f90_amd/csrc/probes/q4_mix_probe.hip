// Hardware probe (not product code): q4_0 nibbles as f16 SUBNORMALS.
// A 16-bit half whose only set bits are a nibble at bits 0-3 is the f16 subnormal n * 2^-24 (at bits 4-7: 16 n * 2^-24),
// so (q & 0x000F000F) / (q & 0x00F000F0) ARE two weights each, with no conversion instruction at all, and
// v_fma_mix_f32 (f16 source selected from the low / high half by op_sel, f32 multiplicand, f32 accumulator) multiplies them
// exactly: 1 shift + 4 ands + 8 fma_mix = 13 full-rate VALU operations per dword (8 weights) against 2 ands + 8
// half-rate v_cvt_f32_ubyteN + 8 v_fmac of the token kernel's recipe.  Questions: (1) does v_fma_mix_f32 honour f16
// subnormal inputs on gfx950 under HIP's default mode register? (2) is the result bit-identical to the cvt recipe
// after the exact 2^24 rescale? (3) cycles per dword, 1 and 2 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 q4_mix_probe.hip -o q4_mix_probe && ./q4_mix_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

constexpr int ITERS = 4000;

__device__ __forceinline__ void dword_cvt(unsigned q, const float4& xl, const float4& xh, float& lo, float& hi16) {
    unsigned l, h;
    float t0, t1;
    asm("v_and_b32 %[l], 0x0f0f0f0f, %[q]\n\t"
        "v_and_b32 %[h], 0xf0f0f0f0, %[q]\n\t"
        "v_cvt_f32_ubyte0 %[t0], %[l]\n\t"
        "v_cvt_f32_ubyte0 %[t1], %[h]\n\t"
        "v_fmac_f32 %[lo], %[t0], %[a0]\n\t"
        "v_fmac_f32 %[hi], %[t1], %[b0]\n\t"
        "v_cvt_f32_ubyte1 %[t0], %[l]\n\t"
        "v_cvt_f32_ubyte1 %[t1], %[h]\n\t"
        "v_fmac_f32 %[lo], %[t0], %[a1]\n\t"
        "v_fmac_f32 %[hi], %[t1], %[b1]\n\t"
        "v_cvt_f32_ubyte2 %[t0], %[l]\n\t"
        "v_cvt_f32_ubyte2 %[t1], %[h]\n\t"
        "v_fmac_f32 %[lo], %[t0], %[a2]\n\t"
        "v_fmac_f32 %[hi], %[t1], %[b2]\n\t"
        "v_cvt_f32_ubyte3 %[t0], %[l]\n\t"
        "v_cvt_f32_ubyte3 %[t1], %[h]\n\t"
        "v_fmac_f32 %[lo], %[t0], %[a3]\n\t"
        "v_fmac_f32 %[hi], %[t1], %[b3]"
        : [lo] "+v"(lo), [hi] "+v"(hi16), [l] "=&v"(l), [h] "=&v"(h), [t0] "=&v"(t0), [t1] "=&v"(t1)
        : [q] "v"(q), [a0] "v"(xl.x), [a1] "v"(xl.y), [a2] "v"(xl.z), [a3] "v"(xl.w), [b0] "v"(xh.x), [b1] "v"(xh.y),
          [b2] "v"(xh.z), [b3] "v"(xh.w));
}

// bytes 4j..4j+3 of a block: byte i = elem (low nibble) | elem 16+i (high nibble).  In the dword: bits 0-3 elem 0,
// 4-7 elem 16, 8-11 elem 1, 12-15 elem 17, 16-19 elem 2, 20-23 elem 18, 24-27 elem 3, 28-31 elem 19 (relative to 4j).
// lo += 2^-24 sum n_lo x ; hi16 += 2^-24 sum (16 n_hi) x
__device__ __forceinline__ void dword_mix(unsigned q, const float4& xl, const float4& xh, float& lo, float& hi16) {
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

template <int V>
__global__ __launch_bounds__(512) void check(const unsigned* __restrict__ qin, const float* __restrict__ xin, float* out) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    float4 xl, xh;
    xl = reinterpret_cast<const float4*>(xin)[tid * 2];
    xh = reinterpret_cast<const float4*>(xin)[tid * 2 + 1];
    float lo = 0.f, hi = 0.f;
    for (int i = 0; i < 4; ++i) {      // a chain of 4 dwords, as in a block
        if (V == 0) dword_cvt(qin[tid * 4 + i], xl, xh, lo, hi);
        else dword_mix(qin[tid * 4 + i], xl, xh, lo, hi);
    }
    if (V == 1) { lo *= 16777216.0f; hi *= 16777216.0f; }
    out[tid * 2] = lo;
    out[tid * 2 + 1] = hi;
}

template <int V>
__global__ __launch_bounds__(512) void timeit(const unsigned* __restrict__ qin, const float* __restrict__ xin, float* out,
                                              unsigned long long* cyc) {
    const int tid = threadIdx.x;
    unsigned q[4];
    for (int i = 0; i < 4; ++i) q[i] = qin[tid * 4 + i];
    const float4 xl = reinterpret_cast<const float4*>(xin)[tid * 2], xh = reinterpret_cast<const float4*>(xin)[tid * 2 + 1];
    float lo[4] = {0, 0, 0, 0}, hi[4] = {0, 0, 0, 0};
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (V == 0) dword_cvt(q[i] + it, xl, xh, lo[i], hi[i]);
            else dword_mix(q[i] + it, xl, xh, lo[i], hi[i]);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + tid] = (lo[0] + lo[1] + lo[2] + lo[3]) + (hi[0] + hi[1] + hi[2] + hi[3]);
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    const int N = 512 * 64;
    unsigned* hq = (unsigned*)malloc(N * 4 * sizeof(unsigned));
    float* hx = (float*)malloc(N * 8 * sizeof(float));
    srand(7);
    for (int i = 0; i < N * 4; ++i) hq[i] = ((unsigned)rand() << 16) ^ (unsigned)rand();
    for (int i = 0; i < N * 8; ++i) hx[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
    hq[0] = 0xFFFFFFFFu; hq[1] = 0; hq[2] = 0x0F0F0F0Fu; hq[3] = 0xF0F0F0F0u;
    unsigned *dq; float *dx, *o0, *o1; unsigned long long* dc;
    hipMalloc(&dq, N * 16); hipMalloc(&dx, N * 32); hipMalloc(&o0, N * 8); hipMalloc(&o1, N * 8); hipMalloc(&dc, 4096);
    hipMemcpy(dq, hq, N * 16, hipMemcpyHostToDevice); hipMemcpy(dx, hx, N * 32, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(check<0>, dim3(N / 512), dim3(512), 0, 0, dq, dx, o0);
    hipLaunchKernelGGL(check<1>, dim3(N / 512), dim3(512), 0, 0, dq, dx, o1);
    float* r0 = (float*)malloc(N * 8); float* r1 = (float*)malloc(N * 8);
    hipMemcpy(r0, o0, N * 8, hipMemcpyDeviceToHost); hipMemcpy(r1, o1, N * 8, hipMemcpyDeviceToHost);
    int bad = 0, zero = 0; double maxrel = 0;
    for (int i = 0; i < N * 2; ++i) {
        if (memcmp(&r0[i], &r1[i], 4)) { ++bad; double d = fabs((double)r0[i] - r1[i]) / (fabs((double)r0[i]) + 1e-30); if (d > maxrel) maxrel = d; }
        if (r1[i] == 0.f && r0[i] != 0.f) ++zero;
    }
    printf("{\"probe\": \"q4_mix\", \"values\": %d, \"not_bit_identical\": %d, \"flushed_to_zero\": %d, \"max_rel_diff\": %.3g, \"sample\": [%g, %g, %g, %g]}\n",
           N * 2, bad, zero, maxrel, r0[0], r1[0], r0[1], r1[1]);
    for (int waves = 1; waves <= 2; ++waves) {
        const int threads = 256 * waves;       // 4 or 8 waves per workgroup, one workgroup per CU: 1 or 2 waves per SIMD
        unsigned long long c[2][256];
        for (int v = 0; v < 2; ++v) {
            for (int rep = 0; rep < 2; ++rep) {
                if (v == 0) hipLaunchKernelGGL(timeit<0>, dim3(256), dim3(threads), 0, 0, dq, dx, o0, dc);
                else hipLaunchKernelGGL(timeit<1>, dim3(256), dim3(threads), 0, 0, dq, dx, o0, dc);
                hipDeviceSynchronize();
            }
            hipMemcpy(c[v], dc, 256 * 8, hipMemcpyDeviceToHost);
        }
        double a0 = 0, a1 = 0;
        for (int b = 0; b < 256; ++b) { a0 += c[0][b]; a1 += c[1][b]; }
        printf("{\"probe\": \"q4_mix\", \"waves_per_simd\": %d, \"cycles_per_dword_cvt\": %.2f, \"cycles_per_dword_mix\": %.2f}\n", waves,
               a0 / 256 / ITERS / 4, a1 / 256 / ITERS / 4);
    }
    return 0;
}
