// Hardware probe (not product code): the q4_0 dots of one 4-row x 128-block tile (the token kernel's tile at K = 4096)
//   (a) as the kernel does them today: lane = block, x fragment in registers, 5 bit operations + 8 v_fma_mix_f32 per dword;
//   (b) on v_mfma_f32_4x4x4_16B_f16 (csrc/q4_mfma.h): lane = (block, row), x = hi + lo f16 pieces read from an LDS image.
// Checks (b)'s operand layout and numerics against a double-precision dot on the host, and prints cycles per tile and wave
// with 8 waves per CU (2 per SIMD), 256 workgroups.
//   hipcc --offload-arch=gfx950 -O3 q4_mfma_probe.hip -o q4_mfma_probe && ./q4_mfma_probe
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../kernels.h"
#include "../q4_mfma.h"

using namespace llmk;

constexpr int K = 4096, NBLK = K / 32, ROWS = 4;
constexpr int RB = K / 2 + NBLK * 2;              // device row: nibble plane, then the f16 scales
constexpr int ITERS = 400;
constexpr int PL = NBLK * Q4M_PB;             // bytes per plane of the image

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// ---- (b) one tile ---------------------------------------------------------------------------------------------------
struct TileM { uint4 q[8]; unsigned short sc[8]; };
__device__ __forceinline__ void load_m(TileM& t, const char* W, int lane) {
    const char* row = W + (size_t)(lane & 3) * RB;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        const int b = 16 * g + (lane >> 2);
        t.q[g] = reinterpret_cast<const uint4*>(row)[b];
        t.sc[g] = reinterpret_cast<const unsigned short*>(row + K / 2)[b];
    }
}
__device__ __forceinline__ float dot_m(const TileM& t, const char* img, const float* xs8, int lane) {
    float acc = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        const int b = 16 * g + (lane >> 2);
        const float s = q4m_block<PL>(t.q[g], img + b * Q4M_PB + (lane & 1) * 16);
        const float d = __half2float(*reinterpret_cast<const __half*>(&t.sc[g]));
        acc = fmaf(d, fmaf(s, Q4M_RESCALE, -xs8[b]), acc);
    }
    // lanes with the same lane & 3: row_shr:4, row_shr:8 inside the DPP row (lanes 12..15 hold the row's sums), then the four rows
    acc += dpp_mov<0x114, 0xf, true>(0.f, acc);
    acc += dpp_mov<0x118, 0xf, true>(0.f, acc);
    acc += __shfl_xor(acc, 16, 64);
    acc += __shfl_xor(acc, 32, 64);
    return acc;                                     // valid in lanes 12..15 (+16 r): row lane & 3
}

template <int PUT>
__global__ __launch_bounds__(64) void check_kernel(const char* W, const float* x, float* out) {
    __shared__ __attribute__((aligned(16))) char img[NBLK * Q4M_BLK];
    __shared__ float xs8[NBLK];
    const int lane = threadIdx.x;
    if (PUT == 2) for (int e = 2 * lane; e < K; e += 128) q4m_put2<PL>(img, e, x[e], x[e + 1]);
    else for (int e = 4 * lane; e < K; e += 256) q4m_put4<PL>(img, e, *reinterpret_cast<const float4*>(x + e));
    for (int b = lane; b < NBLK; b += 64) { float s = 0.f; for (int i = 0; i < 32; ++i) s += x[32 * b + i]; xs8[b] = 8.f * s; }
    __syncthreads();
    TileM t;
    load_m(t, W, lane);
    const float v = dot_m(t, img, xs8, lane);
    if (lane >= 12 && lane < 16) out[lane & 3] = v;
}

template <int V>
__global__ __launch_bounds__(512) void time_kernel(const char* W, const float* x, float* out, unsigned long long* cyc) {
    __shared__ __attribute__((aligned(16))) char img[NBLK * Q4M_BLK];
    __shared__ float xs8[NBLK];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int e = 2 * tid; e < K; e += 1024) q4m_put2<PL>(img, e, x[e], x[e + 1]);
    for (int b = tid; b < NBLK; b += 512) { float s = 0.f; for (int i = 0; i < 32; ++i) s += x[32 * b + i]; xs8[b] = 8.f * s; }
    __syncthreads();
    float r = 0.f;
    unsigned long long t0, t1;
    if constexpr (V == 1) {
        TileM t;
        load_m(t, W, lane);
        t0 = __builtin_readcyclecounter();
        for (int it = 0; it < ITERS; ++it) {
#pragma unroll
            for (int g = 0; g < 8; ++g) asm volatile("" : "+v"(t.q[g].x), "+v"(t.q[g].y), "+v"(t.q[g].z), "+v"(t.q[g].w));
            r += dot_m(t, img, xs8, lane);
        }
        t1 = __builtin_readcyclecounter();
    } else {
        // today's recipe: lane = blocks lane and lane + 64 of every row, x fragment (64 floats) in registers
        uint4 q[8];
        __half d[8];
        float4 xv[16];
        float x8[2];
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const char* row = W + (size_t)s * RB;
                q[s * 2 + jj] = reinterpret_cast<const uint4*>(row)[jj * 64 + lane];
                d[s * 2 + jj] = reinterpret_cast<const __half*>(row + K / 2)[jj * 64 + lane];
            }
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            float s = 0.f;
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                xv[jj * 8 + m] = *reinterpret_cast<const float4*>(x + 32 * (jj * 64 + lane) + 4 * m);
                s += xv[jj * 8 + m].x + xv[jj * 8 + m].y + xv[jj * 8 + m].z + xv[jj * 8 + m].w;
            }
            x8[jj] = 8.f * s;
        }
        t0 = __builtin_readcyclecounter();
        for (int it = 0; it < ITERS; ++it) {
#pragma unroll
            for (int g = 0; g < 8; ++g) asm volatile("" : "+v"(q[g].x), "+v"(q[g].y), "+v"(q[g].z), "+v"(q[g].w));
            float v[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                float acc = 0.f;
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const unsigned w[4] = {q[s * 2 + jj].x, q[s * 2 + jj].y, q[s * 2 + jj].z, q[s * 2 + jj].w};
                    float tl = 0.f, th = 0.f;
#pragma unroll
                    for (int i = 0; i < 4; ++i) q4_dword_dot(w[i], xv[jj * 8 + i], xv[jj * 8 + 4 + i], tl, th);
                    acc = fmaf(__half2float(d[s * 2 + jj]), q4_block_fold(tl, th) - x8[jj], acc);
                }
                v[s] = wave_sum(acc);
            }
            r += (v[0] + v[1]) + (v[2] + v[3]);
        }
        t1 = __builtin_readcyclecounter();
    }
    if (lane == 0) cyc[blockIdx.x * 8 + (tid >> 6)] = t1 - t0;
    if (r == 12345.678f) out[0] = r;
}

int main() {
    srand(20260929);
    std::vector<unsigned char> W(ROWS * RB);
    std::vector<float> x(K);
    std::vector<double> ref(ROWS, 0.0);
    for (int i = 0; i < K; ++i) {
        const float u = (float)(rand() & 0xffffff) / 16777216.f * 2.f - 1.f;
        x[i] = u * expf(((float)(rand() & 0xffffff) / 16777216.f * 14.f) - 9.f);       // magnitudes from 1e-4 to 150
    }
    for (int r = 0; r < ROWS; ++r) {
        unsigned char* row = W.data() + (size_t)r * RB;
        for (int i = 0; i < K / 2; ++i) row[i] = (unsigned char)(rand() & 0xff);
        for (int b = 0; b < NBLK; ++b) {
            const __half d = __float2half(((float)(rand() & 0xffffff) / 16777216.f - 0.5f) * 0.05f);
            reinterpret_cast<__half*>(row + K / 2)[b] = d;
            double s = 0.0;
            for (int i = 0; i < 16; ++i) {
                s += ((row[b * 16 + i] & 15) - 8) * (double)x[32 * b + i];
                s += ((row[b * 16 + i] >> 4) - 8) * (double)x[32 * b + 16 + i];
            }
            ref[r] += (double)__half2float(d) * s;
        }
    }
    char* dW; float *dx, *dout; unsigned long long* dc;
    CK(hipMalloc(&dW, W.size())); CK(hipMalloc(&dx, K * 4)); CK(hipMalloc(&dout, 64)); CK(hipMalloc(&dc, 256 * 8 * 8));
    CK(hipMemcpy(dW, W.data(), W.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(dx, x.data(), K * 4, hipMemcpyHostToDevice));
    double norm = 0.0;
    for (int r = 0; r < ROWS; ++r) norm = fmax(norm, fabs(ref[r]));
    for (int put = 2; put <= 4; put += 2) {
        float out[4];
        if (put == 2) hipLaunchKernelGGL(check_kernel<2>, dim3(1), dim3(64), 0, 0, dW, dx, dout);
        else hipLaunchKernelGGL(check_kernel<4>, dim3(1), dim3(64), 0, 0, dW, dx, dout);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(out, dout, 16, hipMemcpyDeviceToHost));
        double worst = 0.0;
        for (int r = 0; r < ROWS; ++r) worst = fmax(worst, fabs(out[r] - ref[r]) / norm);
        printf("{\"probe\": \"q4_mfma\", \"check\": \"put%d\", \"rows\": [%.9g, %.9g, %.9g, %.9g], \"ref\": [%.9g, %.9g, %.9g, %.9g], \"max_rel_err\": %.3e}\n",
               put, out[0], out[1], out[2], out[3], ref[0], ref[1], ref[2], ref[3], worst);
    }
    for (int v = 0; v < 2; ++v) {
        for (int rep = 0; rep < 2; ++rep) {
            if (v == 0) hipLaunchKernelGGL(time_kernel<0>, dim3(256), dim3(512), 0, 0, dW, dx, dout, dc);
            else hipLaunchKernelGGL(time_kernel<1>, dim3(256), dim3(512), 0, 0, dW, dx, dout, dc);
            CK(hipDeviceSynchronize());
        }
        std::vector<unsigned long long> c(256 * 8);
        CK(hipMemcpy(c.data(), dc, c.size() * 8, hipMemcpyDeviceToHost));
        double s = 0.0; unsigned long long mx = 0;
        for (auto t : c) { s += (double)t; mx = t > mx ? t : mx; }
        printf("{\"probe\": \"q4_mfma\", \"variant\": \"%s\", \"cycles_per_tile_per_wave_avg\": %.1f, \"max\": %.1f, \"counter\": \"s_memtime clocks, 8 waves per CU\"}\n",
               v == 0 ? "valu fma_mix, x in registers" : "mfma 4x4x4 f16, x from LDS", s / c.size() / ITERS, (double)mx / ITERS);
    }
    return 0;
}
