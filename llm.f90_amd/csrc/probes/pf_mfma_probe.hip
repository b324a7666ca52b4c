// Hardware probe (not product code): what does the inner loop of the prefill GEMM (prefill.h pf_gemm_kernel) sustain on
// gfx950 when nothing but the matrix core and LDS is involved?  Weights sit in registers, the activation tile in LDS; no
// global loads, no barriers.  One 64-column step per iteration, 4 waves per block, grid = 256 * BPC blocks.
//   0  v_mfma_f32_16x16x4_f32, wave tile 16 rows x  64 positions (4 accumulators):  16 ds_read_b128 +  64 MFMA / step
//   1  v_mfma_f32_16x16x4_f32, wave tile 16 rows x 128 positions (8 accumulators):  32 ds_read_b128 + 128 MFMA / step
//   2  v_mfma_f32_32x32x2_f32, wave tile 32 rows x  64 positions (2 x 16 acc):      16 ds_read_b128 +  64 MFMA / step
//   3  v_mfma_f32_32x32x2_f32, wave tile 32 rows x 128 positions (4 x 16 acc):      32 ds_read_b128 + 128 MFMA / step
//   4  16x16x4, 4 accumulators, operands from registers only (matrix-core ceiling at this occupancy)
//   5  32x32x2, 2 accumulators, operands from registers only
//   6  v_mfma_f32_16x16x4_f32, wave tile 32 rows x 128 positions (2 x 8 accumulators): 32 ds_read_b128 + 256 MFMA / step
//   7  variant 1 + what the GEMM adds per step: 8 ds_write_b128 per lane into the other buffer, then __syncthreads
//   8  variant 7 with the barrier only (no LDS writes)        9  variant 7 with the writes only (no barrier)
//  10  variant 7 + the weight stream: 4 global_load_dwordx4 (nt) per lane per step from a 92 MB matrix, used two steps later
//  11  variant 10 + the activation staging: the 8 vectors written to LDS come from global memory (L2-resident), one step ahead;
//      every block reads the SAME 32 KB tile at the same time
//  12  variant 11 with the blocks spread over the 32 column tiles (block b starts at tile 11*b, as the GEMM's units do)
//  13  variant 12 with the activation loads TWO steps ahead of their LDS writes (second register set)
//  14  variant 12 reading a tile-contiguous activation layout ([tile][128][64]: 32 KB in a row instead of 128 rows 8 KB apart)
// Prints TFLOP/s for 1..4 blocks per CU.
//   hipcc --offload-arch=gfx950 -O3 pf_mfma_probe.hip -o pf_mfma_probe && ./pf_mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

constexpr int LDW = 68;

template <int V>
struct Cfg {
    static constexpr int TOK = (V == 0 || V == 2 || V == 4 || V == 5) ? 64 : 128;   // 7, 8, 9: 128
    static constexpr int ROWS = (V == 2 || V == 3 || V == 5 || V == 6) ? 32 : 16;
};

template <int V>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ win, const float* __restrict__ xin, float* out, int iters,
                                             const float* __restrict__ wbig, const float* __restrict__ xbig) {
    constexpr int TOK = Cfg<V>::TOK;
    extern __shared__ __attribute__((aligned(16))) float xs[];   // [2][TOK][LDW]
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 2 * TOK * LDW; i += 256) xs[i] = xin[i % 4096];
    float w[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) w[i] = win[(tid * 32 + i) % 4096];
    __syncthreads();
    float res = 0.f;
    if constexpr (V == 0 || V == 1 || V == 4 || V == 6 || V >= 7) {
        constexpr int NG = TOK / 16, NR = Cfg<V>::ROWS / 16;
        v4f acc[NR][NG];
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int g = 0; g < NG; ++g) acc[r][g] = (v4f){0.f, 0.f, 0.f, 0.f};
        const int li = lane & 15, lk = (lane >> 4) * 4;
        // V >= 10: this wave's 16 rows of a [rows][2048] f32 matrix, 64 columns per step
        const float* wrow = wbig + ((size_t)((blockIdx.x * 4 + (tid >> 6)) * 16 + li) % 11264) * 2048 + lk;
        v4f wa[4], wb[4], xq[8], xq2[8];
        if constexpr (V >= 10) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { wa[j] = *reinterpret_cast<const v4f*>(wrow + j * 16); wb[j] = *reinterpret_cast<const v4f*>(wrow + 64 + j * 16); }
#pragma unroll
            for (int i = 0; i < 8; ++i) xq[i] = xq2[i] = (v4f){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll 1
        for (int s = 0; s < iters; ++s) {
            const float* xb = xs + (s & 1) * TOK * LDW;
            if constexpr (V >= 10) {
                if constexpr (V >= 11) {
                    float* xw = xs + ((s + 1) & 1) * TOK * LDW;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int idx = tid + i * 256, t = idx / 16, c4 = idx % 16;
                        *reinterpret_cast<v4f*>(xw + t * LDW + c4 * 4) = xq[i];
                        if constexpr (V == 13) xq[i] = xq2[i];
                    }
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int idx = tid + i * 256, t = idx / 16, c4 = idx % 16;
                        const int tile = (s + 2 + (V >= 12 ? blockIdx.x * 11 : 0)) & 31;
                        const v4f ld = *reinterpret_cast<const v4f*>(V == 14 ? xbig + (size_t)tile * 8192 + idx * 4 : xbig + (size_t)t * 2048 + tile * 64 + c4 * 4);
                        if constexpr (V == 13) xq2[i] = ld; else xq[i] = ld;
                    }
                }
                // stage rotation by value: w[] <- wa <- wb <- fresh load (the compiler renames, no moves survive)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    w[j * 4 + 0] = wa[j][0]; w[j * 4 + 1] = wa[j][1]; w[j * 4 + 2] = wa[j][2]; w[j * 4 + 3] = wa[j][3];
                    wa[j] = wb[j];
                    wb[j] = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(wrow + ((s + 2) & 31) * 64 + j * 16));
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v4f x[NG];
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    if constexpr (V == 4) x[g] = (v4f){w[16 + g], w[17 + g], w[18 + g], w[19 + g]};
                    else x[g] = *reinterpret_cast<const v4f*>(xb + (g * 16 + li) * LDW + j * 16 + lk);
                }
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int r = 0; r < NR; ++r)
#pragma unroll
                        for (int g = 0; g < NG; ++g)
                            acc[r][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[(r * 4 + j) * 4 + c], x[g][c], acc[r][g], 0, 0, 0);
            }
            if constexpr (V == 7 || V == 9 || V == 10) {
                float* xw = xs + ((s + 1) & 1) * TOK * LDW;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int idx = tid + i * 256, t = idx / 16, c4 = idx % 16;
                    *reinterpret_cast<v4f*>(xw + t * LDW + c4 * 4) = (v4f){w[i], w[i + 1], w[i + 2], w[i + 3]};
                }
            }
            if constexpr (V == 7 || V == 8 || V >= 10) __syncthreads();
            asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int g = 0; g < NG; ++g) res += acc[r][g][0] + acc[r][g][1] + acc[r][g][2] + acc[r][g][3];
    } else {
        constexpr int NT = TOK / 32;
        v16f acc[NT];
#pragma unroll
        for (int g = 0; g < NT; ++g)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[g][i] = 0.f;
        const int li = lane & 31, lk = (lane >> 5) * 4;
#pragma unroll 1
        for (int s = 0; s < iters; ++s) {
            const float* xb = xs + (s & 1) * TOK * LDW;
#pragma unroll
            for (int j = 0; j < 8; ++j) {          // 8 columns per round: 2 lane groups x 4
                v4f x[NT];
#pragma unroll
                for (int g = 0; g < NT; ++g) {
                    if constexpr (V == 5) x[g] = (v4f){w[8 + g], w[9 + g], w[10 + g], w[11 + g]};
                    else x[g] = *reinterpret_cast<const v4f*>(xb + (g * 32 + li) * LDW + j * 8 + lk);
                }
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int g = 0; g < NT; ++g)
                        acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[j * 4 + c], x[g][c], acc[g], 0, 0, 0);
            }
            asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int g = 0; g < NT; ++g)
#pragma unroll
            for (int i = 0; i < 16; ++i) res += acc[g][i];
    }
    out[blockIdx.x * 256 + tid] = res;
}

template <int V>
void run(const float* w, const float* x, float* out, const float* wbig = nullptr, const float* xbig = nullptr, int iters = 400) {
    constexpr int TOK = Cfg<V>::TOK, ROWS = Cfg<V>::ROWS;
    const size_t smem = (size_t)2 * TOK * LDW * sizeof(float);
    hipFuncSetAttribute((const void*)probe<V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    int occ = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, probe<V>, 256, smem);
    printf("variant %d (%d rows x %d positions per wave, occupancy %d blocks/CU):", V, ROWS, TOK, occ);
    for (int bpc = 1; bpc <= 4 && bpc <= occ; ++bpc) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        probe<V><<<256 * bpc, 256, smem>>>(w, x, out, iters, wbig, xbig);
        hipEventRecord(e0);
        probe<V><<<256 * bpc, 256, smem>>>(w, x, out, iters, wbig, xbig);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double flop = 2.0 * ROWS * TOK * 64 * (double)iters * 4 * 256 * bpc;
        printf("  %d/CU %.1f TF", bpc, flop / (ms * 1e-3) / 1e12);
    }
    printf("\n");
}

int main() {
    float *w, *x, *out;
    hipMalloc(&w, 4096 * 4); hipMalloc(&x, 4096 * 4); hipMalloc(&out, 1024 * 256 * 4);
    float h[4096];
    for (int i = 0; i < 4096; ++i) h[i] = (float)((i * 2654435761u) >> 20) / 4096.f - 0.5f;
    hipMemcpy(w, h, sizeof(h), hipMemcpyHostToDevice);
    hipMemcpy(x, h, sizeof(h), hipMemcpyHostToDevice);
    run<4>(w, x, out); run<5>(w, x, out);
    run<0>(w, x, out); run<1>(w, x, out); run<2>(w, x, out); run<3>(w, x, out); run<6>(w, x, out);
    run<7>(w, x, out); run<8>(w, x, out); run<9>(w, x, out);
    float *wbig, *xbig;
    hipMalloc(&wbig, (size_t)11264 * 2048 * 4); hipMalloc(&xbig, (size_t)128 * 2048 * 4);
    hipMemset(wbig, 0, (size_t)11264 * 2048 * 4); hipMemset(xbig, 0, (size_t)128 * 2048 * 4);
    run<10>(w, x, out, wbig, xbig); run<11>(w, x, out, wbig, xbig); run<12>(w, x, out, wbig, xbig); run<13>(w, x, out, wbig, xbig); run<14>(w, x, out, wbig, xbig);
    printf("11 steps per launch (the w1|w3 GEMM's block length), launch overhead included:\n");
    run<7>(w, x, out, wbig, xbig, 11); run<10>(w, x, out, wbig, xbig, 11); run<11>(w, x, out, wbig, xbig, 11); run<12>(w, x, out, wbig, xbig, 11);
    return 0;
}
