// Hardware probe (not product code): how fast can gfx950 stream a buffer of a given size when it
// is read repeatedly (Infinity-Cache / L2 residency) vs streamed once, with plain vs non-temporal
// loads; and what a kernel boundary costs inside a hipGraph.  Output: one JSON object per line.
//   hipcc --offload-arch=gfx950 -O3 stream_probe.hip -o stream_probe && ./stream_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));

template <bool NT>
__global__ __launch_bounds__(256) void read_kernel(const v4f* __restrict__ p, size_t n4, float* out) {
    v4f acc = {0, 0, 0, 0};
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 7 * stride < n4; i += 8 * stride) {
        v4f r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = NT ? __builtin_nontemporal_load(p + i + j * stride) : p[i + j * stride];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += r[j];
    }
    for (; i < n4; i += stride) acc += p[i];
    float s = acc.x + acc.y + acc.z + acc.w;
    if (s == 12345.678f) out[blockIdx.x] = s;
}

__global__ void empty_kernel(float* out) { if (out == (float*)1) out[0] = 0; }

int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    const size_t maxb = (size_t)6 << 30;
    char* buf; CK(hipMalloc(&buf, maxb)); CK(hipMemset(buf, 1, maxb));
    float* out; CK(hipMalloc(&out, 1 << 20));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // A: repeated reads of the SAME region of size sz (cache residency), and rotating regions (pure HBM)
    const size_t sizes_mb[] = {8, 16, 32, 64, 96, 128, 192, 256, 384, 512, 1024, 4096};
    for (int nt = 0; nt < 2; ++nt)
        for (int rot = 0; rot < 2; ++rot)
            for (size_t smb : sizes_mb) {
                const size_t sz = smb << 20, n4 = sz / 16;
                const int grids[] = {1024, 2048, 4096};
                for (int grid : grids) {
                    const int iters = (int)(((size_t)8 << 30) / sz); const int it = iters < 5 ? 5 : (iters > 200 ? 200 : iters);
                    const size_t nreg = rot ? maxb / sz : 1;
                    for (int w = 0; w < 2; ++w) { if (nt) read_kernel<true><<<grid, 256, 0, st>>>((const v4f*)buf, n4, out); else read_kernel<false><<<grid, 256, 0, st>>>((const v4f*)buf, n4, out); }
                    CK(hipEventRecord(e0, st));
                    for (int i = 0; i < it; ++i) {
                        const v4f* p = (const v4f*)(buf + (i % nreg) * sz);
                        if (nt) read_kernel<true><<<grid, 256, 0, st>>>(p, n4, out); else read_kernel<false><<<grid, 256, 0, st>>>(p, n4, out);
                    }
                    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    printf("{\"probe\":\"read\",\"nt\":%d,\"rotate\":%d,\"MB\":%zu,\"grid\":%d,\"us\":%.2f,\"GBps\":%.0f}\n", nt, rot, smb, grid,
                           ms * 1000 / it, (double)sz * it / (ms * 1e-3) / 1e9);
                }
            }
    // B: kernel boundary inside a graph: N empty kernels
    {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < 200; ++i) empty_kernel<<<256, 256, 0, st>>>(out);
        CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < 10; ++r) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("{\"probe\":\"graph_empty_kernel\",\"us_per_kernel\":%.3f}\n", ms * 1000 / 2000);
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < 2000; ++i) empty_kernel<<<256, 256, 0, st>>>(out);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("{\"probe\":\"eager_empty_kernel\",\"us_per_kernel\":%.3f}\n", ms * 1000 / 2000);
    }
    return 0;
}
