// Hardware probe (not product code): ceiling of a persistent "few fat waves" streaming engine.
// One workgroup per CU, W waves, each wave keeps NB tiles of 8 x 16-byte nt loads (8 KB) in flight
// in a register ring (consume oldest, refill), no barriers, no dependencies.
//   hipcc --offload-arch=gfx950 -O3 ring_probe.hip -o ring_probe && ./ring_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));

template <int NB>
struct Ring { v4f b[NB][8]; };

template <int NB, int K>
__device__ __forceinline__ void step(Ring<NB>& r, const v4f* base, size_t& next, size_t stride, v4f& acc, int lane) {
    if constexpr (K < NB) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += r.b[K][j];
        const v4f* p = base + next;
#pragma unroll
        for (int j = 0; j < 8; ++j) r.b[K][j] = __builtin_nontemporal_load(p + lane + j * 64);
        next += stride;
        step<NB, K + 1>(r, base, next, stride, acc, lane);
    }
}

template <int NB>
__global__ __launch_bounds__(1024) void ring_kernel(const v4f* p, size_t ntiles_per_wave, float* out) {
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    const size_t wave = (size_t)blockIdx.x * nw + wid;
    const size_t total_waves = (size_t)gridDim.x * nw;
    // tile t of wave w is at (t * total_waves + w) * 512 float4 (8 KB)
    const v4f* base = p + wave * 512;
    const size_t stride = total_waves * 512;
    Ring<NB> r;
    size_t next = 0;
#pragma unroll
    for (int k = 0; k < NB; ++k) {
#pragma unroll
        for (int j = 0; j < 8; ++j) r.b[k][j] = __builtin_nontemporal_load(base + next + lane + j * 64);
        next += stride;
    }
    v4f acc = {0, 0, 0, 0};
    for (size_t t = 0; t + NB <= ntiles_per_wave; t += NB) step<NB, 0>(r, base, next, stride, acc, lane);
    float s = acc.x + acc.y + acc.z + acc.w;
    if (s == 12345.678f) out[wave] = s;
}

template <int NB>
void run(const v4f* buf, size_t bytes, int waves, float* out, hipStream_t st) {
    const int grid = 256;
    const size_t total_waves = (size_t)grid * waves;
    size_t ntiles = bytes / 8192 / total_waves - NB - 1;   // stay inside the buffer incl. run-ahead
    ntiles = ntiles / NB * NB;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    ring_kernel<NB><<<grid, waves * 64, 0, st>>>(buf, ntiles, out);
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < 3; ++i) ring_kernel<NB><<<grid, waves * 64, 0, st>>>(buf, ntiles, out);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double b = (double)ntiles * total_waves * 8192 * 3;
    printf("{\"probe\":\"ring\",\"waves_per_cu\":%d,\"NB\":%d,\"KB_in_flight_per_cu\":%d,\"GBps\":%.0f}\n", waves, NB, waves * NB * 8,
           b / (ms * 1e-3) / 1e9);
}

int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    const size_t bytes = (size_t)4 << 30;
    char* buf; CK(hipMalloc(&buf, bytes)); CK(hipMemset(buf, 1, bytes));
    float* out; CK(hipMalloc(&out, 1 << 20));
    const int ws[] = {4, 7, 8, 12, 16};
    for (int w : ws) {
        run<1>((const v4f*)buf, bytes, w, out, st);
        run<2>((const v4f*)buf, bytes, w, out, st);
        run<4>((const v4f*)buf, bytes, w, out, st);
        if (w <= 8) run<6>((const v4f*)buf, bytes, w, out, st);
    }
    return 0;
}
