// Hardware probe (not product code): is a 16-byte aligned agent-scope (sc1) store observed atomically by a 16-byte
// sc1 load issued from another CU / XCD?  Writer workgroups store {i, i, i, i} (one dwordx4 buffer store per lane)
// to their slots for i = 1..ITERS; reader workgroups sweep all slots with 16-byte sc1 buffer loads and count
// vectors whose four words differ (a torn write) and vectors that moved backwards.
//   hipcc --offload-arch=gfx950 -O3 granule16_probe.hip -o granule16_probe && ./granule16_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef unsigned v4u __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

__global__ __launch_bounds__(256) void probe(unsigned* slots, int nslots, int nwriters, unsigned iters, unsigned long long* stats,
                                             unsigned* done) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const __amdgpu_buffer_rsrc_t rs = rsrc(slots, nslots * 16);
    if (b < nwriters) {   // writer: slots [b*256, b*256+256)
        const int s = b * 256 + tid;
        for (unsigned i = 1; i <= iters; ++i) {
            const v4u v = {i, i, i, i};
            __builtin_amdgcn_raw_buffer_store_b128(v, rs, s * 16, 0, 16 /* sc1 */);
        }
        __syncthreads();
        if (tid == 0) atomicAdd(done, 1u);
    } else {              // reader: sweeps every writer slot until all writers are done
        unsigned long long torn = 0, reads = 0, back = 0;
        unsigned last = 0;
        const int total = nwriters * 256;
        for (;;) {
            for (int s = tid + (b - nwriters) * 7 % 256; s < total; s += 256) {
                const v4u r = __builtin_amdgcn_raw_buffer_load_b128(rs, s * 16, 0, 16);
                ++reads;
                if (r.x != r.y || r.x != r.z || r.x != r.w) ++torn;
                if (s == tid) { if (r.x < last) ++back; last = r.x; }
            }
            if (__hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)nwriters) break;
        }
        atomicAdd(&stats[0], reads);
        atomicAdd(&stats[1], torn);
        atomicAdd(&stats[2], back);
    }
}

int main(int argc, char** argv) {
    const unsigned iters = argc > 1 ? (unsigned)atol(argv[1]) : 200000u;
    const int nwriters = 128, nreaders = 128, nslots = nwriters * 256;
    unsigned *slots, *done;
    unsigned long long* stats;
    CK(hipMalloc(&slots, (size_t)nslots * 16));
    CK(hipMalloc(&done, 4));
    CK(hipMalloc(&stats, 24));
    CK(hipMemset(slots, 0, (size_t)nslots * 16));
    CK(hipMemset(done, 0, 4));
    CK(hipMemset(stats, 0, 24));
    hipLaunchKernelGGL(probe, dim3(nwriters + nreaders), dim3(256), 0, 0, slots, nslots, nwriters, iters, stats, done);
    CK(hipDeviceSynchronize());
    unsigned long long h[3];
    CK(hipMemcpy(h, stats, 24, hipMemcpyDeviceToHost));
    printf("{\"probe\": \"granule16\", \"writers\": %d, \"readers\": %d, \"stores_per_slot\": %u, \"reads\": %llu, \"torn\": %llu, \"backwards\": %llu}\n",
           nwriters, nreaders, iters, h[0], h[1], h[2]);
    return 0;
}
