// Does v_mfma_f32_16x16x32_f16 honour f16 subnormal INPUTS on gfx950?  (prefill.h pf_gemm_h_kernel feeds the low piece of an
// activation, which is an f16 subnormal for |x| < 2^-3 unless it is scaled.)  A = 1.0 everywhere, B = a subnormal s in every
// element: D = 32 s if subnormals are honoured, 0 if they are flushed.  Also prints an exactness check: A = 1 + 2^-10
// (largest-significand f16), B = 1 + 2^-10: the product needs 21 significand bits, D = 32 (1 + 2^-10)^2 exactly in f32.
//   hipcc -O3 --offload-arch=gfx950 mfma_f16_denorm_probe.hip -o /tmp/mfma_f16_denorm_probe && /tmp/mfma_f16_denorm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void k(float* out, float bval, float aval) {
    v8h a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)aval; b[i] = (_Float16)bval; }
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = acc[0];
}
int main() {
    float* d; hipMalloc(&d, 4);
    const float sub = 3.0e-6f;       // f16 subnormal (min normal 6.1e-5): rounds to 50 * 2^-24 = 2.98e-6
    float h;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, sub, 1.0f); hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    printf("{\"probe\": \"mfma_f16_subnormal_input\", \"expected\": %.9g, \"got\": %.9g, \"honoured\": %s}\n", 32.0 * (double)(float)(_Float16)sub, h, h != 0.f ? "true" : "false");
    const float q = 1.0f + 0.0009765625f;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, q, q); hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    printf("{\"probe\": \"mfma_f16_product_exact\", \"expected\": %.9g, \"got\": %.9g}\n", (double)(32.0f * q * q), h);
    return 0;
}
