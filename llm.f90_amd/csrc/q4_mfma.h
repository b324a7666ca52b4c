// q4_0 dots on the matrix core in f16 form, for 4-row tiles (token_kernel.h, LLMK_TK_Q4_MFMA; round 4, experimental).
//
// The arithmetic of /root/reference's dequantised weights, sum_k (n_k - 8) d x_k per 32-weight block (the reference itself
// reads f32 weights, llama2.f90:480-640; q4_0 is ggml's block_q4_0), is kept; what changes is who multiplies:
//   * a nibble where it lies in a 16-bit half IS an f16 subnormal (0x000n = n 2^-24, 0x00n0 = 16 n 2^-24: kernels.h,
//     q4_dword_dot), and the matrix instructions honour subnormal inputs (probes/mfma_f16_denorm_probe.hip);
//   * x = hi + lo, two f16 pieces (hi = x toward zero, lo = x - hi toward zero: |x - hi - lo| <= 2^-21 |x|, or 2^-25 absolute
//     where lo is subnormal), every product nibble x piece exact in the instruction's f32 accumulator;
//   * v_mfma_f32_4x4x4_16B_f16 = 16 independent 4x4x4 products: matrix block beta = lane / 4 is ONE q4_0 block of FOUR weight
//     rows.  B (columns j) = lane (beta, j)'s own nibbles -- the lane that loaded row j's 16 bytes of block beta --, A (rows i)
//     = piece i of x for that block, D[i][j] lands in lane (beta, j), register i: the lane that holds row j's nibbles and its
//     block scale receives row j's hi and lo sums.  Rows 2 and 3 of A are never read back (whatever lanes (beta, 2..3) hold).
//   One dword of nibbles = 5 bit operations + 2 matrix instructions instead of 5 + 8 v_fma_mix_f32.
// What it costs: x cannot live in registers any more (a lane meets 8 different blocks per tile): 4 ds_read_b128 per block, and
// activations beyond the f16 range (|x| >= 65504) are not representable in the hi piece (unchecked here: the instantiation is a
// measurement, not the product path).
// MEASURED (MI355X, profiles/r04_q4_mfma_probe.jsonl, r04_ab.jsonl): 2,930 clocks per tile and wave against 3,300 for the VALU
// recipe, and in the kernel 850 tok/s against 890 (parity green): a 2-pass matrix instruction takes the issue slot like a VALU
// operation.  NOT the product path (LLMK_TK_Q4_MFMA = 0); kept as the record of the experiment.
//
// Image of x in LDS (built by the gathers, q4m_put2 / q4m_put4): FOUR PLANES, one per dword j = 0..3 of a block; in a plane
// 32 bytes per block = piece 0 (16 bytes) | piece 1, a piece = [L_j | H_j], L_j = x[4j], x[4j+2], x[4j+1], x[4j+3] (the order the
// masks 0x000f000f on q and on q >> 8 leave dword j's LOW nibbles in), H_j = x[16+4j], x[16+4j+2], x[16+4j+1], x[16+4j+3], each
// / 16 (the high nibbles stay where they are: 16 n 2^-24).  Planes, because the sixteen groups of a wave read the same j at the
// same time: 16 blocks x 2 pieces x 16 bytes = 512 contiguous bytes per ds_read_b128, no bank conflict.  (First version: 128
// contiguous bytes per block -- the 32 addresses of an instruction fell on 8 of the 32 banks: probes/q4_mfma_probe measured the
// whole tile at 3,018 clocks per wave, the planes at 2,930.)
#pragma once
#include <hip/hip_runtime.h>

namespace llmk {

typedef _Float16 q4m_h4 __attribute__((ext_vector_type(4)));
typedef float q4m_f4 __attribute__((ext_vector_type(4)));
typedef unsigned q4m_u2 __attribute__((ext_vector_type(2)));

constexpr int Q4M_BLK = 128;                       // image bytes per block (all four planes together)
constexpr int Q4M_PB = 32;                         // bytes per block in one plane
constexpr float Q4M_RESCALE = 16777216.0f;         // the sums are 2^-24 times the integers' (exact)

// byte offset of element e of the vector (hi piece; the lo piece 16 bytes further); PLANE = bytes per plane = blocks x 32
template <int PLANE>
__host__ __device__ constexpr int q4m_elem_off(int e) {
    const int eps = e & 31;
    return ((eps & 15) >> 2) * PLANE + (e >> 5) * Q4M_PB + (eps >> 4) * 8 + ((((eps & 3) >> 1) | ((eps & 1) << 1)) * 2);
}

// (x0, x1) -> packed f16 pairs hi, lo (both rounded toward zero; x - hi is exact in f32)
__device__ __forceinline__ void q4m_split2(float x0, float x1, unsigned& hi, unsigned& lo) {
    typedef __fp16 h2 __attribute__((ext_vector_type(2)));
    union { h2 h; unsigned u; } H, L;
    H.h = __builtin_amdgcn_cvt_pkrtz(x0, x1);
    float d0, d1;
    asm("v_fma_mix_f32 %0, %2, -1.0, %3 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %1, %2, -1.0, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(d0), "=&v"(d1) : "v"(H.u), "v"(x0), "v"(x1));
    L.h = __builtin_amdgcn_cvt_pkrtz(d0, d1);
    hi = H.u; lo = L.u;
}

// elements e, e + 1 (e even) of the vector: four 2-byte stores at p = q4m_pair_ptr(img, e), scaled by q4m_pair_scale(e).
// (128 elements further = 4 blocks = p + 128: a gather's loads differ by an immediate offset.)
template <int PLANE>
__device__ __forceinline__ char* q4m_pair_ptr(char* img, int e) { return img + q4m_elem_off<PLANE>(e); }
__device__ __forceinline__ float q4m_pair_scale(int e) { return (e & 16) ? 0.0625f : 1.0f; }
__device__ __forceinline__ void q4m_put2_at(char* p, float sc, float x0, float x1) {
    unsigned hi, lo;
    q4m_split2(x0 * sc, x1 * sc, hi, lo);
    *reinterpret_cast<unsigned short*>(p) = (unsigned short)hi;       // element e; e + 1 sits 4 bytes further (same pair register)
    *reinterpret_cast<unsigned short*>(p + 4) = (unsigned short)(hi >> 16);
    *reinterpret_cast<unsigned short*>(p + 16) = (unsigned short)lo;
    *reinterpret_cast<unsigned short*>(p + 20) = (unsigned short)(lo >> 16);
}
template <int PLANE>
__device__ __forceinline__ void q4m_put2(char* img, int e, float x0, float x1) { q4m_put2_at(q4m_pair_ptr<PLANE>(img, e), q4m_pair_scale(e), x0, x1); }
// elements e .. e + 3 (e % 4 == 0): one 8-byte store per piece
template <int PLANE>
__device__ __forceinline__ void q4m_put4(char* img, int e, float4 x) {
    const float sc = q4m_pair_scale(e);
    unsigned h02, l02, h13, l13;
    q4m_split2(x.x * sc, x.z * sc, h02, l02);                         // k order: 4j, 4j+2, 4j+1, 4j+3
    q4m_split2(x.y * sc, x.w * sc, h13, l13);
    char* p = img + q4m_elem_off<PLANE>(e);
    *reinterpret_cast<q4m_u2*>(p) = (q4m_u2){h02, h13};
    *reinterpret_cast<q4m_u2*>(p + 16) = (q4m_u2){l02, l13};
}

// this lane's piece of a block (xp = img + block * 32 + piece * 16): one 16-byte read per plane -> 16 registers
template <int PLANE>
__device__ __forceinline__ void q4m_xload(uint4 (&xv)[4], const char* xp) {
#pragma unroll
    for (int j = 0; j < 4; ++j) xv[j] = *reinterpret_cast<const uint4*>(xp + j * PLANE);
}
// one block of one row against that piece: returns 2^-24 sum n x summed over the hi (register 0) and lo (register 1) pieces --
// valid in every lane, for the row whose nibbles the lane holds
__device__ __forceinline__ float q4m_block(const uint4& q, const uint4 (&xv)[4]) {
    const unsigned w[4] = {q.x, q.y, q.z, q.w};
    q4m_f4 D = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned l0 = w[j] & 0x000f000fu, h0 = w[j] & 0x00f000f0u, s = w[j] >> 8, l1 = s & 0x000f000fu, h1 = s & 0x00f000f0u;
        D = __builtin_amdgcn_mfma_f32_4x4x4f16(__builtin_bit_cast(q4m_h4, (q4m_u2){xv[j].x, xv[j].y}), __builtin_bit_cast(q4m_h4, (q4m_u2){l0, l1}), D, 0, 0, 0);
        D = __builtin_amdgcn_mfma_f32_4x4x4f16(__builtin_bit_cast(q4m_h4, (q4m_u2){xv[j].z, xv[j].w}), __builtin_bit_cast(q4m_h4, (q4m_u2){h0, h1}), D, 0, 0, 0);
    }
    return D[0] + D[1];
}
template <int PLANE>
__device__ __forceinline__ float q4m_block(const uint4& q, const char* xp) {
    uint4 xv[4];
    q4m_xload<PLANE>(xv, xp);
    return q4m_block(q, xv);
}

}  // namespace llmk
