// q4_0 dots of the persistent kernel on the matrix core: 16-row UNITS on v_mfma_f32_16x16x32_f16 (round 5).
//
// Replaces the dot loops of /root/reference/llama2.f90:529-531, 603-605, 610-612, 618-620, 634-636 on q4_0 weights (the
// four_bit_dev branch: not under /root/reference; ggml's public block format -- 32 weights = one f16 scale d + 16 bytes, low
// nibbles elements 0..15, high nibbles elements 16..31, value (nibble - 8) d).
//
// Why: round 4 read the q4_0 phases as VALU-bound: 13 full-rate operations per dword of nibbles, ~455 per 4-row tile and wave.
// Here a dword costs 5 bit operations and ONE matrix instruction, the block scales and the "-8 sum x" term go through the
// matrix core as well, and -- because x then lives in LDS, not in 64 registers per lane -- all EIGHT waves of a CU stream weights
// (the service wave had no registers for a fragment).  Probed in round 4 (csrc/probes/q4_mfma16_probe.hip: 1.58x on equal work,
// 5.5e-6 against a double-precision dot).  What it bought in the kernel (profiles/r05_llama2-7b_q4_0_pmc_sq_{before,after}.txt, DESIGN.md
// section 3d): plain VALU instructions -27 %, the token +2 % -- the slots wait for the memory pipeline, not for the VALU.
//
// UNIT = 16 rows x 32 blocks (1024 columns): 8 KB of nibbles + 1 KB of scales, 9,216 contiguous bytes, the bytes of a round-4
// tile (4 rows x 128 blocks).  Device layout of a matrix [R][K]: units in (row group, column slot) order,
//     unit (rg, cs) at ((rg * NCS + cs) * 9216),   NCS = ceil(K / 1024)
//     bytes [0, 8192):    W[j 0..7][lane 0..63] 16 bytes: lane (m = lane % 16, g = lane / 16), dwords g of the blocks
//                          32 cs + 4 j + (0..3) of row 16 rg + m  -- a lane's dword IS a matrix operand's k group g of a block
//     bytes [8192, 9216): S[lane] 16 bytes: the f16 scales of row 16 rg + m, blocks 32 cs + 8 g + (0..7)
// blocks past the end of a row (K = 11008: the last unit holds 24 blocks) are zero nibbles with zero scales.
// q16_units_kernel (llmk.hip) builds it from the row layout the other kernels read; both stay resident.
//
// The matrix instruction D[16][16] += A[16][32] B[32][16], lane (i = lane % 16, g = lane / 16): A[i][8 g ..], B[8 g ..][i] as 8
// halves, D[4 g + (0..3)][i]:
//   * A = one block of 16 rows.  A nibble where it lies in a 16-bit half -- bits 0-3, or bits 4-7 -- is the f16 subnormal
//     n 2^-24 (16 n 2^-24); the instruction honours subnormals (probes/mfma_f16_denorm_probe).  q & 0x000f000f,
//     q & 0x00f000f0 and the same masks on q >> 8 ARE the four operand registers: k order (4g, 4g+2, 16+4g, 16+4g+2, 4g+1,
//     4g+3, 16+4g+1, 16+4g+3) of the block's 32 elements.
//   * B = x as two f16 pieces hi + lo (|x - hi - lo| <= 2^-22 |x|; every product exact in f32) in the same k order, the
//     high-nibble elements divided by 16, from an IMAGE in LDS the gathers write (128 bytes per block: hi | lo).  Block
//     b = 8 c + t of a group of eight puts its pieces into COLUMNS 2 t, 2 t + 1 and zeros into the other fourteen (a lane
//     reads the image where lane % 16 / 2 == t and a line of zeros otherwise): eight blocks accumulate into ONE accumulator set,
//     each in its own two columns, with no arithmetic in between.
//   * the group's 16 x 8 scales reach those columns through one more instruction: the rows' scale plane (k groups other than
//     c zeroed) times the constant selector SEL[k][n] = (k % 8 == n / 2) is d[row][8 c + n / 2] in column n; four v_fma_f32
//     then scale eight blocks.
//   * -8 d sum(x): the scale plane times the blocks' sums of x (hi | lo in columns 0 | 1): one instruction per unit.
// Per unit and wave: 37 matrix instructions and ~230 VALU operations for 512 (row, block) pairs (round 4: ~455 + 0).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

#include "kernels.h"

namespace llmk {

constexpr int Q16_ROWS = 16, Q16_BLOCKS = 32, Q16_UNIT_BYTES = 9216, Q16_W_BYTES = 8192;
constexpr int Q16_IMG_BLK = 128;          // image bytes per block: 4 k groups x 8 halves of hi, then of lo
constexpr int Q16_ZERO_BYTES = 4096;      // a column that holds no block reads here: group offsets up to 3 x 1 KB + 16
constexpr float Q16_RESCALE = 16777216.0f;
// the blocks' sums of x sit in the image as sum / 32: |sum / 32| <= the block's largest |element|, so whatever range the elements
// fit (the gathers' power-of-two scaling, token_kernel.h) the sums fit too; the -8 d sum(x) term is then -256 d (sum / 32)
constexpr float Q16_SUM_DIV = 32.0f;

typedef _Float16 q16_v8h __attribute__((ext_vector_type(8)));
typedef float q16_v4f __attribute__((ext_vector_type(4)));
typedef unsigned q16_v4u __attribute__((ext_vector_type(4)));

__host__ __device__ constexpr int q16_ncs(int K) { return (K + 1023) / 1024; }
__host__ __device__ constexpr size_t q16_bytes(size_t rows, int K) { return rows / Q16_ROWS * q16_ncs(K) * Q16_UNIT_BYTES; }

// LDS image of a vector of up to NBI blocks: | image NBI x 128 | sums hi NBI halves | sums lo NBI halves | zeros |
template <int NBI>
struct Q16Img {
    static constexpr int SUM = NBI * Q16_IMG_BLK, SUM_LO = SUM + NBI * 2, ZERO = SUM + NBI * 4, BYTES = ZERO + Q16_ZERO_BYTES;
    static_assert(NBI % 32 == 0 && (NBI * 2) % 16 == 0, "whole units; 16-byte aligned planes");
};

__device__ __forceinline__ q16_v8h q16_as_v8h(const q16_v4u& u) { return __builtin_bit_cast(q16_v8h, u); }
__device__ __forceinline__ q16_v4u q16_lds16(const char* p) { return *reinterpret_cast<const q16_v4u*>(p); }
// the four operand registers of one dword of nibbles: 5 VALU operations
__device__ __forceinline__ q16_v8h q16_unpack(unsigned q) {
    const unsigned s = q >> 8;
    const q16_v4u u = {q & 0x000f000fu, q & 0x00f000f0u, s & 0x000f000fu, s & 0x00f000f0u};
    return q16_as_v8h(u);
}

// ---- writing the image (the gathers; layer 0: the staging of the embedding row) -------------------------------------------
__device__ __forceinline__ unsigned short q16_bits(_Float16 h) { return __builtin_bit_cast(unsigned short, h); }
// where elements e0, e0 + 1 (e0 even) go: bytes from the image's start to the hi piece of e0; e0 + 1 lies 8 bytes further, the
// lo pieces 64 bytes further; 128 elements on (the next load of a gather) is 512 bytes on
__device__ __forceinline__ int q16_pair_off(int e0) {
    const int el = e0 & 31;
    return (e0 >> 5) * Q16_IMG_BLK + ((el & 15) >> 2) * 16 + (((el & 3) >> 1) + 2 * (el >> 4)) * 2;
}
__device__ __forceinline__ float q16_pair_scale(int e0) { return (e0 & 16) ? 0.0625f : 1.0f; }
// (the gathers sit on the layer's critical path: both hi pieces in one v_cvt_pkrtz_f16_f32 -- toward zero, so the remainder has the
// sign of x and at most 11 significant bits --, x - hi through v_fma_mix_f32 with the half as a source (no conversion back), both lo
// pieces in one more: |x - hi - lo| <= 2^-21 |x|, 4 operations for the pair instead of 8)
__device__ __forceinline__ void q16_put2(char* p, float sc, float x0, float x1) {
    const float a = x0 * sc, b = x1 * sc;                 // (a power of two: exact)
    typedef __fp16 q16_h2 __attribute__((ext_vector_type(2)));
    union { q16_h2 h; unsigned u; } H, L;
    H.h = __builtin_amdgcn_cvt_pkrtz(a, b);
    float ra, rb;
    asm("v_fma_mix_f32 %0, %2, -1.0, %3 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %1, %2, -1.0, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=&v"(ra), "=&v"(rb) : "v"(H.u), "v"(a), "v"(b));
    L.h = __builtin_amdgcn_cvt_pkrtz(ra, rb);
    *reinterpret_cast<unsigned short*>(p) = (unsigned short)H.u;
    *reinterpret_cast<unsigned short*>(p + 8) = (unsigned short)(H.u >> 16);
    *reinterpret_cast<unsigned short*>(p + 64) = (unsigned short)L.u;
    *reinterpret_cast<unsigned short*>(p + 72) = (unsigned short)(L.u >> 16);
}
// four consecutive elements e .. e + 3 (e % 4 == 0): elements 0, 2 are neighbours in the image, and so are 1, 3
__device__ __forceinline__ void q16_put4(char* img, int e, const float4& x) {
    const float sc = q16_pair_scale(e);
    char* p = img + q16_pair_off(e);
    const float v[4] = {x.x * sc, x.z * sc, x.y * sc, x.w * sc};
    unsigned hi[2], lo[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const _Float16 h0 = (_Float16)v[2 * i], h1 = (_Float16)v[2 * i + 1];
        const _Float16 l0 = (_Float16)(v[2 * i] - (float)h0), l1 = (_Float16)(v[2 * i + 1] - (float)h1);
        hi[i] = (unsigned)q16_bits(h0) | ((unsigned)q16_bits(h1) << 16);
        lo[i] = (unsigned)q16_bits(l0) | ((unsigned)q16_bits(l1) << 16);
    }
    *reinterpret_cast<unsigned*>(p) = hi[0];
    *reinterpret_cast<unsigned*>(p + 8) = hi[1];
    *reinterpret_cast<unsigned*>(p + 64) = lo[0];
    *reinterpret_cast<unsigned*>(p + 72) = lo[1];
}
// the sum of a block's 32 activations as hi | lo halves (p = the block's slot in the hi plane, NBI halves per plane)
template <int NBI>
__device__ __forceinline__ void q16_put_sum(char* p, float s) {
    const _Float16 h = (_Float16)s;
    *reinterpret_cast<unsigned short*>(p) = q16_bits(h);
    *reinterpret_cast<unsigned short*>(p + NBI * 2) = q16_bits((_Float16)(s - (float)h));
}

// ---- one unit against the image --------------------------------------------------------------------------------------------
struct Q16Lane {
    const char* xa[8];   // B operand of block t = 0..7 of a group: the image (block 0 of the unit's first group) or the zeros
    const char* sa;      // B operand of the -8 sum(x) instruction
};
// img = the LDS image (Q16Img<NBI>), cs = the unit's column slot
template <int NBI>
__device__ __forceinline__ void q16_lane(Q16Lane& ln, const char* img, int cs, int lane) {
    const int n = lane & 15, g = lane >> 4;
    const char* zero = img + Q16Img<NBI>::ZERO;
    const char* mine = img + cs * (Q16_BLOCKS * Q16_IMG_BLK) + (n & 1) * 64 + g * 16;
#pragma unroll
    for (int t = 0; t < 8; ++t) ln.xa[t] = (n >> 1) == t ? mine + t * Q16_IMG_BLK : zero;
    ln.sa = n < 2 ? img + Q16Img<NBI>::SUM + n * (NBI * 2) + (cs * Q16_BLOCKS + 8 * g) * 2 : zero;
}
// SEL[k][n] = (k % 8 == n / 2) as this lane's B operand (k = 8 g .. 8 g + 7: the same for every g): half (n / 2) is 1.0
__device__ __forceinline__ q16_v8h q16_sel(int lane) {
    const int n = lane & 15;
    const unsigned one = (n & 2) ? 0x3C000000u : 0x00003C00u;
    q16_v4u s = {0u, 0u, 0u, 0u};
    if ((n >> 2) == 0) s.x = one; else if ((n >> 2) == 1) s.y = one; else if ((n >> 2) == 2) s.z = one; else s.w = one;
    return q16_as_v8h(s);
}
// w = the lane's 8 x 16 bytes of the unit's nibbles, sc = its 16 bytes of scales.
// acc (lane (n, g), component j) += d[4 g + j][b] * 2^-24 sum n x of the block b = 8 c + n / 2 whose piece n % 2 sits in column n;
// acc8 += the scale plane times the sums of x.  The B operands of the next HALF group (4 reads, 16 registers) are requested
// before the current half group is multiplied; left to itself hipcc reads each operand one instruction ahead and the LDS latency
// shows (probe: 2,634 against 2,095 clocks per unit).
// mid(integral_constant<P>), P = 0, 1, 2: called behind half groups 1, 3 and 5 (the caller's request of its next unit, in three pieces)
template <class F>
__device__ __forceinline__ void q16_unit_dot(const float4 (&w)[9], const Q16Lane& ln, const q16_v8h& sel, int g, q16_v4f& acc, q16_v4f& acc8, F mid) {
    const float4& sc4 = w[8];
    const q16_v4f z = {0.f, 0.f, 0.f, 0.f};
    const q16_v4u zu = {0u, 0u, 0u, 0u};
    const q16_v4u sc = {__float_as_uint(sc4.x), __float_as_uint(sc4.y), __float_as_uint(sc4.z), __float_as_uint(sc4.w)};
    q16_v4u bx[2][4];
#pragma unroll
    for (int t = 0; t < 4; ++t) bx[0][t] = q16_lds16(ln.xa[t]);
    acc8 = __builtin_amdgcn_mfma_f32_16x16x32_f16(q16_as_v8h(sc), q16_as_v8h(q16_lds16(ln.sa)), acc8, 0, 0, 0);
    q16_v4f s = z, d = z;
#pragma unroll
    for (int h = 0; h < 8; ++h) {                  // half group h: blocks 4 h .. 4 h + 3 of the unit = the lane's w[h]
        const int c = h >> 1;
        if (h < 7) {
#pragma unroll
            for (int t = 0; t < 4; ++t) bx[(h + 1) & 1][t] = q16_lds16(ln.xa[4 * ((h + 1) & 1) + t] + ((h + 1) >> 1) * (8 * Q16_IMG_BLK));
        }
        __builtin_amdgcn_sched_barrier(0);
        if ((h & 1) == 0) {
            const q16_v4u scv = g == c ? sc : zu;                                  // the group's eight scales, the other k groups zeroed
            d = __builtin_amdgcn_mfma_f32_16x16x32_f16(q16_as_v8h(scv), sel, z, 0, 0, 0);   // d[4 g + j][8 c + n / 2] in column n
            s = z;
        }
        const unsigned q[4] = {__float_as_uint(w[h].x), __float_as_uint(w[h].y), __float_as_uint(w[h].z), __float_as_uint(w[h].w)};
#pragma unroll
        for (int t = 0; t < 4; ++t) s = __builtin_amdgcn_mfma_f32_16x16x32_f16(q16_unpack(q[t]), q16_as_v8h(bx[h & 1][t]), s, 0, 0, 0);
        if (h & 1) {
            acc.x = fmaf(d.x, s.x, acc.x); acc.y = fmaf(d.y, s.y, acc.y); acc.z = fmaf(d.z, s.z, acc.z); acc.w = fmaf(d.w, s.w, acc.w);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (h == 1) mid(std::integral_constant<int, 0>());       // (h is a constant of the unrolled loop)
        if (h == 3) mid(std::integral_constant<int, 1>());
        if (h == 5) mid(std::integral_constant<int, 2>());
    }
}
// the sixteen columns of a row added up: every lane of the 16-lane row gets the sum
__device__ __forceinline__ float q16_row_sum(float v) {
    v += dpp_mov<0xB1, 0xf, true>(0.f, v);      // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E, 0xf, true>(0.f, v);      // quad_perm [2,3,0,1]
    v += dpp_mov<0x141, 0xf, true>(0.f, v);     // row_half_mirror
    v += dpp_mov<0x140, 0xf, true>(0.f, v);     // row_mirror
    return v;
}
// the unit's 16 row sums sum_b d (sum n x - 8 sum x) -> part[0..15] (row 4 g + j by lane (n = j, g))
__device__ __forceinline__ void q16_unit_store(const q16_v4f& acc, const q16_v4f& acc8, float* part, int lane) {
    constexpr float M8 = -8.0f * Q16_SUM_DIV;
    const float y0 = q16_row_sum(fmaf(acc.x, Q16_RESCALE, M8 * acc8.x)), y1 = q16_row_sum(fmaf(acc.y, Q16_RESCALE, M8 * acc8.y));
    const float y2 = q16_row_sum(fmaf(acc.z, Q16_RESCALE, M8 * acc8.z)), y3 = q16_row_sum(fmaf(acc.w, Q16_RESCALE, M8 * acc8.w));
    const int n = lane & 15;
    const float y = n == 0 ? y0 : n == 1 ? y1 : n == 2 ? y2 : y3;
    if (n < 4) part[4 * (lane >> 4) + n] = y;
}

// ---- building the unit layout from the row layout (llmk.hip: at the first token after an upload) -------------------------
// rows: [nrows][row_stride] -- K / 2 nibble bytes (16 per block), then the row's K / 32 f16 scales; one wave per unit
__global__ __launch_bounds__(64) void q16_units_kernel(const char* __restrict__ rows, size_t row_stride, int K, int ncs, char* __restrict__ units) {
    const int lane = threadIdx.x, m = lane & 15, g = lane >> 4;
    const size_t unit = blockIdx.x;
    const int cs = (int)(unit % (size_t)ncs);
    const size_t rg = unit / (size_t)ncs;
    const char* row = rows + (rg * Q16_ROWS + m) * row_stride;
    const int nblk = K / 32;
    char* out = units + unit * Q16_UNIT_BYTES;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        q16_v4u v;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int b = cs * Q16_BLOCKS + 4 * j + i;
            v[i] = b < nblk ? *reinterpret_cast<const unsigned*>(row + (size_t)b * 16 + 4 * g) : 0u;
        }
        reinterpret_cast<q16_v4u*>(out)[j * 64 + lane] = v;
    }
    q16_v4u s;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int b = cs * Q16_BLOCKS + 8 * g + 2 * i;
        const unsigned short* sp = reinterpret_cast<const unsigned short*>(row + K / 2);
        const unsigned lo = b < nblk ? sp[b] : 0u, hi = b + 1 < nblk ? sp[b + 1] : 0u;
        s[i] = lo | (hi << 16);
    }
    reinterpret_cast<q16_v4u*>(out + Q16_W_BYTES)[lane] = s;
}

}  // namespace llmk
