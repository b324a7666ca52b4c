// Persistent whole-token kernel for gfx950: the complete `transformer` pass
// (/root/reference/llama2.f90:480-640) in ONE launch, one 8-wave workgroup per CU.
//
// Why: with one kernel per GEMV the token is bounded by ~1.55 us of boundary + ramp per launch
// (5 launches per layer; DESIGN.md / HISTORY.md section 3) -- the weight stream stops at every dependency.  Weights
// do not depend on activations, so here the stream never stops: each of the 7 STREAMING waves of a CU
// walks a static list of 8 KB row tiles (its share of qkv, wo, w1|w3, w2 of every layer, then the
// classifier) and always has NB (TkShape::NB) tiles requested ahead in registers (non-temporal 16-byte loads),
// across phase and layer boundaries.  The one SERVICE wave per CU never touches the weight stream
// (so its polls are not queued behind 40 KB of outstanding loads -- vmcnt retires in order): it
// gathers the phase's input vector from the exchange buffers into LDS, applies rmsnorm, and after
// the streaming waves have dotted their tiles against it, runs the epilogue (RoPE / SwiGLU /
// residual) and publishes the CU's outputs.
//
// Exchange = 8-byte {value, epoch-tag} granules written with ONE agent-scope (sc1, write-through)
// store and swept with agent-scope loads until every tag matches: no flags, no fences, placement
// independent (cdna_hip_programming.md Guideline 16, form R2).  Epochs are unique per
// (token serial, phase), so buffers are never reset.  Every spin is bounded; a timeout raises a
// sticky error word and lets the kernel drain.
//
// Work split per phase: CU c owns a contiguous row range of the phase's matrix (w1|w3, w2 and the
// classifier: R/256 rows each; QKV and wo: split over the 224 CUs that do NOT run attention, see
// TkShape); tile t of that range goes to streaming wave t % 7.  A tile is one row x up to 2048
// columns; its wave-reduced partial dot goes to LDS and the service wave folds the parts of a row.
// The x fragment a wave dots its tiles against is the same for every tile of a phase and lives in
// registers (TkX).  DESIGN.md / HISTORY.md section 3b has the measurements behind each of these choices.
//
// q4_0 matrices (round 5) take the same kernel with another kind of tile: a UNIT of 16 rows x 32 blocks in its own device layout,
// dotted on the matrix core against an f16 image of x in LDS (q4_units.h), all EIGHT waves taking units and gathering an eighth
// of every input vector each (TkShape::COOP, tk_step, tk_stream_coop); DESIGN.md / HISTORY.md section 3d.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "kernels.h"
#include "q4_units.h"
#include "q6k.h"

// ---- measurement knobs (defaults = the product; -D... builds a variant library for an A/B) ---------------------------
// loads per lane the service wave keeps in flight in one pass of the hb gather (no gains are held across it).  H = 5632 is
// 44 loads per lane: 32 -> two pieces of 22.  Round-3 sweep (profiles/r03_gather_piece_sweep.jsonl, kernel us, f16 / f32):
// one piece of 44: 570 / 723; 22+22: 556 / 719; 15+15+14: 555 / 717 (spills 20 bytes in the f32 kernel); 4 x 11: 553 / 738;
// 6 x 8: 561 / 743; E-vectors in pieces of 8: 563 / 724.  Nothing beats two pieces by more than the box-to-box noise.
// Round 6, with the sweeps led by "gather first" (a first pass practically never fails now: passes mean 1.0 in the traces): ONE piece
// of 44 for f32 -- 1,470.7-1,482.6 against 1,447.4-1,471.7 tok/s, four interleaved rounds on one box (profiles/r06_ab.txt) -- while f16
// keeps two (2,046-2,061 against 2,074-2,105 with one).  -1 = these per-type defaults (TkShape::HB_NL).
#ifndef LLMK_TK_HB_NL
#define LLMK_TK_HB_NL -1
#endif
// the same for the E-vectors x / xa (rmsnorm gains are held across these) and xb
#ifndef LLMK_TK_E_NL
#define LLMK_TK_E_NL 24
#endif
#ifndef LLMK_TK_XB_NL
#define LLMK_TK_XB_NL 32
#endif
// s_sleep argument between two polling passes of a gather (units of 64 clocks)
#ifndef LLMK_TK_POLL_SLEEP
#define LLMK_TK_POLL_SLEEP 1
#endif
// "Gather first" (f32 / f16 kernels).  A CU's memory pipeline is a FIFO: the refill burst a phase ends with (up to NB tiles
// x 7 waves = 224-280 KB) is issued right after barrier B, the service wave's gather of the NEXT phase's input a moment
// later -- and waits behind all of it (round-3 traces with every inter-CU wait disabled: the hb sweep alone takes 5.7 us
// of an 18 us f16 layer).  With LLMK_TK_GF the streaming waves issue only LLMK_TK_GF_NOW tiles each at once and the rest of
// the burst when the service wave has ISSUED (not completed) the first pass of that gather: the sweep then leads the
// burst through the pipeline, the dots run on tiles that are already resident while the burst streams.
// LLMK_TK_GF_DELAY: s_sleep units before the first pass of the hb / x / xa gathers (a pass issued before the last producer
// has published fails and the retry queues behind the released burst, as before).
#ifndef LLMK_TK_GF
#define LLMK_TK_GF 1
#endif
// Round-3 sweeps on TinyLlama (profiles/r03_gather_first_sweep.jsonl; kernel us at KV length 256, f16 / f32; off: 555 / 720):
//   tiles per wave issued at once 1, delay 0: 539 / 727;  delay 16: 531 / 705;  40: 516 / 689-696;  48: 515 / 694;
//   64: 517 / 701;  96: 567 / 747;  127: 620 / 797;   2 tiles, delay 48-64: 516-526 / 683-687;   3 tiles, 64: 523 / 684;
//   0 tiles, delay 16: 544 / 735;   per-gather delays (hb 24 / 80, x 16 / 80, xa 16 around a common 48): all within +-1 % or worse.
//   burst released by the LAST piece's first pass instead of the first piece's (hb is two pieces: with the first piece's pass
//   the second one queued behind the burst again, 2.1 us instead of 1.0): f16 1 tile 533, 2 tiles 509, 3 tiles delay 36 / 48 /
//   64: 503-506 / 505-509 / 516, 4 tiles 507; f32 2 / 3 tiles 683 (first piece: 679).
// => f16: last piece, 3 tiles, delay 36; f32: first piece, 2 tiles, delay 56.  (-1 = these per-type defaults.)
// The q4_0 kernel has no bursts (its tiles are requested from inside the dots); leaving the LAST slot's request of a phase to
// the gather that follows, behind its first pass's loads, was measured too: 783 -> 759 tok/s on Llama-2-7B, not kept.
#ifndef LLMK_TK_GF_NOW
#define LLMK_TK_GF_NOW -1
#endif
// q4_0 kernel: s_sleep units every wave lets the producers have before the first pass of its x / xa / hb slice: the two
// tiles each wave still has in flight drain meanwhile and the pass, now likely to succeed, finds a short queue.
// Round 3 (profiles/r03_gather_first_sweep.jsonl run "cd", Llama-2-7B q4_0 kernel us): 0: 1,316-1,320; 12 / 24 / 40: 1,236-1,239;
// 56: 1,252; 80: 1,298; 110: 1,373.  790 -> 843 tok/s.
#ifndef LLMK_TK_COOP_DELAY
#define LLMK_TK_COOP_DELAY 24
#endif
// 1: even the first tiles of a burst wait until the service wave has ISSUED the phase's publish stores (second LDS word)
#ifndef LLMK_TK_GF_PUB
#define LLMK_TK_GF_PUB -1           // -1 = per type (TkShape::GF_PUB): f16 yes, f32 no
#endif
#ifndef LLMK_TK_GF_LAST          // which pass releases the burst: the first piece's (0) or the last piece's (1); -1 = per type
#define LLMK_TK_GF_LAST -1
#endif
#ifndef LLMK_TK_GF_DELAY
#define LLMK_TK_GF_DELAY -1
#endif
// (round 6, measured and removed: the residual-stream gathers staging x * gains in the pass that delivers them -- tk_coop_part on the
// service wave instead of tk_gather + TkNorm::apply: f16 2,005 against 2,090 tok/s, f32 1,433 against 1,465; csrc/variants/
// fused_norm_gather.patch, profiles/r06_ab.txt)
#ifndef LLMK_TK_ATT_SPLIT
#define LLMK_TK_ATT_SPLIT 1          // contexts longer than 256 timesteps: a head's attention in parts on the CUs of its group (TkAttPlan)
#endif
// f32 / f16 kernels, round 4: slots at the START of the w1|w3 (A) phase that are refilled inside the phase, right after
// they are consumed, instead of in the burst behind barrier B.  The tiles they request are the NEXT phase's first (w2's,
// the next layer's QKV / wo), so they get a head start of a phase, and the burst that the following sweep has to share the
// CU's memory pipeline with is that much shorter.  (All refills were late since round 1, when a refill issued inside a
// phase blocked on a pipeline full of earlier prefetch; with the sweeps ordered ahead of the bursts -- LLMK_TK_GF -- the
// pipeline is all but empty when a phase starts.)
// Round-4 sweep on TinyLlama f16 (profiles/r04_f16_early_refill.jsonl; tok/s, same box, base 1,942-1,955 = none early, 3 at once):
//   1 early / 2 at once: 2,017;   1 / 1: 1,943;   2 / 2: 1,988;   2 / 1: 1,971;   2 / 0: 1,878;   2 early in A + 1 in D / 1: 1,904.
// => f16: one slot of the w1|w3 phase early, two tiles of the burst at once (-1 = these per-type defaults; f32 keeps none early).
#ifndef LLMK_TK_EARLY_A
#define LLMK_TK_EARLY_A -1
#endif
// q4_0 kernel, round 4.  Its phases are bound by how fast the CU's memory pipeline ACCEPTS the tile requests issued from
// inside the dots (one request per consumed tile: 57 KB per slot and CU at the CU's 25 KB/us share of HBM = the 2.3 us a slot
// takes, whatever the ALU needs -- an attention CU, whose QKV slots are empty, spends the same 4.5 us in them), while in the
// windows between phases nothing is requested at all.  During the attention hop (7.4 us at 7B: 224 CUs wait for 32) the
// ring entry the QKV phase's last slot has just consumed is FREE: the request the wo slot would issue (tile KO + NB - 1, the
// second w1|w3 tile) is issued there instead, right behind the QKV phase (attention CUs and attention parts: right behind
// their attention), and the wo phase's first slot issues none.  1 = on.
#ifndef LLMK_TK_ADV_HOP
#define LLMK_TK_ADV_HOP 1
#endif
// (round 5, measured and removed: the QKV slots' requests issued in the hop too -- three units per wave at its start: kernel
// 1,111 -> 1,143 us, and worse with a pause before the first xb poll (1,154 / 1,195 us for 0.85 / 2.6 us): 216 KB per CU keep
// HBM saturated for 8.6 us, the attention CUs' K/V rows and q polls queue in it, and xb arrives later everywhere;
// profiles/r05_ab.jsonl "hopq".  One unit per wave is what the hop takes.)

// ---- experiment switches: MEASUREMENT BUILDS ONLY (tests/host_tools/build_variant.sh NAME -DLLMK_EXP_...; never in libllmk.so) ----
// Wrong results by construction, only the kernel's time is read (profiles/r05_slot_decomposition.txt, r06_f16_stream_decomposition.txt):
//   LLMK_EXP_NOHBM  every tile / unit request reads one line of zeros (same instructions, same counts): the exchange chain and
//                   the dots without the weight stream;
//   LLMK_EXP_NODOT  f32 / f16: tiles are requested and waited for, not multiplied: the stream and the chain without the dots.
// Combined with the debug library's LLMK_TK_NOSYNC=1 (nothing waits for a tag: the stream and the dots without the chain).
#if defined(LLMK_EXP_NOHBM)
constexpr bool TK_EXP_NOHBM = true;
#else
constexpr bool TK_EXP_NOHBM = false;
#endif
#if defined(LLMK_EXP_NODOT)
constexpr bool TK_EXP_NODOT = true;
#else
constexpr bool TK_EXP_NODOT = false;
#endif

namespace llmk {

constexpr int TK_NCU = 256;               // one workgroup per CU
constexpr int TK_WAVES = 8;               // 7 streaming + 1 service (2 waves/SIMD: 256 VGPRs each)
// register tiles a streaming wave keeps requested ahead: TkShape::NB (5 x 8 KB for f32, 4 for f16; q4_0: 4 units of 9 KB)
constexpr int TK_NS = TK_WAVES - 1;
constexpr int TK_THREADS = TK_WAVES * WAVE;
constexpr int TK_TCOLS = 8;               // 16-byte vector columns per tile (8 x 64 lanes x 4 floats = 2048)
constexpr unsigned TK_SPIN_LIMIT = 1u << 22;
constexpr int TK_TRACE_N = 16 * 64;       // stamps per CU: 16 per layer, first 64 layers
// Timing aids (wall-clock stamps of the service wave, "do not wait for tags") exist only in the debug build
// (make -C llm.f90_amd debug -> libllmk_debug.so, -DLLMK_TK_DEBUG): in the product library every use below folds away.
#ifdef LLMK_TK_DEBUG
constexpr bool TK_DEBUG = true;
#else
constexpr bool TK_DEBUG = false;
#endif

struct TokenArgs {
    const float* emb;        // [V][E]
    const float* rms;        // rmsnorm gains, ONE allocation: att [L][E] | ffn [L][E] | final [E]  (tk_rms_att ...: see g_qkv)
    const void* wqkv;        // [L][E+2KV][E]   f32 or f16 rows (TkShape::WT)
    const void* wo;          // [L][E][E]
    const void* w13;         // [L][2H][E]
    const void* w2;          // [L][E][H]
    const void* wcls;        // [V][E]
    float* kc;               // [L][S][KV]
    float* vc;
    const float* rope;       // [hs/2]
    const int* tokpos;       // {token0, pos1, serial} in device memory (graph replay), or null: the three fields below
    int tok_imm, pos_imm, serial_imm;
    unsigned* herr;          // optional sticky error word in HOST memory (direct mode, where logits also point at host memory; the greedy pipeline: its ids sit behind it)
    // exchange granules, ONE allocation: qkv [E+2KV] | xb [E] (attention output) | xa [E] (x after the attention residual) |
    // hb [H] | x [E] (x after the FFN residual).  One pointer (tk_g_xb<SH>(a) ... add compile-time offsets) instead of five:
    // the kernel is at its SGPR ceiling and every pointer kept live across the layer loop is two more registers that hipcc
    // spills to VGPR lanes and reloads (v_readlane) inside the service wave's polling loops -- the token's critical path.
    unsigned long long* g_qkv;
    float* logits;           // [V]
    unsigned* err;           // sticky error word (0 = ok)
    const float4* zeros;     // [NCU*TK_WAVES] 1 KB blocks of zeros: what empty ring slots / ragged row ends read
#ifdef LLMK_TK_DEBUG
    unsigned long long* trace;  // debug build only: [NCU][TK_TRACE_N] wall-clock stamps of each CU's service wave
#endif
    int L, S;
    float eps;               // rmsnorm epsilon
    int gflags;              // TKG_* bits (pipelined greedy decode, host error word; debug build: TKG_NOSYNC)
    // filled in per workgroup by the kernel: this CU's rows of the QKV and wo matrices (none on an attention CU)
    int q0, qn, o0, on;
    int c0, cn;              // this CU's rows of the classifier
    // Pipelined greedy decode (llmk_decode_greedy): `token = maxloc(logits,DIM=1)` (llama2.f90:388) without a host round
    // trip.  Every CU leaves the first maximum of ITS classifier rows in cand_out[c] = {logit, 0-based row}; the NEXT launch
    // (ordered behind this one by the stream) starts by folding the 256 candidates of cand_in -- 2 KB, first maximum wins --
    // on every CU that needs the embedding row, so the token never leaves the device and no exchange is added.
    // The candidates live behind the error word (err + 4: two buffers of TK_NCU float2, alternating by launch parity) and
    // the resolved ids behind the host error word (herr + 4: ints) -- no further pointer arguments: the kernel is at its
    // SGPR ceiling, and three more pointers cost the f32 instantiation 36 bytes of scratch.
    // The flags share the word of the debug build's "do not wait" switch, and with TKG_CAND_IN the unused tok_imm carries
    // the index of the id to store: the argument block is exactly as large as before.
};
// What a q4_0 kernel's workgroup adds for itself (token_kernel fills it in; never part of the launch's argument block -- the
// f32 / f16 instantiations sit at the SGPR ceiling and their code is sensitive to that block's very layout): the kernel works on
// a TokenArgsQ4 and hands it on as the TokenArgs it is; tk_q4() reads the rest back.
struct TokenArgsQ4 : TokenArgs {
    int h0, hn;              // this CU's hidden units (w1|w3 gate / up row pairs; f32 / f16: c * H / 256, compile-time count)
    int nw;                  // waves of this CU that take units (8; 7 on an attention CU)
    int d0, dn;              // this CU's rows of w2 (f32 / f16: c * E / 256, compile-time count)
};
__device__ __forceinline__ const TokenArgsQ4& tk_q4(const TokenArgs& a) { return static_cast<const TokenArgsQ4&>(a); }
// A launch of the GR instantiation always leaves its candidates in buffer pos & 1 (pos is live through the whole kernel
// anyway; a flag tested after the classifier would be one more register held across the layer loop).
constexpr int TKG_GREEDY = 1;      // host side only: launch the GR instantiation
constexpr int TKG_CAND_IN = 4;     // the token is the fold of the previous position's buffer, (pos & 1) ^ 1 (else tok_imm / tokpos[0])
constexpr int TKG_ID = 8;
constexpr int TKG_NOSYNC = 32;     // debug build only: do not wait for exchange tags (wrong results; measures the pure streaming rate)
__device__ __forceinline__ float2* tk_cand(const TokenArgs& a, int buf) {
    return reinterpret_cast<float2*>(a.err + 4) + buf * TK_NCU;
}
// first-maximum-wins fold of {value, index-as-float-bits} pairs over a wave; every lane ends with the winner
__device__ __forceinline__ void tk_wave_argmax(float& v, int& i) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(v, o, 64);
        const int oi = __shfl_xor(i, o, 64);
        if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
    }
}

constexpr int tk_cdiv(int a, int b) { return (a + b - 1) / b; }
constexpr int tk_cmax(int a, int b) { return a > b ? a : b; }
// CLS_: the classifier's own row type when it differs from the matrices' -- WT_Q6_K with q4_0 matrices, what stock llama.cpp
// q4_0 files hold (q6k.h); -1 = the matrices' type
template <int E_, int H_, int NH_, int NKV_, int V_, int WT_ = WT_F32, int CLS_ = -1>
struct TkShape {
    static constexpr int E = E_, H = H_, NH = NH_, NKV = NKV_, V = V_, WT = WT_, CLS = CLS_ < 0 ? WT_ : CLS_;
    static constexpr bool CLSQ6 = CLS == WT_Q6_K;
    static_assert(CLS == WT || (WT == WT_Q4_0 && CLSQ6 && E_ % 256 == 0 && E_ <= 4096),
                  "a classifier of its own type: q6_K rows beside q4_0 matrices, one quad per lane");
    static constexpr int HS = E / NH, KV = NKV * HS, KVMUL = NH / NKV, QKV = E + 2 * KV;
    // ---- weight tiles.  A tile is TK_TCOLS (8) lane loads of 16 bytes = 8 "segments" of 1 KB.  f32: one row x 8
    // segments (2048 columns).  f16: a row of 2048 columns is 4 segments, so a tile is RPT = 2 consecutive rows x LPT = 4
    // segments -- the x fragment a lane needs (LPT segments x VPL columns) is 32 floats either way.
    // q4_0 (round 5): the ring entry is a UNIT of 16 rows x 32 blocks in its own device layout (q4_units.h: 8 KB of nibbles as
    // matrix-core operands + 1 KB of scales, 9,216 contiguous bytes), dotted on v_mfma_f32_16x16x32_f16 against an f16 hi | lo
    // image of x in LDS.  A K = E row group is NCS_E units wide, a K = H one NCS_H (the last one ragged: zero blocks).
    static constexpr bool Q4 = WT == WT_Q4_0;
    static constexpr int NCS_E = q16_ncs(E), NCS_H = q16_ncs(H);
    static constexpr int NBI = (NCS_H > NCS_E ? NCS_H : NCS_E) * Q16_BLOCKS;   // blocks of the x image in LDS
    static constexpr int VPL = Q4 ? 32 : (WT == WT_F16 ? 8 : 4);         // weights per 16-byte lane load
    static constexpr int SEGW = WAVE * VPL;                              // weights per 1 KB segment
    // bytes between rows, K = E / K = H (q4_0 device row: K/2 nibble bytes, then the row's K/32 f16 scales, 16-byte aligned)
    // (the scale area is zero-padded to whole groups of 64 scales: llmk.hip q4_row_stride)
    static constexpr int RB_E = Q4 ? E / 2 + (E / 32 + 63) / 64 * 128 : E / VPL * 16, RB_H = Q4 ? H / 2 + (H / 32 + 63) / 64 * 128 : H / VPL * 16;
    static constexpr int LPR_E = (E + SEGW - 1) / SEGW, LPR_H = (H + SEGW - 1) / SEGW;   // 1 KB segments per row (q4_0: last may be ragged)
    // ring depth = tiles per streaming wave requested ahead of the dots.  Deeper is not better: what is queued at the memory
    // controllers when a phase ends is what the next exchange's polls wait behind.  f32 (streaming-bound): 5 (4: -2.6 %);
    // f16 (exchange-bound, half the bytes per tile-time): 4 (5: 1,710 tok/s, 4: 1,855, 3: 1,808, 6 spills);
    // q4_0: 4 (a unit is 36 registers; the x fragment of round 4's tiles -- 64 registers -- is gone: x lives in LDS).
#ifndef LLMK_NB_Q4
#define LLMK_NB_Q4 4
#endif
#ifndef LLMK_NB_F16
#define LLMK_NB_F16 4
#endif
    static constexpr int NB = Q4 ? LLMK_NB_Q4 : (WT == WT_F16 ? LLMK_NB_F16 : 5);
    // COOP (q4_0): all eight waves gather the phase's input vector, one eighth each, and the units are requested from
    // inside the dot products (tk_step): a unit's dots take ~1.3 us, the loads' issue is spread through them instead of bursting
    // behind the exchange, and a wave that polls right after its phase has little of its own queued ahead of the poll -- the
    // reason the f32 / f16 kernels keep a wave that never streams (section 3b of HISTORY.md) weighs less here than an 88 KB hb
    // sweep by ONE wave did (7.3 us per layer at 7B, round 2).
    // Measured for f16 too (round 2, ring depth 4): any vector gathered by the streaming waves loses -- hb only 1,710 tok/s,
    // xb only 1,787, all four 1,631, against 1,855 with the service wave alone.  So does holding back the late refills
    // that the next phase does not need until its input vector has been gathered (1,675).
    static constexpr bool COOP = Q4;
    static constexpr int HB_NL = LLMK_TK_HB_NL > 0 ? LLMK_TK_HB_NL : (WT == WT_F32 ? 44 : 32);      // loads per lane in flight in one pass of the hb gather
    // "gather first" (LLMK_TK_GF): tiles per wave a phase's refill burst issues before the next sweep is in the pipeline, and
    // s_sleep units the service wave lets the producers have before the first pass of the x / xa / hb sweeps
    static constexpr int EARLY_A = LLMK_TK_EARLY_A >= 0 ? LLMK_TK_EARLY_A : (WT == WT_F16 ? 1 : 0);
    static constexpr int GF_NOW = LLMK_TK_GF_NOW >= 0 ? LLMK_TK_GF_NOW : 2;     // (f16: 3 until round 4 put a slot of w1|w3 early)
    static constexpr int GF_DELAY = LLMK_TK_GF_DELAY >= 0 ? LLMK_TK_GF_DELAY : (WT == WT_F16 ? 36 : 56);
    static constexpr bool GF_LAST = LLMK_TK_GF_LAST >= 0 ? LLMK_TK_GF_LAST != 0 : WT == WT_F16;
    // the first tiles of a refill burst also wait until the service wave has PUBLISHED the phase's results (stores share the CU's
    // memory pipeline with loads).  Round 4, interleaved on two boxes (profiles/r04_ab.jsonl): f16 +0.5 % four times out of four
    // (kernel 475.3 vs 477.7 us), f32 -1 % twice and equal once (1,452 vs 1,466 tok/s)
    static constexpr bool GF_PUB = LLMK_TK_GF_PUB >= 0 ? LLMK_TK_GF_PUB != 0 : WT == WT_F16;
    static constexpr int RPT = Q4 ? Q16_ROWS : ((WT == WT_F16 && 2 * LPR_E <= TK_TCOLS) ? 2 : 1);   // rows per tile (q4_0: per unit)
    static constexpr int LPT = Q4 ? 1 : TK_TCOLS / RPT;                  // segments of ONE row in a tile (f32 / f16)
    // rows per CU and tiles per CU for each phase
    // The NH attention CUs own NO rows of the QKV and wo matrices (the two phases either side of attention): their q poll,
    // K/V rows and attention never queue behind their own weight prefetch, and nobody waits for them to catch up on
    // streaming after attention.  The other NCU_W CUs split those rows as evenly as whole RoPE pairs / tile rows allow.
    static constexpr int NCU_W = TK_NCU - NH;
    static constexpr int QG = Q4 ? Q16_ROWS : 2;                              // granularity of the QKV split: RoPE pairs, or whole units
    static constexpr int QB = (QKV / QG) / NCU_W, QX = (QKV / QG) % NCU_W;    // pairs (row groups) per CU, CUs with one more
    static constexpr int OB = (E / RPT) / NCU_W, OX = (E / RPT) % NCU_W;      // tile-row groups per CU, CUs with one more
    static constexpr int CB = (V / RPT) / TK_NCU, CX = (V / RPT) % TK_NCU;    // classifier: the same over all CUs
    static constexpr int R_Q = QG * (QB + (QX > 0 ? 1 : 0)), R_O = RPT * (OB + (OX > 0 ? 1 : 0));   // MAX rows per CU
    static constexpr int R_C = RPT * (CB + (CX > 0 ? 1 : 0));
    // hidden units (w1|w3 gate / up row pairs) per CU.  f32 / f16: H / 256 everywhere.  q4_0: whole 16-row groups -- an attention
    // CU, whose service wave never streams (7 waves share its units), takes AG_ATT groups, the others split the rest
    static constexpr int AG_ATT = (H / Q16_ROWS) / TK_NCU, AG_W = H / Q16_ROWS - NH * AG_ATT, AB = AG_W / NCU_W, AX = AG_W % NCU_W;
    // w2 rows per CU: E / 256; q4_0: whole 16-row groups, DB or DB + 1 of them (TinyLlama: 128 groups, one on every other... CU c < DX)
    static constexpr int DB = (E / Q16_ROWS) / TK_NCU, DX = (E / Q16_ROWS) % TK_NCU;
    static constexpr int R_A = Q4 ? 2 * Q16_ROWS * (AB + (AX > 0 ? 1 : 0)) : 2 * (H / TK_NCU),
                         R_D = Q4 ? Q16_ROWS * (DB + (DX > 0 ? 1 : 0)) : E / TK_NCU;
    static constexpr int TPR_H = (LPR_H + LPT - 1) / LPT;                // column parts of a w2 row
    // how the staging code is told which image of the input vector to write: 0 natural order (f32), > 0 the f16 hi | lo image
    // of that many blocks (q4_0: q4_units.h Q16Img)
    static constexpr int TR_E = Q4 ? NBI : 0, TR_H = TR_E;
    // a CU's row count need not be a multiple of RPT: the last tile then also covers rows of the NEXT CU (recomputed,
    // their partial sums land in slots nobody reads).  The weight allocations carry RPT rows of slack at their end.
    static constexpr int NG_A = (R_A / 2 + RPT - 1) / RPT;               // gate (= up) tiles per CU
    static constexpr int NT_Q = (R_Q + RPT - 1) / RPT, NT_O = R_O / RPT, NT_A = 2 * NG_A, NT_D = (R_D / RPT) * TPR_H, NT_C = R_C / RPT;
    // q4_0: units per CU at most, per phase; ALL EIGHT waves take units (wave w: units w, w + 8, ...) except on the attention
    // CUs, whose service wave stays out of the weight stream (its q poll and publish are the layer's critical path): 7 waves
    static constexpr int UQ = R_Q / RPT * NCS_E, UO = R_O / RPT * NCS_E, UA = R_A / RPT * NCS_E, UA_ATT = 2 * AG_ATT * NCS_E,
                         UD = R_D / RPT * NCS_H, UC = R_C / RPT * NCS_E;
    static constexpr int SL_Q = Q4 ? tk_cdiv(UQ, TK_WAVES) : (NT_Q + TK_NS - 1) / TK_NS, SL_O = Q4 ? tk_cdiv(UO, TK_WAVES) : (NT_O + TK_NS - 1) / TK_NS,
                         SL_A = Q4 ? tk_cmax(tk_cdiv(UA, TK_WAVES), tk_cdiv(UA_ATT, TK_NS)) : (NT_A + TK_NS - 1) / TK_NS,
                         SL_C = CLSQ6 ? 0 : Q4 ? tk_cdiv(UC, TK_WAVES) : (NT_C + TK_NS - 1) / TK_NS;   // (q6_K rows: their own loop, tk_q6_phase)
    // w2 rows are TPR_H parts wide: streaming wave sw only ever takes column part sw % TPR_H (so its x fragment can
    // live in registers for the whole phase); the part with the fewest waves (TK_NS / TPR_H of them) sets the slot count
    static constexpr int NW_D = tk_cmax(TK_NS / TPR_H, 1), SL_D = Q4 ? tk_cdiv(UD, TK_NS) : (R_D / RPT + NW_D - 1) / NW_D;
    static constexpr int SL_LAYER = SL_Q + SL_O + SL_A + SL_D;
    static constexpr int RA_P = 2 * RPT * NG_A, RQ_P = RPT * NT_Q;       // partial slots incl. the recomputed neighbour rows
    static constexpr int MAXP00 = RA_P > R_C ? RA_P : R_C, MAXP0 = MAXP00 > RQ_P ? MAXP00 : RQ_P, MAXP1 = R_D * TPR_H,
                         MAXPT = MAXP0 > MAXP1 ? MAXP0 : MAXP1,          // partial sums per phase (f32 / f16: one per tile row)
                         MAXP = Q4 ? tk_cmax(Q16_ROWS * tk_cmax(tk_cmax(UQ, UA), tk_cmax(UD, UC)), R_C) : MAXPT;   // q4_0: 16 per unit
    static_assert(QKV % 2 == 0 && (Q4 || (E % TK_NCU == 0 && H % TK_NCU == 0)) && V % RPT == 0, "rows must split over CUs");
    // K = H rows may end inside a segment (Llama-2-7B: H = 11008 = 21.5 segments of f16): the lanes past the row end multiply the
    // next row's first weights (the tensor's slack behind the last row) with the ZEROS the staged vector is padded with (TkLds::XS_BYTES)
    static_assert(E % SEGW == 0 && E % 32 == 0 && H % 128 == 0, "K = E rows are whole 1 KB segments; hb is gathered in 16-byte loads of two granules");
    static_assert((Q4 || (LPR_E <= LPT && R_Q % RPT == 0 && (R_A / 2) % RPT == 0)) && R_D % RPT == 0 && R_O % RPT == 0,
                  "a K = E row is one tile row; row ranges are whole tiles");
    static_assert(!Q4 || (E % 1024 == 0 && QKV % Q16_ROWS == 0 && KV % Q16_ROWS == 0 && H % Q16_ROWS == 0 && AG_ATT >= 1),
                  "q4_0: K = E rows are whole units, every matrix whole 16-row groups, a hidden group (or more) per attention CU");
    static_assert(NH <= TK_NCU && TK_NCU % NH == 0, "one CU per head");
    static_assert(R_Q <= 64 && R_A / 2 <= 64 && R_O <= 64, "one service lane per output");
    static_assert(HS == 64 || HS == 128, "in-kernel attention is written for head sizes 64 and 128");
    static_assert(TPR_H <= TK_NS, "every column part of a w2 row needs a wave");
};

template <class SH> __device__ __forceinline__ unsigned long long* tk_g_xb(const TokenArgs& a) { return a.g_qkv + SH::QKV; }
template <class SH> __device__ __forceinline__ unsigned long long* tk_g_xa(const TokenArgs& a) { return a.g_qkv + SH::QKV + SH::E; }
template <class SH> __device__ __forceinline__ unsigned long long* tk_g_hb(const TokenArgs& a) { return a.g_qkv + SH::QKV + 2 * SH::E; }
template <class SH> __device__ __forceinline__ unsigned long long* tk_g_x(const TokenArgs& a) { return a.g_qkv + SH::QKV + 2 * SH::E + SH::H; }
__device__ __forceinline__ const float* tk_rms_att(const TokenArgs& a, int l, int E) { return a.rms + (size_t)l * E; }
__device__ __forceinline__ const float* tk_rms_ffn(const TokenArgs& a, int l, int E) { return a.rms + (size_t)(a.L + l) * E; }
__device__ __forceinline__ const float* tk_rms_final(const TokenArgs& a, int E) { return a.rms + (size_t)(2 * a.L) * E; }
__device__ __forceinline__ unsigned long long* tk_trace(const TokenArgs& a) {
#ifdef LLMK_TK_DEBUG
    return a.trace;
#else
    return nullptr;
#endif
}

// LDS carve (bytes): xs (streaming input, up to H floats) | xraw (E) | partial | attention scratch
template <class SH>
struct TkLds {
    static constexpr int XS = 0;
    // q4_0: the streaming input is staged as the matrix core's B operand: an f16 hi | lo image, the blocks' sums, a line of zeros
    // (q4_units.h Q16Img)
    static constexpr int XS_BYTES = SH::Q4 ? Q16Img<SH::NBI>::BYTES : SH::LPR_H * SH::SEGW * 4;      // (whole segments: zeros behind a ragged H)
    static constexpr int XRAW = XS + XS_BYTES;
    static constexpr int PART = XRAW + SH::E * 4;
    static constexpr int ATT_Q = PART + (((SH::MAXP + 1) * 4 + 15) / 16) * 16;   // q_h, k_cur, v_cur: 3*HS floats
    static constexpr int ATT_RED = ATT_Q + 3 * SH::HS * 4;                 // [16 waves][HS/4] float4
    static constexpr int ATT_R4 = ATT_RED + TK_WAVES * (256 / SH::HS) * SH::HS * 4;   // [waves][TPW][HS/4] float4
    static constexpr int RED8 = ATT_R4;                                    // COOP: the eight waves' partial sums of x^2 (32 of the 64 bytes)
    // (q4_0: 256 bytes -- the scales and per-wave maxima of the xb / hb images sit behind the partial sums: TK_XSC_B ... -- and behind
    // them the per-layer records of the previous position's largest xb / hb elements: TK_QSC_LMAX x 16 bytes, tk_qsc)
    static constexpr int QSC = RED8 + 256;
    static constexpr int ROPE = ATT_R4 + (SH::Q4 ? 256 + 128 * 16 : 64);   // cos[HS/2] | sin[HS/2] of pos*freq
    static constexpr int ATT_P = ROPE + SH::HS * 4;                        // [waves][32] softmax weights of the wave's own timesteps
    static constexpr int ATT_S = ATT_P + TK_WAVES * 32 * 4;                // scores [S], then exp(score - max) [S]
};

__device__ __forceinline__ void tk_barrier() {
    // LDS traffic ordered by lgkmcnt; outstanding global LOADS deliberately stay in flight across it
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// the "gather issued" sequence number of a CU (LDS word behind the COOP partial sums): see LLMK_TK_GF.  Read and written
// with explicit ds instructions: through a `volatile int*` hipcc loses the address space and emits FLAT loads, which count
// in vmcnt -- every poll would wait for the whole prefetch ring (s_waitcnt vmcnt(0)).
__device__ __forceinline__ unsigned tk_lds_addr(const volatile void* p) {
    return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const volatile char*)p;
}
__device__ __forceinline__ void tk_flag_set(volatile int* flag, int seq, int lane) {
    __builtin_amdgcn_sched_barrier(0);
    if (lane == 0) asm volatile("ds_write_b32 %0, %1" : : "v"(tk_lds_addr(flag)), "v"(seq) : "memory");   // issued after the pass's buffer loads
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void tk_flag_wait(volatile int* flag, int seq) {
    const unsigned addr = tk_lds_addr(flag);
    for (int spin = 0; spin < (1 << 16); ++spin) {     // bounded: a wedged service wave must not hang the streaming waves too
        int v;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
        if (v - seq >= 0) return;
        __builtin_amdgcn_s_sleep(1);
    }
}
__device__ __forceinline__ void tk_publish(unsigned long long* g, unsigned epoch, float v) {
    __hip_atomic_store(g, ((unsigned long long)epoch << 32) | (unsigned long long)__float_as_uint(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}

typedef unsigned tk_v4u __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t tk_rsrc(const void* p, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

// K/V cache rows through a buffer descriptor: a 32-bit byte offset per lane instead of a 64-bit address pair
__device__ __forceinline__ float4 tk_ldkv(__amdgpu_buffer_rsrc_t rs, int voff) {
    const tk_v4u r = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 0, 0);
    return make_float4(__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w));
}

// Service wave: collect N granules (N % 128 == 0) whose tag == epoch; values -> dst (LDS, N floats).
// A single wave pulls only a few GB/s through agent-scope loads when it waits per batch (the price
// list's handoff-payload row), so: 16-byte sc1 loads (two granules each), ALL of a lane's NL loads in
// flight at once, so a pass costs one round trip.
// Returns false on timeout / sticky error.
// (Measured and dropped, round 2: two passes in flight half a round trip apart, so that a pass that just missed the last
// producer is not followed by a whole further round trip -- f16 1,850 -> 1,480 tok/s, f32 1,380 -> 1,290: twice the poll
// traffic in the service wave's queue costs more than the shorter wait saves.)
template <int NBP>
__device__ __forceinline__ int tk_xoff(int e) { static_assert(NBP == 0, "natural order"); return e; }
template <int NL, int NBP>
__device__ __forceinline__ bool tk_gather_part(__amdgpu_buffer_rsrc_t rs, int first_pair, unsigned epoch, float* dst,
                                               unsigned* err, int lane, bool nowait, unsigned long long* dbg,
                                               volatile int* flag = nullptr, int seq = 0) {
    const int dst0 = tk_xoff<NBP>(2 * (first_pair + lane));
    for (unsigned spin = 0;; ++spin) {
        const unsigned long long tp0 = (TK_DEBUG && dbg) ? wall_clock64() : 0;
        tk_v4u r[NL];
        // NO predicate on the loads: a per-load condition makes hipcc branch around each one and wait
        // vmcnt(0) per element (NL dependent round trips instead of one)
#pragma unroll
        for (int k = 0; k < NL; ++k)
            r[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, (first_pair + lane) * 16, k * WAVE * 16, 16);   // k in the scalar offset: one VGPR address
        if (LLMK_TK_GF && flag && spin == 0) tk_flag_set(flag, seq, lane);   // the first pass is in the pipeline: release the burst
        bool ok = true;
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            ok = ok & (r[k].y == epoch) & (r[k].w == epoch);
            // element 2 * (first_pair + lane + 64 k): lane part + a compile-time stride per k (128 floats), so the address is ONE
            // per-lane register plus an immediate
            *reinterpret_cast<float2*>(dst + dst0 + k * (NBP > 0 ? 16 : 2 * WAVE)) = make_float2(__uint_as_float(r[k].x), __uint_as_float(r[k].z));
        }
        if (__all(ok) || nowait) {
            if (TK_DEBUG && dbg && lane == 0) { dbg[0] = spin + 1; dbg[1] = wall_clock64() - tp0; }
            return true;
        }
        if ((spin & 63) == 63) {
            if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
            if (spin > TK_SPIN_LIMIT) {
                if (lane == 0) __hip_atomic_store(err, 0x100u + (epoch & 0xff), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
        __builtin_amdgcn_s_sleep(LLMK_TK_POLL_SLEEP);
    }
}
// pieces of at most MAXNL loads per lane, one after the other (compile-time recursion): piece k+1 is only polled once piece
// k has arrived.  Smaller pieces make the passes that FAIL cheaper (a pass costs about as much whether or not its tags match,
// and the first pass after a phase almost always fails) but add dependent round trips: see LLMK_TK_HB_NL for the sweep.
template <int NL, int NBP, int MAXNL, int FIRST = 0, bool GFLAST = false>
__device__ __forceinline__ bool tk_gather_pieces(__amdgpu_buffer_rsrc_t rs, unsigned epoch, float* dst, unsigned* err, int lane,
                                                 bool nowait, unsigned long long* dbg, volatile int* flag = nullptr, int seq = 0) {
    constexpr int NP = (NL + MAXNL - 1) / MAXNL;          // pieces still to go
    constexpr int THIS = (NL + NP - 1) / NP;              // as even as possible
    // LLMK_TK_GF: the burst is released by the first pass of the FIRST piece or (GFLAST) of the LAST piece: the whole sweep leads it
    constexpr bool MINE = GFLAST ? (NL <= THIS) : (FIRST == 0);
    const bool a = tk_gather_part<THIS, NBP>(rs, FIRST * WAVE, epoch, dst, err, lane, nowait, dbg, MINE ? flag : nullptr, seq);
    if constexpr (NL > THIS) {
        const bool b = tk_gather_pieces<NL - THIS, NBP, MAXNL, FIRST + THIS, GFLAST>(rs, epoch, dst, err, lane, nowait,
                                                                                     (dbg && FIRST == 0) ? dbg + 2 : nullptr, flag, seq);
        return a && b;
    } else {
        return a;
    }
}
template <int N, int NBP = 0, int MAXNL = 24, bool GFLAST = false>
__device__ __forceinline__ bool tk_gather(const unsigned long long* g, unsigned epoch, float* dst, unsigned* err,
                                          int lane, bool nowait = false, unsigned long long* dbg = nullptr,
                                          volatile int* flag = nullptr, int seq = 0) {
    static_assert(N % 128 == 0, "two granules per 16-byte load, 64 lanes");
    constexpr int NL = N / 128;   // 16-byte loads per lane
    const __amdgpu_buffer_rsrc_t rs = tk_rsrc(g, N * 8);
    // all of a piece's loads are in flight at once (a pass is latency-bound: ~1.4 us per 16 loads per lane under load)
    return tk_gather_pieces<NL, NBP, MAXNL, 0, GFLAST>(rs, epoch, dst, err, lane, nowait, dbg, flag, seq);
}

// ---- cooperative gather (TkShape::COOP): wave w of 8 takes a contiguous eighth of the vector's 16-byte loads ----------
// NLW loads per lane, all in flight per pass.  Values reach LDS only once the whole slice carries the epoch: a slice that
// holds this CU's OWN rows cannot complete before the service wave has published them, i.e. after its epilogue has read
// xraw -- so a wave may start on the next vector while the epilogue of the previous phase still runs.
//   xraw (optional) <- x;  xs <- x * gains (NORM) or x;  *ss += sum x^2 (NORM)
// sum over the 16 lanes of a DPP row (= the HS/4 lanes that share a timestep); valid in lane 15 of the row
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_mov<0xB1, 0xf, true>(0.f, v);
    v += dpp_mov<0x4E, 0xf, true>(0.f, v);
    v += dpp_mov<0x114, 0xf, true>(0.f, v);
    v += dpp_mov<0x118, 0xf, true>(0.f, v);
    return v;
}
// the power of two next below 1 / x (x > 0 finite): the image of a residual-stream vector is written as x * gains * 2^-e, with e
// from the rmsnorm of the vector gathered BEFORE it on this CU -- the stream grows slowly, and the f16 pieces of the image hold
// 2^15 either way of the magnitude they are scaled to; the row sums are divided by the same power of two (exactly) afterwards
__device__ __forceinline__ float tk_pow2_inv(float x) {
    const int e = (int)((__float_as_uint(x) >> 23) & 0xffu);            // biased exponent of x
    return __uint_as_float((unsigned)min(max(254 - e, 1), 254) << 23);
}
constexpr int TK_XSC = 12;      // red8[TK_XSC]: that power of two (q4_0), next to the eight partial sums and the gather-first words
// The images that are NOT residual-stream vectors -- xb (attention output) and hb (SwiGLU output) -- have no rmsnorm to take a scale
// from, and written unscaled a real model's 1e-2 .. 1e-3 activations sit where the lo piece is an f16 subnormal: the absolute floor of
// an element under a high nibble is 2^-20, i.e. 1e-4 .. 1e-3 of such a value (advisor, round 5).  They are written as x 2^e too, with
// 2^e chosen so that the SAME vector of the previous layer had its largest |element| in [64, 128): 2^9 of head room to the end of the
// f16 range, and elements down to 2^-10 of the largest keep a relative error below 2^-21; layer 0 starts from 2^6 (largest element
// assumed in [1, 2)).  red8[TK_XSC_B / _H]: the scale in force; red8[TK_AMX_B / _H + w]: wave w's largest |scaled element| of the image
// it has just written (every wave writes one eighth).  The service wave divides the row sums by the scale (exactly) and sets the
// next layer's from the eight maxima behind the phase's second barrier.  A vector that still does not fit raises 0x4000 as before.
// ACROSS POSITIONS (round 6, second step): the layer before is a poor predictor where a model's activations jump between layers and
// stay put between tokens (Llama-2's down-projection inputs of layers 1 and 30 are hundreds of times their neighbours') -- a vector
// 2^9 above its predecessor would raise 0x4000 at EVERY position and retire the kernel after four.  So each launch leaves, per layer,
// the largest |element| of its xb and hb vectors with its position (tk_qsc: 16 bytes per layer behind the exchange granules, written
// by CU 1 -- every CU computes the same numbers), and the next launch, if it is the NEXT position, scales layer l's images from
// layer l's record of the position before; the layer-before rule is the fallback (position 1, the position after a prefill or a redo).
constexpr int TK_XSC_B = 16, TK_XSC_H = 17, TK_AMX_B = 24, TK_AMX_H = 32;
constexpr int TK_EVT = 40;             // red8[TK_EVT] (as unsigned): the first range event THIS workgroup met or heard of, 0 = none (tk_rec_ok)
constexpr int TK_QSC_LMAX = 128;        // layers with a record (deeper models: the layer-before rule beyond)
constexpr float TK_IMG_TARGET = 64.f;
// the largest |element| of the image just written, as it is (cur = the power of two it was scaled with: the division is exact)
__device__ __forceinline__ float tk_amax_true(float cur, const float* amx8) {
    const float4 a = reinterpret_cast<const float4*>(amx8)[0], b = reinterpret_cast<const float4*>(amx8)[1];      // (two 16-byte LDS reads)
    return fmaxf(fmaxf(fmaxf(a.x, a.y), fmaxf(a.z, a.w)), fmaxf(fmaxf(b.x, b.y), fmaxf(b.z, b.w))) / cur;
}
__device__ __forceinline__ float tk_scale_for(float amax, float fallback) {
    return (amax > 0.f && amax < 3.0e38f) ? tk_pow2_inv(amax * (1.f / TK_IMG_TARGET)) : fallback;
}
// record of layer l: {largest |xb|, largest |hb|, position that wrote .x, position that wrote .y} (positions as int bits).  Two
// buffers by position parity: a launch reads the one the position before wrote and writes its own -- no workgroup can meet a record
// of its own launch, however late it starts (the choice of scale must not depend on timing: reruns are bit-identical)
template <class SH>
__device__ __forceinline__ float4* tk_qsc(const TokenArgs& a, int pos) {
    constexpr size_t NG = (size_t)SH::QKV + 3 * (size_t)SH::E + SH::H + (size_t)SH::NH * ((TK_NCU / SH::NH < 8 ? TK_NCU / SH::NH : 8) - 1) * (SH::HS + 2);
    return reinterpret_cast<float4*>(a.g_qkv + NG) + (pos & 1) * TK_QSC_LMAX;
}
// Two predictions of the next image's largest |element|: this token's vector of the layer before (`now`), the same layer's vector
// of the position before (`hist`, 0 = none).  Models differ in which one jumps -- between LAYERS (Llama-2's massive activations: the
// history is right, the layer before 2^11 off) or between TOKENS (the bench's synthetic 7B weights: hb is ~1 or ~92 by token, the layer
// before is right) -- and an image holds 2^9 above its target and full precision 2^6 below: the GEOMETRIC MEAN halves the exponent
// of whichever is wrong (measured round 6: the maximum of the two lost 1.2e-4 on the layer behind a jump, either alone lost the kernel
// or a position per jump)
__device__ __forceinline__ float tk_predict(float now, float hist) { return hist > 0.f ? sqrtf(now * hist) : now; }
// may the vector gathered under `epoch` leave its record?  Yes while the sticky word is clear, and when it is THIS gather's own range event
// (decided on the workgroup's own LDS word, set by the waves that met the event or gave up on hearing of it, in front of the barrier this
// is read behind: the device word is stored by other waves without an order against this load, and later gathers over the debris overwrite it)
__device__ __forceinline__ bool tk_rec_ok(const float* red8, unsigned epoch) {
    const unsigned e = reinterpret_cast<const unsigned*>(red8)[TK_EVT];
    return e == 0 || e == (0x4000u | (epoch & 0xffu));
}
__device__ __forceinline__ void tk_evt_note(unsigned* evt, unsigned epoch, int lane) {
    if (evt && lane == 0 && *evt == 0) *evt = 0x4000u | (epoch & 0xffu);
}
// the largest |element| layer l's vector KIND (0 xb, 1 hb) had at the position before (0 when that position left no record)
__device__ __forceinline__ float tk_hist_amax(const float4* qsc_lds, int l, int L, int kind, int pos) {
    if (l >= L || l >= TK_QSC_LMAX) return 0.f;
    const float4 r = qsc_lds[l];
    const int tag = __float_as_int(kind ? r.w : r.z);
    const float v = kind ? r.y : r.x;
    return (tag == pos - 1 && v > 0.f && v < 3.0e38f) ? v : 0.f;
}
template <int NLW, int NBP, bool NORM>
__device__ __forceinline__ bool tk_coop_part(__amdgpu_buffer_rsrc_t rs, int first_pair, unsigned epoch, float* xraw, float* xs,
                                             const float* __restrict__ gains, float* ss, unsigned* err, int lane, bool nowait, float psc = 1.f,
                                             float* amx = nullptr, unsigned* evt = nullptr) {
    if constexpr (NLW == 0) { if (amx && lane == 0) *amx = 0.f; return true; }
    else {
        float2 gn[NORM ? NLW : 1];
        if constexpr (NORM) {
#pragma unroll
            for (int k = 0; k < NLW; ++k) {
                gn[k] = *reinterpret_cast<const float2*>(gains + 2 * (first_pair + lane + k * WAVE));
                if constexpr (NBP > 0) { gn[k].x *= psc; gn[k].y *= psc; }      // (a power of two: x (g 2^-e) = (x g) 2^-e, and this is done before the poll)
            }
        }
        const int e0 = 2 * (first_pair + lane);
        // NBP > 0: the f16 hi | lo image of NBP blocks (q4_units.h): where this lane's pair goes, its factor (1, or 1/16 under a high
        // nibble), and the slot of its block's sum; load k lies 128 elements = 4 blocks further
        char* ip = reinterpret_cast<char*>(xs) + (NBP > 0 ? q16_pair_off(e0) : 0);
        const float isc = NBP > 0 ? q16_pair_scale(e0) : 1.f;
        float amax = 0.f;               // NBP > 0: the largest |value| written as an f16 hi piece (65504 is the end of that format)
        char* sp = reinterpret_cast<char*>(xs) + (NBP > 0 ? Q16Img<(NBP > 0 ? NBP : 32)>::SUM + (e0 >> 5) * 2 : 0);
        for (unsigned spin = 0;; ++spin) {
            tk_v4u r[NLW];
#pragma unroll
            for (int k = 0; k < NLW; ++k) r[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, (first_pair + lane) * 16, k * WAVE * 16, 16);
            bool ok = true;
#pragma unroll
            for (int k = 0; k < NLW; ++k) ok = ok & (r[k].y == epoch) & (r[k].w == epoch);
            if (__all(ok) || nowait) {
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < NLW; ++k) {
                    const float x0 = __uint_as_float(r[k].x), x1 = __uint_as_float(r[k].z);
                    if (xraw) *reinterpret_cast<float2*>(xraw + e0 + k * 2 * WAVE) = make_float2(x0, x1);
                    float y0 = x0, y1 = x1;
                    if constexpr (NORM) {
                        acc = fmaf(x0, x0, acc);
                        acc = fmaf(x1, x1, acc);
                        y0 = x0 * gn[k].x; y1 = x1 * gn[k].y;
                    }
                    if constexpr (NBP > 0) {
                        if constexpr (!NORM) { y0 *= psc; y1 *= psc; }
                        amax = fmaxf(amax, fmaxf(fabsf(y0), fabsf(y1)));
                        q16_put2(ip + k * (4 * Q16_IMG_BLK), isc, y0, y1);
                        const float bs = row16_sum(y0 + y1);             // a load's 64 lanes hold 4 whole blocks, one per DPP row
                        // (the sum is held as sum / 32 -- never above the block's largest element, so it needs no head room of its own:
                        // q4_units.h Q16_SUM_DIV; written whole, 32 coherent elements of 2^12 overflowed where none of them did)
                        if ((lane & 15) == 15) q16_put_sum<(NBP > 0 ? NBP : 32)>(sp + k * 8, bs * (1.0f / Q16_SUM_DIV));
                    } else {
                        *reinterpret_cast<float2*>(xs + e0 + k * 2 * WAVE) = make_float2(y0, y1);
                    }
                }
                if constexpr (NORM) *ss += acc;
                if constexpr (NBP > 0) {
                    // an activation beyond the f16 range (or not finite): the image is useless -- raise the sticky word, the host retires
                    // the kernel for this context and redoes the position on the multi-kernel path (f32 throughout)
                    if (amx) { const float wm = wave_max(amax); if (lane == 0) *amx = wm; }
                    if (!nowait && __any(!(amax < 60000.f))) {
                        if (lane == 0) __hip_atomic_store(err, 0x4000u + (epoch & 0xff), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        tk_evt_note(evt, epoch, lane);
                        return false;
                    }
                }
                return true;
            }
            if ((spin & 63) == 63) {
                // (a range event another workgroup met in this very vector -- all of them read the same one; anything else ends the launch)
                if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { tk_evt_note(evt, epoch, lane); return false; }
                if (spin > TK_SPIN_LIMIT) {
                    if (lane == 0) __hip_atomic_store(err, 0x400u + (epoch & 0xff), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    tk_evt_note(evt, epoch, lane);
                    return false;
                }
            }
            __builtin_amdgcn_s_sleep(LLMK_TK_POLL_SLEEP);
        }
    }
}
// wave w (0..7) of the workgroup; red8[w] receives the slice's sum of squares when NORM
// KIND (q4_0 images that are not residual-stream vectors): 1 = xb, 2 = hb -- which scale word and which maxima (TK_XSC_B ...)
template <int N, int NBP, bool NORM, int KIND = 0>
__device__ __forceinline__ bool tk_coop_gather(const unsigned long long* g, unsigned epoch, float* xraw, float* xs, const float* gains,
                                               float* red8, unsigned* err, int w, int lane, bool nowait) {
    static_assert(N % 128 == 0, "two granules per 16-byte load, 64 lanes");
    constexpr int NL = N / 128, B = NL / TK_WAVES, X = NL % TK_WAVES;      // waves < X take B + 1 loads per lane, the others B
    const __amdgpu_buffer_rsrc_t rs = tk_rsrc(g, N * 8);
    float ss = 0.f;
    bool ok;
    static_assert(KIND == 0 || (!NORM && NBP > 0), "scaled xb / hb images are the q4_0 kernels'");
    const float psc = NBP > 0 ? (NORM ? red8[TK_XSC] : KIND == 1 ? red8[TK_XSC_B] : KIND == 2 ? red8[TK_XSC_H] : 1.f) : 1.f;   // q4_0: see tk_pow2_inv, TK_XSC_B
    float* amx = KIND == 0 ? nullptr : red8 + (KIND == 1 ? TK_AMX_B : TK_AMX_H) + w;
    unsigned* evt = NBP > 0 ? reinterpret_cast<unsigned*>(red8) + TK_EVT : nullptr;
    if (w < X) ok = tk_coop_part<B + 1, NBP, NORM>(rs, w * (B + 1) * WAVE, epoch, xraw, xs, gains, &ss, err, lane, nowait, psc, amx, evt);
    else ok = tk_coop_part<B, NBP, NORM>(rs, (X * (B + 1) + (w - X) * B) * WAVE, epoch, xraw, xs, gains, &ss, err, lane, nowait, psc, amx, evt);
    if constexpr (NORM) {
        ss = wave_sum(ss);
        if (lane == 0) red8[w] = ss;
    }
    return ok;
}
// the eight partial sums -> sqrt(mean(x^2) + eps)                                                    llama2.f90:454
template <int E>
__device__ __forceinline__ float tk_coop_xn(const float* red8, float eps) {
    const float ss = ((red8[0] + red8[1]) + (red8[2] + red8[3])) + ((red8[4] + red8[5]) + (red8[6] + red8[7]));
    return sqrtf(ss / (float)E + eps);
}

// rmsnorm on one wave, lane <-> 4 consecutive elements per 256: the gains are requested (plain
// loads, static data) BEFORE the exchange is polled, so their HBM latency hides under the gather.
//   xs = x*w/sqrt(mean(x^2)+1e-5)                                         llama2.f90:450-457
template <int E>
struct TkNorm {
    static constexpr int PER = E / (4 * WAVE);
    float4 w[PER];
    __device__ __forceinline__ void prefetch(const float* __restrict__ gains, int lane) {
#pragma unroll
        for (int k = 0; k < PER; ++k) w[k] = reinterpret_cast<const float4*>(gains)[lane + k * WAVE];
    }
    // Stages xs = x*w and returns xn = sqrt(mean(x^2)+1e-5).  The division by xn is linear in the dot
    // product, so it is applied ONCE to each finished row sum (W.(x*w))/xn by the epilogue instead
    // of 2048 times here -- the service wave is the serial section of every phase.
    template <int NBP = 0>   // (natural order only: the q4_0 kernels stage their image in pieces, tk_stage_q16)
    __device__ __forceinline__ float apply(const float* xraw, float* xs, int lane, float eps) const {
        float ss = 0.f;
        const int xs0 = tk_xoff<NBP>(4 * lane);
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const float4 x = reinterpret_cast<const float4*>(xraw)[lane + k * WAVE];
            ss = dot4(x, x, ss);
            float4 o;
            o.x = x.x * w[k].x;
            o.y = x.y * w[k].y;
            o.z = x.z * w[k].z;
            o.w = x.w * w[k].w;
            *reinterpret_cast<float4*>(xs + xs0 + k * (4 * WAVE)) = o;
        }
        ss = wave_sum(ss);
        return sqrtf(ss / (float)E + eps);
    }
};

// Every ring slot is exactly TK_TCOLS unconditional loads: a slot with no tile, and the columns past a
// ragged row end, read a 1 KB block of ZEROS (L2-resident) instead of being skipped.  With no control
// flow around the loads hipcc can count them, so consuming the oldest of the NB tiles waits with
// vmcnt(24) and leaves the three younger tiles in flight; a skipped load would force vmcnt(0).
struct TkTile {
    const float4* p;  // first segment of the tile's first row (the zero block when the slot is empty); q4_0: the unit's 9,216 bytes
    int rstride;      // float4 units between the tile's rows (RPT > 1)
    int ncol;         // real segments per tile row (0..LPT); segments >= ncol read zeros; q4_0: 1 = a unit, 0 = none
    int pidx;         // partial index of the tile's first row (MAXP = junk slot); row s of the tile: pidx + s * pstep
    int pstep;        // q4_0: the unit's column slot (which 32 blocks of the x image it is dotted with)
};

typedef _Float16 tk_h2 __attribute__((ext_vector_type(2)));
// one ring entry: the 8 weight vectors of a tile (+ for q4_0 the 8 block scales that go with them)
template <class SH>
struct TkSlot {
    float4 b[TK_TCOLS + (SH::Q4 ? 1 : 0)];      // q4_0: b[8] = the lane's 16 bytes of the unit's scale plane
};
// a wave's ring: the tiles (q4_0: units) it has requested ahead, and their descriptors
template <class SH>
struct TkRing {
    TkSlot<SH> b[SH::NB];
    TkTile t[SH::NB];
};

template <class SH>
__device__ __forceinline__ void tk_issue(TkSlot<SH>& e, const TkTile& t, const float4* zp, int lane) {
    if constexpr (SH::Q4) {
        // a unit: 8 x 1 KB of operand dwords + 1 KB of scales, contiguous (q4_units.h); no unit = nine reads of one line of zeros
        const bool real = t.ncol != 0 && !TK_EXP_NOHBM;
#pragma unroll
        for (int j = 0; j <= TK_TCOLS; ++j) e.b[j] = ldg_nt(real ? t.p + j * WAVE + lane : zp);
        return;
    }
#pragma unroll
    for (int j = 0; j < TK_TCOLS; ++j) {
        const int s = j / SH::LPT, jj = j % SH::LPT;                             // compile-time
        const bool real = jj < t.ncol && !TK_EXP_NOHBM;                            // wave-uniform select, no branch
        const float4* pj = real ? t.p + s * t.rstride + jj * WAVE : zp;
        // an empty segment is ONE 16-byte access for the whole wave (every lane reads the same zero vector), not a 1 KB
        // sweep of the zero block: it still counts in vmcnt, but costs the CU's memory pipeline one line instead of eight
        e.b[j] = ldg_nt(pj + (real ? lane : 0));
    }
}
// Every tile of a phase is dotted against the same x fragment (the LPT segments of a row, or of one column part of a
// w2 row): it is read from LDS once per phase into registers, not once per tile -- 7 waves x 8 KB of ds_read per
// slot was ~0.2 us of LDS time in the middle of every slot of the critical path.  32 floats per lane for f32 and f16.
template <class SH>
struct TkX {
    static constexpr int F4 = SH::Q4 ? 1 : SH::VPL / 4;   // float4 of x per segment and lane: 1 (f32), 2 (f16); q4_0 keeps x in LDS
    float4 v[SH::LPT * F4];
    // segments seg0 .. seg0+LPT-1 of a vector with nseg segments; segments past the end read as zero
    __device__ __forceinline__ void load(const float4* xs, int seg0, int nseg, int lane) {
#pragma unroll
        for (int j = 0; j < SH::LPT; ++j) {
            const bool in = seg0 + j < nseg;
#pragma unroll
            for (int h = 0; h < F4; ++h) {
                const float4 x = xs[in ? ((seg0 + j) * WAVE + lane) * F4 + h : lane];
                v[j * F4 + h] = in ? x : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }
};
__device__ __forceinline__ float4 tk_h2f_lo(const float4& w) {   // halves 0..3 of a 16-byte vector of 8
    const __half2 a = *reinterpret_cast<const __half2*>(&w.x), b = *reinterpret_cast<const __half2*>(&w.y);
    return make_float4(__low2float(a), __high2float(a), __low2float(b), __high2float(b));
}
__device__ __forceinline__ float4 tk_h2f_hi(const float4& w) {   // halves 4..7
    const __half2 a = *reinterpret_cast<const __half2*>(&w.z), b = *reinterpret_cast<const __half2*>(&w.w);
    return make_float4(__low2float(a), __high2float(a), __low2float(b), __high2float(b));
}
// per-lane partial dots of one tile (one per tile row): four independent FMA chains (x,y,z,w) instead of one long chain
template <class SH>
__device__ __forceinline__ void tk_dot(const TkSlot<SH>& e, const TkX<SH>& x, float (&out)[SH::RPT]) {
    static_assert(!SH::Q4, "q4_0 units are dotted on the matrix core (tk_unit)");
    const float4 (&b)[TK_TCOLS] = e.b;
    if constexpr (TK_EXP_NODOT) {          // the tile is waited for (every register named), not multiplied
#pragma unroll
        for (int j = 0; j < TK_TCOLS; ++j) asm volatile("" :: "v"(b[j].x), "v"(b[j].y), "v"(b[j].z), "v"(b[j].w));
#pragma unroll
        for (int s = 0; s < SH::RPT; ++s) out[s] = x.v[0].x;
        return;
    }
#pragma unroll
    for (int s = 0; s < SH::RPT; ++s) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int jj = 0; jj < SH::LPT; ++jj) {
            const float4& w = b[s * SH::LPT + jj];
            if constexpr (SH::WT == WT_F16) {
                const float4 lo = tk_h2f_lo(w), hi = tk_h2f_hi(w);
                const float4 &xl = x.v[2 * jj], &xh = x.v[2 * jj + 1];
                acc.x = fmaf(lo.x, xl.x, acc.x); acc.y = fmaf(lo.y, xl.y, acc.y);
                acc.z = fmaf(lo.z, xl.z, acc.z); acc.w = fmaf(lo.w, xl.w, acc.w);
                acc.x = fmaf(hi.x, xh.x, acc.x); acc.y = fmaf(hi.y, xh.y, acc.y);
                acc.z = fmaf(hi.z, xh.z, acc.z); acc.w = fmaf(hi.w, xh.w, acc.w);
            } else {
                acc.x = fmaf(w.x, x.v[jj].x, acc.x);
                acc.y = fmaf(w.y, x.v[jj].y, acc.y);
                acc.z = fmaf(w.z, x.v[jj].z, acc.z);
                acc.w = fmaf(w.w, x.v[jj].w, acc.w);
            }
        }
        out[s] = (acc.x + acc.y) + (acc.z + acc.w);
    }
}
template <class SH>
__device__ __forceinline__ void tk_consume(const TkSlot<SH>& b, const TkTile& t, const TkX<SH>& x, float* part, int lane) {
    float v[SH::RPT];
    tk_dot<SH>(b, x, v);
#pragma unroll
    for (int s = 0; s < SH::RPT; ++s) v[s] = wave_sum(v[s]);
    if (lane == 0) {
#pragma unroll
        for (int s = 0; s < SH::RPT; ++s) part[t.pidx + s * t.pstep] = v[s];
    }
}

// Static per-wave schedule: SLP slots per layer (padded to an even count so the 2-deep ring has the
// same parity at every layer start), then the classifier slots.  Slot K (compile-time) belongs to
// one phase; the s-th slot of a phase is tile s*15 + sw of the CU's tile range of that phase
// (null when past the end).  Everything but l, c, sw is a compile-time constant.
template <class SH>
struct TkSched {
    static constexpr int SLP = (SH::SL_LAYER + SH::NB - 1) / SH::NB * SH::NB;   // padded slots per layer
    static constexpr int KQ = 0, KO = KQ + SH::SL_Q, KA = KO + SH::SL_O, KD = KA + SH::SL_A, KP = KD + SH::SL_D;
};

template <class SH>
__device__ __forceinline__ TkTile tk_null(const float4* zp) {
    TkTile t;
    t.p = zp; t.rstride = 0; t.ncol = 0; t.pidx = SH::MAXP; t.pstep = 0;
    return t;
}
// row r of a [rows][K] matrix of SH's weight type
template <class SH, int K>
__device__ __forceinline__ const float4* tk_rowp(const void* mat, long long r) {
    return reinterpret_cast<const float4*>(static_cast<const char*>(mat) + (size_t)r * (K == SH::E ? SH::RB_E : SH::RB_H));
}

// tile ti of a CU's run-time row range [row0, row0+n) of a K = E matrix: RPT consecutive (= contiguous) rows
template <class SH>
__device__ __forceinline__ TkTile tk_row_tile(const void* mat, long long row0, int ti, int n, const float4* zp) {
    TkTile t;
    const bool live = ti * SH::RPT < n;
    t.ncol = live ? SH::LPR_E : 0;
    t.p = live ? tk_rowp<SH, SH::E>(mat, row0 + ti * SH::RPT) : zp;
    t.rstride = SH::RB_E / 16;
    t.pidx = live ? ti * SH::RPT : SH::MAXP;
    t.pstep = live ? 1 : 0;
    return t;
}
// q4_0: unit u of the U units a CU owns in a phase (units unit0 .. of the matrix, q4_units.h), for wave sw of the nw that take
// units there (8; 7 on an attention CU, whose service wave stays out of the weight stream).  16 partial sums at u * 16.
template <class SH>
__device__ __forceinline__ TkTile tk_unit(const void* mat, int unit0, int u, int U, int ncs, int sw, int nw, const float4* zp) {
    TkTile t;
    const bool live = sw < nw && u < U;
    t.ncol = live ? 1 : 0;
    t.p = live ? reinterpret_cast<const float4*>(static_cast<const char*>(mat) + (size_t)(unsigned)(unit0 + u) * Q16_UNIT_BYTES) : zp;
    t.rstride = 0;
    t.pidx = live ? u * Q16_ROWS : 0;
    t.pstep = u % ncs;
    return t;
}

template <class SH, int K>
__device__ __forceinline__ TkTile tk_cls_at(const TokenArgs& a, int c, int sw) {
    if constexpr (K >= SH::SL_C) return tk_null<SH>(a.zeros);
    else if constexpr (SH::Q4) return tk_unit<SH>(a.wcls, a.c0 / Q16_ROWS * SH::NCS_E, K * TK_WAVES + sw, a.cn / Q16_ROWS * SH::NCS_E, SH::NCS_E, sw, TK_WAVES, a.zeros);
    else if constexpr (SH::CX == 0) return tk_row_tile<SH>(a.wcls, (long long)c * SH::R_C, K * TK_NS + sw, SH::R_C, a.zeros);   // even split: compile-time count
    else return tk_row_tile<SH>(a.wcls, a.c0, K * TK_NS + sw, a.cn, a.zeros);
}

// w1|w3 tile ti of CU c: tile 2m is RPT gate rows, tile 2m+1 the RPT up rows of the same hidden units
template <class SH>
__device__ __forceinline__ TkTile tk_w13_tile(const TokenArgs& a, int l, int c, int ti, bool live) {
    const int m = ti >> 1, gu = ti & 1;
    const long long row = (long long)l * 2 * SH::H + gu * SH::H + c * (SH::R_A / 2) + m * SH::RPT;
    TkTile t;
    t.ncol = live ? SH::LPR_E : 0;
    t.p = live ? tk_rowp<SH, SH::E>(a.w13, row) : a.zeros;
    t.rstride = SH::RB_E / 16;
    t.pidx = live ? 2 * m * SH::RPT + gu : SH::MAXP;
    t.pstep = live ? 2 : 0;
    return t;
}
// w2 tile of streaming wave sw in slot k of the phase: RPT rows x column part `part` (LPT segments; the last part is ragged)
template <class SH>
__device__ __forceinline__ TkTile tk_w2_tile(const TokenArgs& a, int l, int c, int sw, int k) {
    constexpr int P = SH::TPR_H;
    const int part = sw % P, nw = (TK_NS - part + P - 1) / P;      // waves that share this column part
    const int rg = k * nw + sw / P;                                // group of RPT rows
    const bool live = rg < SH::R_D / SH::RPT;
    const long long row = (long long)l * SH::E + c * SH::R_D + rg * SH::RPT;
    TkTile t;
    t.ncol = live ? min(SH::LPT, SH::LPR_H - part * SH::LPT) : 0;
    t.p = live ? tk_rowp<SH, SH::H>(a.w2, row) + part * SH::LPT * WAVE : a.zeros;
    t.rstride = SH::RB_H / 16;
    t.pidx = live ? rg * SH::RPT * P + part : SH::MAXP;
    t.pstep = live ? P : 0;
    return t;
}
// descriptor of slot K (compile-time) of layer l; K >= SLP looks into layer l+1; past the last
// layer the stream continues with the classifier slots
template <class SH, int K>
__device__ __forceinline__ TkTile tk_at(const TokenArgs& a, int l, int c, int sw) {
    typedef TkSched<SH> SC;
    static_assert(K < 2 * SC::SLP, "lookahead of at most one layer");
    if constexpr (K >= SC::SLP) {
        return tk_at<SH, K - SC::SLP>(a, l + 1, c, sw);
    } else {
        if (l >= a.L) return tk_cls_at<SH, K>(a, c, sw);
        if constexpr (SH::Q4) {
            // q4_0: the CU's units of the phase, wave sw takes units sw, sw + nw, ...  (w1|w3: the gate groups' units, then the up groups')
            if constexpr (K < SC::KO) {
                return tk_unit<SH>(a.wqkv, (l * SH::QKV + a.q0) / Q16_ROWS * SH::NCS_E, (K - SC::KQ) * tk_q4(a).nw + sw,
                                   a.qn / Q16_ROWS * SH::NCS_E, SH::NCS_E, sw, tk_q4(a).nw, a.zeros);
            } else if constexpr (K < SC::KA) {
                return tk_unit<SH>(a.wo, (l * SH::E + a.o0) / Q16_ROWS * SH::NCS_E, (K - SC::KO) * tk_q4(a).nw + sw,
                                   a.on / Q16_ROWS * SH::NCS_E, SH::NCS_E, sw, tk_q4(a).nw, a.zeros);
            } else if constexpr (K < SC::KD) {
                const int u = (K - SC::KA) * tk_q4(a).nw + sw, ug = tk_q4(a).hn / Q16_ROWS * SH::NCS_E, up = u >= ug ? 1 : 0;
                TkTile t = tk_unit<SH>(a.w13, (l * 2 * SH::H + up * SH::H + tk_q4(a).h0) / Q16_ROWS * SH::NCS_E - up * ug, u, 2 * ug,
                                       SH::NCS_E, sw, tk_q4(a).nw, a.zeros);
                return t;
            } else if constexpr (K < SC::KP) {
                return tk_unit<SH>(a.w2, (l * SH::E + tk_q4(a).d0) / Q16_ROWS * SH::NCS_H, (K - SC::KD) * tk_q4(a).nw + sw,
                                   tk_q4(a).dn / Q16_ROWS * SH::NCS_H, SH::NCS_H, sw, tk_q4(a).nw, a.zeros);
            } else {
                return tk_null<SH>(a.zeros);
            }
        } else
        if constexpr (K < SC::KO) {
            return tk_row_tile<SH>(a.wqkv, (long long)l * SH::QKV + a.q0, (K - SC::KQ) * TK_NS + sw, a.qn, a.zeros);
        } else if constexpr (K < SC::KA) {
            return tk_row_tile<SH>(a.wo, (long long)l * SH::E + a.o0, (K - SC::KO) * TK_NS + sw, a.on, a.zeros);
        } else if constexpr (K < SC::KD) {
            // w1|w3: tile 2m is RPT gate rows, tile 2m+1 the RPT up rows of the same hidden units (SwiGLU pairs stay in the CU);
            // partials are laid out (gate, up) per hidden unit
            const int ti = (K - SC::KA) * TK_NS + sw;
            return tk_w13_tile<SH>(a, l, c, ti, ti < SH::NT_A);
        } else if constexpr (K < SC::KP) {
            return tk_w2_tile<SH>(a, l, c, sw, K - SC::KD);
        } else {
            return tk_null<SH>(a.zeros);
        }
    }
}

// COOP (q4_0): "lagged refill".  Slot K consumes ring entry K % NB and, INSIDE its dots, requests unit K + NB - 1 into the
// entry slot K - 1 has just freed.  Issuing a load blocks while the CU's memory pipeline is full; spread through the dots it
// never does, there is no refill burst anywhere (the f32 / f16 kernels place theirs behind the exchange instead, section 3b),
// and a wave that polls after its phase has few requests of its own queued ahead of the poll.  Prefetch distance NB - 1 units.
// the request slot K of a phase would issue from inside its dots (unit K + NB - 1 into the entry slot K - 1 has freed), as a
// statement of its own: LLMK_TK_ADV_HOP issues it in the window BEFORE the phase, and the slot issues none
template <class SH, int K, bool CLS>
__device__ __forceinline__ void tk_request(TkRing<SH>& r, const TokenArgs& a, int l, int c, int sw, int lane) {
    constexpr int RN = (K + SH::NB - 1) % SH::NB;
    if constexpr (CLS) r.t[RN] = tk_cls_at<SH, K + SH::NB - 1>(a, c, sw);
    else r.t[RN] = tk_at<SH, K + SH::NB - 1>(a, l, c, sw);
    tk_issue<SH>(r.b[RN], r.t[RN], a.zeros, lane);
}
// One slot of a q4_0 phase: the unit in ring entry K % NB against the x image (q4_units.h), 16 partial sums to LDS; a wave
// whose slot holds no unit (a CU with fewer row groups, the last round of a phase, the service wave of an attention CU, padding)
// skips the dots -- its SIMD is then the partner wave's alone -- but issues the same nine loads: both arms of the branch leave
// the same loads outstanding, so hipcc's counters stay exact (a load inside ONE arm would make every later wait a vmcnt(0)).
template <class SH, int K, bool CLS, bool REQ = true>
__device__ __forceinline__ void tk_step(TkRing<SH>& r, const TokenArgs& a, int l, int c, int sw, const char* img, float* part, int lane) {
    static_assert(SH::Q4, "q4_0 units");
    constexpr int R = K % SH::NB, RN = (K + SH::NB - 1) % SH::NB;
    const TkSlot<SH>& e = r.b[R];
    TkTile tn;
    if constexpr (!REQ) tn = r.t[RN];              // already requested (tk_request): the entry and its descriptor stay as they are
    else if constexpr (CLS) tn = tk_cls_at<SH, K + SH::NB - 1>(a, c, sw);
    else tn = tk_at<SH, K + SH::NB - 1>(a, l, c, sw);
    TkSlot<SH>& n = r.b[RN];
    const bool nreal = tn.ncol != 0 && !TK_EXP_NOHBM;
    // the request in three pieces (4 + 4 + 1 loads), placed between the half groups of the dots
    auto req = [&](auto piece) {
        if constexpr (REQ) {
            constexpr int P = decltype(piece)::value;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = (P == 0 ? 0 : P == 1 ? 4 : 8); j < (P == 0 ? 4 : P == 1 ? 8 : 9); ++j) n.b[j] = ldg_nt(nreal ? tn.p + j * WAVE + lane : a.zeros);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // a padding slot (the layer's slots rounded up to the ring depth) holds no unit on any wave of any CU
    constexpr bool PAD = !CLS && (K % TkSched<SH>::SLP) >= TkSched<SH>::KP;
    const TkTile tc = r.t[R];
    if (!PAD && tc.ncol != 0) {                    // wave-uniform
        Q16Lane ln;
        q16_lane<SH::NBI>(ln, img, tc.pstep, lane);
        const q16_v8h sel = q16_sel(lane);
        q16_v4f acc = {0.f, 0.f, 0.f, 0.f}, acc8 = {0.f, 0.f, 0.f, 0.f};
        q16_unit_dot(e.b, ln, sel, lane >> 4, acc, acc8, req);
        q16_unit_store(acc, acc8, part + tc.pidx, lane);
    } else {
        req(std::integral_constant<int, 0>());
        req(std::integral_constant<int, 1>());
        req(std::integral_constant<int, 2>());
    }
    r.t[RN] = tn;
}
// REQ0 = false: the first slot's request was issued in the window before the phase (tk_request)
template <class SH, int K, int N, bool CLS, bool REQ0 = true>
__device__ __forceinline__ void tk_steps(TkRing<SH>& r, const TokenArgs& a, int l, int c, int sw, const char* img, float* part, int lane) {
    if constexpr (N > 0) {
        tk_step<SH, K, CLS, REQ0>(r, a, l, c, sw, img, part, lane);
        tk_steps<SH, K + 1, N - 1, CLS>(r, a, l, c, sw, img, part, lane);
    }
}
template <class SH, int K0, int S, bool CLS, bool REQ0 = true>
__device__ __forceinline__ void tk_phase_body(TkRing<SH>& r, const TokenArgs& a, int l, int c, int sw, const char* img, float* part, int lane) {
    tk_barrier();
    tk_steps<SH, K0, S, CLS, REQ0>(r, a, l, c, sw, img, part, lane);
    tk_barrier();
}
// the ring starts with tiles 0 .. NB-2 requested; slot 0 requests tile NB-1
template <class SH, int K>
__device__ __forceinline__ void tk_prime_coop(TkRing<SH>& r, const TokenArgs& a, int c, int sw, int lane) {
    if constexpr (K < SH::NB - 1) {
        r.t[K] = tk_at<SH, K>(a, 0, c, sw);
        tk_issue<SH>(r.b[K], r.t[K], a.zeros, lane);
        tk_prime_coop<SH, K + 1>(r, a, c, sw, lane);
    }
}


// ---- q6_K classifier rows in the persistent q4_0 kernels (TkShape::CLSQ6; q6k.h) ----------------------------------------------
// CU c owns the rows [c0, c0 + cn) it owns with a q4_0 classifier; wave w of 8 takes rows w, w + 8, ...: a row is ONE quad per
// lane (E <= 4096), TK_Q6_NB rows requested ahead -- the first ones before the final gather (weights do not depend on
// activations), like the ring's classifier slots.  x * gains comes from LDS in natural order (the final gather writes it so when
// CLSQ6), 64 floats per lane held for the whole phase; the row sums land in part[row] and the service wave divides them by the
// norm (llama2.f90:627-636) as it does for every other row type.
constexpr int TK_Q6_NB = 3;
template <class SH>
struct TkQ6 {
    Q6Quad b[TK_Q6_NB];
    __device__ __forceinline__ const char* row(const TokenArgs& a, int r) const {
        return static_cast<const char*>(a.wcls) + (size_t)(a.c0 + min(r, a.cn - 1)) * q6k_row_stride(SH::E);      // (clamped: every slot is five loads)
    }
    __device__ __forceinline__ void prime(const TokenArgs& a, int w, int lane) {
        constexpr int Q = SH::E / 64;
#pragma unroll
        for (int i = 0; i < TK_Q6_NB; ++i) q6k_load(b[i], row(a, w + i * TK_WAVES), min(lane, Q - 1), Q);
    }
    // xs: x * gains, f32, natural order (LDS).  Behind the barrier that follows the final gather.
    __device__ __forceinline__ void run(const TokenArgs& a, int w, int lane, const float* xs, float* part) {
        constexpr int Q = SH::E / 64;
        const int qd = min(lane, Q - 1);
        float x[64], sx32[4];
        q6k_load_x(x, sx32, xs, qd, lane < Q);
        for (int r = w; r < a.cn; r += TK_Q6_NB * TK_WAVES) {
#pragma unroll
            for (int i = 0; i < TK_Q6_NB; ++i) {
                const int ri = r + i * TK_WAVES;
                const float v = wave_sum(q6k_quad_dot(b[i], x, sx32, qd & 1));
                if (lane == 0 && ri < a.cn) part[ri] = v;
                q6k_load(b[i], row(a, ri + TK_Q6_NB * TK_WAVES), qd, Q);
            }
        }
    }
};

// ------------------------------------------------------------------------------------------------
// In-kernel attention for one head on one CU, all 16 waves (llama2.f90:572-598).  q_h, k_pos, v_pos
// arrive through the exchange (LDS); rows t < pos-1 come from the caches written by earlier launches.
// ------------------------------------------------------------------------------------------------
template <class SH>
struct TkAtt {
    static constexpr int HS = SH::HS, LPT = HS / 4, TPW = 64 / LPT, TPB = TK_WAVES * TPW, U = 8, TILE = TPB * U;
    float4 kv[U], vv[U];
    // K and V rows of the first TILE (= 256) timesteps (written by earlier launches) are requested at the START of
    // the layer's attention, long before q exists: by the time q arrives they have crossed the loaded memory system.
    // (The 64 registers are the two ring entries the QKV phase has just consumed and not yet refilled; an attention
    // CU streams no QKV / wo tiles, so nothing else of its own is queued ahead of the q poll.)
    // Only contexts longer than TILE pay exposed round trips.
    __device__ __forceinline__ void prefetch(const TokenArgs& a, int l, int h, int pos, int tid, int t0 = 0) {
        const int lane = tid & 63, wid = tid >> 6;
        const int g = h / SH::KVMUL, sub = lane % LPT, tl = lane / LPT;
        const __amdgpu_buffer_rsrc_t rk = tk_rsrc(a.kc + (size_t)l * a.S * SH::KV, a.S * SH::KV * 4);
        const __amdgpu_buffer_rsrc_t rv = tk_rsrc(a.vc + (size_t)l * a.S * SH::KV, a.S * SH::KV * 4);
        const int col = (g * HS + sub * 4) * 4;
        const int tb = t0 + wid * TPW + tl, tmax = max(pos - 2, 0);
#pragma unroll
        for (int u = 0; u < U; ++u) kv[u] = tk_ldkv(rk, min(u * TPB + tb, tmax) * (SH::KV * 4) + col);
#pragma unroll
        for (int u = 0; u < U; ++u) vv[u] = tk_ldkv(rv, min(u * TPB + tb, tmax) * (SH::KV * 4) + col);
    }
};

// Long contexts: a head's timesteps in up to HPC = TK_NCU / NH parts, part 0 on the head's attention CU and part p on the
// p-th other CU of the head's group of HPC (an ordinary row-owning CU, idle during the attention hop anyway).  One part per
// TILE (256) timesteps, so a part's K/V rows are all requested before q arrives, as at short contexts; contexts of <= TILE
// timesteps are one part: nothing changes for them.  Each part is a complete softmax over its own timesteps (its own
// maximum and sum); the head's CU merges them (tk_service).  Part p covers [t0, t1) of the pos timesteps.
template <class SH>
struct TkAttPlan {
    static constexpr int HPC = TK_NCU / SH::NH, TILE = TkAtt<SH>::TILE, TPB = TkAtt<SH>::TPB;
    static constexpr int PMAX = HPC < 8 ? HPC : 8;       // parts per head at most
    static constexpr int STEP = TILE;                    // one more part per STEP timesteps (a part per 128 / 64 / 32: measured
                                                         // slower, profiles/r03_att_part_step_sweep.txt)
    int P, chunk;
    __device__ __forceinline__ TkAttPlan(int pos) {
        P = min(PMAX, (pos + STEP - 1) / STEP);
        if (LLMK_TK_ATT_SPLIT == 0 || P < 1) P = 1;
        chunk = (((pos + P - 1) / P) + TPB - 1) / TPB * TPB;
        while (P > 1 && (P - 1) * chunk >= pos) --P;          // (cannot happen for pos > TILE * (P - 1); kept as a guard)
    }
    __device__ __forceinline__ int t0(int p) const { return p * chunk; }
    __device__ __forceinline__ int t1(int p, int pos) const { return min((p + 1) * chunk, pos); }
};
// this CU's part of its head's timesteps: false = none (recomputed per layer by the streaming waves rather than kept in
// scalar registers across the layer loop: the kernels sit at the register ceiling)
template <class SH>
__device__ __forceinline__ bool tk_att_role(int c, int pos, int& t0, int& t1) {
    constexpr int HPC = TK_NCU / SH::NH;
    const int apart = ((c % HPC) - ((c / HPC / SH::KVMUL) % HPC) + HPC) % HPC;
    if (pos <= TkAttPlan<SH>::STEP || LLMK_TK_ATT_SPLIT == 0) { t0 = 0; t1 = pos; return apart == 0; }
    const TkAttPlan<SH> plan(pos);
    t0 = plan.t0(apart);
    t1 = plan.t1(apart, pos);
    return apart < plan.P;
}
// granules of part p >= 1 of head h: HS output dims, then the part's maximum and sum
template <class SH>
__device__ __forceinline__ unsigned long long* tk_g_part(const TokenArgs& a, int h, int p) {
    return tk_g_x<SH>(a) + SH::E + ((size_t)h * (TkAttPlan<SH>::PMAX - 1) + (p - 1)) * (SH::HS + 2);
}

// sum over the LPT = HS/4 lanes that share a timestep: one DPP row (head size 64) or two (head size 128: row_bcast:15
// adds the even row's total into the odd row); valid in the LAST lane of the group
template <int LPT>
__device__ __forceinline__ float tstep_sum(float v) {
    v = row16_sum(v);
    if constexpr (LPT == 32) v += dpp_mov<0x142, 0xa, true>(0.f, v);
    return v;
}

template <class SH>
__device__ __forceinline__ void tk_attention(const TokenArgs& a, char* lds, int l, int h, int pos, int tid, TkAtt<SH>& pa,
                                             unsigned long long* dbg = nullptr, int t0 = 0, int t1 = -1, float* m_out = nullptr,
                                             float* s_out = nullptr) {
    if (t1 < 0) t1 = pos;           // timesteps [t0, t1) of the pos in all (TkAttPlan); t1 == pos includes this token's own
    constexpr int HS = SH::HS, LPT = HS / 4, TPW = 64 / LPT, TPB = TK_WAVES * TPW, U = TkAtt<SH>::U, TILE = TPB * U;
    static_assert(LPT == 16 || LPT == 32, "a timestep is one or two DPP rows");
    float4 (&kv)[U] = pa.kv;
    float4 (&vv)[U] = pa.vv;
    const int lane = tid & 63, wid = tid >> 6;
    const int g = h / SH::KVMUL;
    const float* qs = reinterpret_cast<const float*>(lds + TkLds<SH>::ATT_Q);
    const float4* kcur = reinterpret_cast<const float4*>(qs + HS);
    const float4* vcur = reinterpret_cast<const float4*>(qs + 2 * HS);
    float4* red = reinterpret_cast<float4*>(lds + TkLds<SH>::ATT_RED);   // [waves][TPW][LPT] float4
    float* att = reinterpret_cast<float*>(lds + TkLds<SH>::ATT_S);
    const int sub = lane % LPT, tl = lane / LPT;
    const float4 qv = reinterpret_cast<const float4*>(qs)[sub];
    const float scale = sqrtf((float)HS);
    const __amdgpu_buffer_rsrc_t rk = tk_rsrc(a.kc + (size_t)l * a.S * SH::KV, a.S * SH::KV * 4);
    const __amdgpu_buffer_rsrc_t rv = tk_rsrc(a.vc + (size_t)l * a.S * SH::KV, a.S * SH::KV * 4);
    constexpr int rowb = SH::KV * 4;                 // bytes per cached timestep
    const int col = (g * HS + sub * 4) * 4;
    const int tb = wid * TPW + tl;
    const int npast = pos - 1;  // rows 0..pos-2 live in the cache; row pos-1 is this token's (LDS)
    const int tmax = max(npast - 1, 0);
    const int tcache = min(npast, t1);

    float* ex = att + a.S;                                                      // exp(score - max), written per wave
    float* pw = reinterpret_cast<float*>(lds + TkLds<SH>::ATT_P) + wid * 32;    // this wave's U*TPW (<= 32) weights of a batch
    for (int base = t0; base < t1; base += TILE) {
        if (base > t0) {
#pragma unroll
            for (int u = 0; u < U; ++u) kv[u] = tk_ldkv(rk, min(base + u * TPB + tb, tmax) * rowb + col);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (base + u * TPB < t1) {   // block-uniform: short contexts skip the empty batches
                const int t = base + u * TPB + tb;
                // rows >= pos-1 read a clamped (wrong) row: t = pos-1 is redone from LDS below, later ones are never used
                const float d = tstep_sum<LPT>(dot4(qv, kv[u], 0.f));
                if (sub == LPT - 1 && t < tcache) att[t] = d / scale;              // :582
            }
        }
    }
    if (t1 == pos) {   // this token's own key never went through the cache
        const float d = tstep_sum<LPT>(dot4(qv, kcur[sub], 0.f));
        if (wid == 0 && lane == LPT - 1) att[npast] = d / scale;
    }
    if (TK_DEBUG && dbg) dbg[0] = wall_clock64();
    tk_barrier();
    if (TK_DEBUG && dbg) dbg[1] = wall_clock64();
    // every wave folds max and sum over ALL scores itself: no cross-wave reduction, no extra barriers.  exp() is
    // evaluated once per score here (each wave keeps its own copy of what it wrote: same values, benign overlap)
    float m = -INFINITY;
    for (int t = t0 + lane; t < t1; t += WAVE) m = fmaxf(m, att[t]);
    m = wave_max(m);
    if (TK_DEBUG && dbg) dbg[3] = wall_clock64();
    float s = 0.f;
    for (int t = t0 + lane; t < t1; t += WAVE) {
        const float e = expf(att[t] - m);
        ex[t] = e;
        s += e;
    }
    s = wave_sum(s);
    if (m_out) { *m_out = m; *s_out = s; }
    if (TK_DEBUG && dbg) dbg[4] = wall_clock64();

    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int base = t0; base < t1; base += TILE) {
        if (base > t0) {
#pragma unroll
            for (int u = 0; u < U; ++u) vv[u] = tk_ldkv(rv, min(base + u * TPB + tb, tmax) * rowb + col);
        }
        // xi/sum(xi) (:476) once per timestep: lane i < U*TPW owns timestep (u = i / TPW, tl = i % TPW) of this wave
        if (lane < U * TPW) {
            const int t = base + (lane / TPW) * TPB + wid * TPW + (lane % TPW);
            pw[lane] = (t < t1) ? ex[t] / s : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (base + u * TPB < t1) {
                const int t = base + u * TPB + tb;
                const float4 v4 = (t == npast) ? vcur[sub] : vv[u];
                const float p = pw[u * TPW + tl];
                acc.x = fmaf(p, v4.x, acc.x);
                acc.y = fmaf(p, v4.y, acc.y);
                acc.z = fmaf(p, v4.z, acc.z);
                acc.w = fmaf(p, v4.w, acc.w);
            }
        }
    }
    red[(wid * TPW + tl) * LPT + sub] = acc;   // the service wave folds the waves*TPW partials per dim
    if (TK_DEBUG && dbg) dbg[2] = wall_clock64();
}

// ------------------------------------------------------------------------------------------------
// SERVICE wave of a CU: everything that depends on other CUs.  Per phase: gather the input vector
// (granule sweep), rmsnorm, barrier A, barrier B, epilogue + publish.
// ------------------------------------------------------------------------------------------------
// the 0-based token of this launch: an argument / device word, or (GR) the fold of the previous launch's candidates
template <bool GR, int SHV>
__device__ __forceinline__ int tk_token(const TokenArgs& a, int c, int lane) {
    if constexpr (GR) {
        if (a.gflags & TKG_CAND_IN) {   // the previous launch's per-CU maxima -> its greedy token (plain loads: a kernel boundary lies between)
            const float2* cand_in = tk_cand(a, (a.pos_imm & 1) ^ 1);
            float bv = -INFINITY;
            int bi = 0x7fffffff;
#pragma unroll
            for (int k = 0; k < TK_NCU / WAVE; ++k) {     // ascending CU = ascending row ranges: '>' keeps the first maximum
                const float2 cd = cand_in[lane + k * WAVE];
                const int ci = __float_as_int(cd.y);
                if (cd.x > bv || (cd.x == bv && ci < bi)) { bv = cd.x; bi = ci; }
            }
            tk_wave_argmax(bv, bi);
            // No finite maximum among the candidates (all NaN / -inf: a bad upload, an overflow) leaves the seed index: raise
            // the sticky word instead of using it as an embedding row (advisor, round 3)
            const bool none = (unsigned)bi >= (unsigned)SHV;
            if (none) bi = 0;
            // The id of the PREVIOUS position is published only while the error word is clear: once a launch has timed out its
            // candidates are garbage, and so is everything folded from them.  The host then sees no id from that position on,
            // retires the token kernel and redoes exactly those positions on the multi-kernel path (llmk_decode_greedy).
            if (c == 0 && lane == 0) {
                if (none) atomicOr(a.err, 0x2000u);
                const unsigned e = __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((a.gflags & TKG_ID) && e == 0 && !none)
                    __hip_atomic_store(reinterpret_cast<int*>(a.herr + 4) + a.tok_imm, bi + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            return bi;
        }
    }
    return a.tokpos ? a.tokpos[0] : a.tok_imm;
}
// q4_0: row r of the CU's rows in a phase whose row groups are NCS units wide: the sum of its units' partials, ascending column
// slot (a fixed order: run-to-run identical); ug0 = the first unit GROUP of the range (w1|w3: the up groups follow the gate groups)
template <int NCS>
__device__ __forceinline__ float tk_unit_row(const float* part, int ug0, int r) {
    const float* p = part + (ug0 + (r >> 4)) * (NCS * Q16_ROWS) + (r & 15);
    float v = p[0];
#pragma unroll
    for (int cs = 1; cs < NCS; ++cs) v += p[cs * Q16_ROWS];
    return v;
}
// q4_0, layer 0: x = the embedding row (llama2.f90:520), xraw <- x, the image <- x * gains, returns sqrt(mean(x^2) + eps) (:450-457).
// One wave, four 1 KB pieces at a time: this wave holds a ring of units, there is no room for TkNorm's 64 registers of gains.
template <class SH>
// fits = false: a scaled element (or block sum) is beyond the f16 range -- the caller raises 0x4000 as the gathers of the later layers do
__device__ __forceinline__ float tk_stage_q16(const float* __restrict__ row, const float* __restrict__ gains, float* xraw, float* xs, int lane, float eps,
                                              float& psc, bool& fits) {
    float ss = 0.f, amax = 0.f;
    constexpr int PER = SH::E / (4 * WAVE);
    // first pass: the row's norm, for the power of two its image is scaled with (tk_pow2_inv); the second pass finds the row in L2
#pragma unroll 1
    for (int k0 = 0; k0 < PER; k0 += 4) {
        float4 x[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] = reinterpret_cast<const float4*>(row)[lane + (k0 + i) * WAVE];
#pragma unroll
        for (int i = 0; i < 4; ++i) ss = dot4(x[i], x[i], ss);
    }
    ss = wave_sum(ss);
    const float xn = sqrtf(ss / (float)SH::E + eps);
    psc = tk_pow2_inv(xn);
#pragma unroll 1
    for (int k0 = 0; k0 < PER; k0 += 4) {
        float4 x[4], w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            x[i] = reinterpret_cast<const float4*>(row)[lane + (k0 + i) * WAVE];
            w[i] = reinterpret_cast<const float4*>(gains)[lane + (k0 + i) * WAVE];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            reinterpret_cast<float4*>(xraw)[lane + (k0 + i) * WAVE] = x[i];
            const float4 o = make_float4(x[i].x * w[i].x * psc, x[i].y * w[i].y * psc, x[i].z * w[i].z * psc, x[i].w * w[i].w * psc);
            const int e = 4 * (lane + (k0 + i) * WAVE);
            q16_put4(reinterpret_cast<char*>(xs), e, o);
            float bs = (o.x + o.y) + (o.z + o.w);
            bs += dpp_mov<0xB1, 0xf, true>(0.f, bs);
            bs += dpp_mov<0x4E, 0xf, true>(0.f, bs);
            bs += dpp_mov<0x114, 0xf, true>(0.f, bs);               // row_shr:4: lanes 4..7 / 12..15 hold their block's sum
            if ((lane & 7) == 7) q16_put_sum<SH::NBI>(reinterpret_cast<char*>(xs) + Q16Img<SH::NBI>::SUM + (e >> 5) * 2, bs * (1.0f / Q16_SUM_DIV));
            amax = fmaxf(fmaxf(amax, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
        }
    }
    fits = !__any(!(amax < 60000.f));
    return xn;
}
// q4_0, layer 0 on an ATTENTION CU: it stages no QKV input, but its first scaled gather (layer 0's xa) wants the same power of two
// the row-owning CUs take from the embedding row's rmsnorm (advisor, round 5: it was 1.0 there)
template <class SH>
__device__ __forceinline__ float tk_row_norm(const float* __restrict__ row, int lane, float eps) {
    float ss = 0.f;
    constexpr int PER = SH::E / (4 * WAVE);
#pragma unroll 1
    for (int k0 = 0; k0 < PER; k0 += 4) {
        float4 x[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] = reinterpret_cast<const float4*>(row)[lane + (k0 + i) * WAVE];
#pragma unroll
        for (int i = 0; i < 4; ++i) ss = dot4(x[i], x[i], ss);
    }
    ss = wave_sum(ss);
    return sqrtf(ss / (float)SH::E + eps);
}
// GR: the pipelined-greedy variant (token from the previous launch's candidates, candidates of its own).  A separate
// instantiation because the f32 kernel sits at the register ceiling: with the candidate code compiled in, hipcc spills 20
// bytes per lane in the STREAMING waves' loop (one s_waitcnt vmcnt(0) + scratch store per slot: the ring drains).
template <class SH, bool GR>
__device__ __forceinline__ void tk_service(const TokenArgs& a, char* lds, int c, int lane_in, int tid) {
    int lane = lane_in;     // (q4_0: made opaque once per layer, so that what is derived from it is recomputed there, not carried across the ring)
    typedef TkLds<SH> LD;
    float* xs = reinterpret_cast<float*>(lds + LD::XS);
    float* xraw = reinterpret_cast<float*>(lds + LD::XRAW);
    float* part = reinterpret_cast<float*>(lds + LD::PART);
    // q4_0: the streaming input is staged as an f16 hi | lo image (TkLds); 0 = natural order
    constexpr int TR_E = SH::TR_E, TR_H = SH::TR_H;
    typedef TkSched<SH> SC;
    // q4_0: this wave takes units like the seven others (not on an attention CU: nw = 7); the row sums of a phase are then
    // spread over its units' partials (tk_unit_row)
    [[maybe_unused]] TkRing<SH> r;
    const char* img = lds + LD::XS;
    if constexpr (SH::Q4) tk_prime_coop<SH, 0>(r, a, c, TK_NS, lane);
    float* red8 = reinterpret_cast<float*>(lds + LD::RED8);     // COOP: per-wave partial sums of squares
    volatile int* gflag = reinterpret_cast<volatile int*>(lds + LD::RED8 + 32);   // LLMK_TK_GF: gathers issued so far on this CU
    constexpr bool GF = LLMK_TK_GF && !SH::COOP;
    constexpr bool GCD = SH::COOP && LLMK_TK_COOP_DELAY > 0;      // coop: the same head start for the producers
    constexpr int GCDN = LLMK_TK_COOP_DELAY;
    if (GF) { tk_flag_set(gflag, -1, lane); tk_flag_set(gflag + 1, -1, lane); }
    const int L = a.L;
    const int tok = tk_token<GR, SH::V>(a, c, lane);
    const int pos = a.tokpos ? a.tokpos[1] : a.pos_imm;
    const unsigned ebase = (unsigned)(a.tokpos ? a.tokpos[2] : a.serial_imm) * (unsigned)(5 * L + 2);
    constexpr int HPC = TK_NCU / SH::NH;
    // head h runs on CU h*HPC + (its kv group mod HPC): with the dispatcher placing block b on XCD b % 8 the heads
    // of one kv group share an XCD (one L2 copy of their K/V rows) and different groups use different XCDs
    const bool att_cu = (c % HPC) == ((c / HPC / SH::KVMUL) % HPC);
    const int my_head = c / HPC;
    bool ok = true;
    // RoPE angles depend on pos only: cos/sin once per token, not once per layer        :544-548
    float* rope_cs = reinterpret_cast<float*>(lds + LD::ROPE);
    if (lane < SH::HS / 2) {
        const float rval = (float)pos * a.rope[lane];
        rope_cs[lane] = cosf(rval);
        rope_cs[SH::HS / 2 + lane] = sinf(rval);
    }
    unsigned long long* tr = (TK_DEBUG && tk_trace(a)) ? tk_trace(a) + (size_t)c * TK_TRACE_N : nullptr;
    const bool nosync = TK_DEBUG && (a.gflags & TKG_NOSYNC) != 0;
#define TK_STAMP(i) do { if (tr && lane == 0 && l < 64) tr[l * 16 + (i)] = wall_clock64(); } while (0)

    for (int l = 0; l < L; ++l) {
        const unsigned e_q = ebase + 5u * l + 1, e_att = e_q + 1, e_o = e_q + 2, e_a = e_q + 3, e_d = e_q + 4;
        if constexpr (SH::Q4) asm volatile("" : "+v"(lane));
        TK_STAMP(0);

        // ---- P0: rmsnorm + QKV + RoPE                                            llama2.f90:527-565
        TkNorm<SH::Q4 ? 4 * WAVE : SH::E> nrm;      // (q4_0: layer 0 is staged in pieces, tk_stage_q16: this wave holds a ring now)
        const bool coop0 = SH::COOP && l > 0;       // layer 0 starts from the embedding row: no exchange, this wave alone
        if constexpr (!SH::Q4) { if (!att_cu && !coop0) nrm.prefetch(tk_rms_att(a, l, SH::E), lane); }
        float xn_att = 1.f;
        [[maybe_unused]] float sc_att = 1.f, sc_ffn = 1.f;      // q4_0: the power of two the gathered image was scaled with (tk_pow2_inv)
        if (GF && att_cu) { tk_flag_set(gflag, 4 * l + 1, lane); if (SH::GF_PUB) tk_flag_set(gflag + 1, 4 * l + 1, lane); }   // no x and no xb gather on this CU: nothing for its bursts to wait for
        if (!att_cu) {   // an attention CU owns no QKV rows: it goes straight to the q poll
            if (coop0) {
                if constexpr (GCD) __builtin_amdgcn_s_sleep(GCDN);
                ok = tk_coop_gather<SH::E, TR_E, true>(tk_g_x<SH>(a), e_q - 1, xraw, xs, tk_rms_att(a, l, SH::E), red8, a.err, TK_NS, lane, nosync) && ok;
            } else if (SH::Q4) {
                bool fits;
                xn_att = tk_stage_q16<SH>(a.emb + (size_t)tok * SH::E, tk_rms_att(a, 0, SH::E), xraw, xs, lane, a.eps, sc_att, fits);   // :520, :527
                if (!fits && !nosync) {      // an embedding row times its gains that no f16 holds: the same sticky word as a gather's (tk_coop_part)
                    if (lane == 0) __hip_atomic_store(a.err, 0x4000u + (e_q & 0xff), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    tk_evt_note(reinterpret_cast<unsigned*>(red8) + TK_EVT, e_q, lane);
                    ok = false;
                }
            } else if (l == 0) {
#pragma unroll 8
                for (int k = 0; k < SH::E / WAVE; ++k) xraw[lane + k * WAVE] = a.emb[(size_t)tok * SH::E + lane + k * WAVE];  // :520
            } else {
                if constexpr (GF && SH::GF_DELAY > 0) __builtin_amdgcn_s_sleep(SH::GF_DELAY);
                ok = tk_gather<SH::E, 0, LLMK_TK_E_NL>(tk_g_x<SH>(a), e_q - 1, xraw, a.err, lane, nosync, (tr && l < 32) ? tr + (32 + l) * 16 + 0 : nullptr, gflag, 4 * l) && ok;
            }
            TK_STAMP(1);
            if constexpr (!SH::Q4) { if (!coop0) xn_att = nrm.template apply<TR_E>(xraw, xs, lane, a.eps); }
        }
        tk_barrier();
        if (coop0 && !att_cu) xn_att = tk_coop_xn<SH::E>(red8, a.eps);
        if constexpr (SH::Q4) {
            // every wave has written its slice with the scale it found; the next residual-stream gather on this CU uses this norm's
            if (!att_cu) {
                if (coop0) sc_att = red8[TK_XSC];
                if (lane == 0) red8[TK_XSC] = tk_pow2_inv(xn_att);
                xn_att *= sc_att;                       // (exact: the row sums are those of x * gains * 2^-e)
            } else if (l == 0) {                         // an attention CU's first scaled gather is layer 0's xa: the embedding row's norm
                const float xn0 = tk_row_norm<SH>(a.emb + (size_t)tok * SH::E, lane, a.eps);
                if (lane == 0) red8[TK_XSC] = tk_pow2_inv(xn0);
            }
        }
        TK_STAMP(2);
        if constexpr (SH::Q4) tk_steps<SH, SC::KQ, SH::SL_Q, false>(r, a, l, c, TK_NS, img, part, lane);
        tk_barrier();
        TK_STAMP(3);
        if (lane < a.qn) {
            float v0, v1;
            if constexpr (SH::Q4) { v0 = tk_unit_row<SH::NCS_E>(part, 0, lane); v1 = tk_unit_row<SH::NCS_E>(part, 0, lane ^ 1); }
            else { v0 = part[lane]; v1 = part[lane ^ 1]; }
            v0 = v0 / xn_att;
            v1 = v1 / xn_att;
            const int r = a.q0 + lane;
            float outv = v0;
            if (r < SH::E + SH::KV) {
                // pairs (i,i+1); 1-based odd i -> head_dim = mod(i,hs) = 2j+1 (table index j)   :543-559
                const int i0 = ((r < SH::E) ? r : r - SH::E) & ~1;
                const int jf = (i0 % SH::HS) >> 1;
                const float fcr = rope_cs[jf], fci = rope_cs[SH::HS / 2 + jf];
                // even lane: own = q0, partner = q1; odd lane: own = q1, partner = q0
                outv = (lane & 1) ? (v1 * fci + v0 * fcr) : (v0 * fcr - v1 * fci);
                if (r >= SH::E) a.kc[((size_t)l * a.S + (pos - 1)) * SH::KV + (r - SH::E)] = outv;          // :564
            } else {
                a.vc[((size_t)l * a.S + (pos - 1)) * SH::KV + (r - SH::E - SH::KV)] = outv;                 // :565
            }
            tk_publish(a.g_qkv + r, e_q, outv);
        }
        if (GF && SH::GF_PUB && !att_cu) tk_flag_set(gflag + 1, 4 * l + 1, lane);
        TK_STAMP(4);
        // ---- P1: attention, one CU per head                                     llama2.f90:572-598
        // long contexts: this CU's part of its head's timesteps (TkAttPlan; part 0 = the attention CU, one part at <= 256
        // timesteps).  Recomputed per layer, as in the streaming waves: nothing of it lives across the layer loop.
        int at0, at1;
        if (tk_att_role<SH>(c, pos, at0, at1)) {
            const TkAttPlan<SH> plan(pos);
            const int apart = ((c % HPC) - ((c / HPC / SH::KVMUL) % HPC) + HPC) % HPC;
            float* qs = reinterpret_cast<float*>(lds + LD::ATT_Q);
            const int g = my_head / SH::KVMUL;
            TkAtt<SH> pa;
            pa.prefetch(a, l, my_head, pos, tid, at0);   // K/V rows cross the memory system while q is awaited
            for (unsigned spin = 0;; ++spin) {
                bool good = true;
#pragma unroll
                for (int d0 = 0; d0 < SH::HS; d0 += WAVE) {   // HS granules each of q_h, k, v: one (head size 64) or two per lane
                    const unsigned long long xq = __hip_atomic_load(a.g_qkv + my_head * SH::HS + d0 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned long long xk = __hip_atomic_load(a.g_qkv + SH::E + g * SH::HS + d0 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned long long xv = __hip_atomic_load(a.g_qkv + SH::E + SH::KV + g * SH::HS + d0 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    good = good && (unsigned)(xq >> 32) == e_q && (unsigned)(xk >> 32) == e_q && (unsigned)(xv >> 32) == e_q;
                    qs[d0 + lane] = __uint_as_float((unsigned)xq);
                    qs[SH::HS + d0 + lane] = __uint_as_float((unsigned)xk);
                    qs[2 * SH::HS + d0 + lane] = __uint_as_float((unsigned)xv);
                }
                if (__all(good) || nosync) break;
                if ((spin & 63) == 63) {
                    if (__hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { ok = false; break; }
                    if (spin > TK_SPIN_LIMIT) {
                        if (lane == 0) __hip_atomic_store(a.err, 0x200u + (unsigned)l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ok = false; break;
                    }
                }
                __builtin_amdgcn_s_sleep(LLMK_TK_POLL_SLEEP);
            }
            TK_STAMP(5);
            tk_barrier();
            float pm, ps;       // this part's maximum and sum of exponentials
            tk_attention<SH>(a, lds, l, my_head, pos, tid, pa, (tr && lane == 0 && l < 22) ? tr + (32 + l) * 16 + 10 : nullptr, at0, at1, &pm, &ps);
            tk_barrier();
            TK_STAMP(6);
            // fold the waves*TPW partial output vectors: lane = output dim, one conflict-free ds_read_b32 per partial
            const float* redf = reinterpret_cast<const float*>(lds + LD::ATT_RED);
            constexpr int ND = SH::HS / WAVE, NPH = TkAttPlan<SH>::PMAX - 1;
            float o[ND];
#pragma unroll
            for (int d = 0; d < ND; ++d) {
                const int d0 = d * WAVE;
                float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
#pragma unroll
                for (int w = 0; w < TK_WAVES * (256 / SH::HS); w += 4) {
                    o0 += redf[(w + 0) * SH::HS + d0 + lane];
                    o1 += redf[(w + 1) * SH::HS + d0 + lane];
                    o2 += redf[(w + 2) * SH::HS + d0 + lane];
                    o3 += redf[(w + 3) * SH::HS + d0 + lane];
                }
                o[d] = (o0 + o1) + (o2 + o3);
            }
            if (plan.P == 1) {
#pragma unroll
                for (int d = 0; d < ND; ++d) tk_publish(tk_g_xb<SH>(a) + my_head * SH::HS + d * WAVE + lane, e_att, o[d]);
            } else if (apart > 0) {
                // a part's softmax-weighted sum of V rows over ITS timesteps, with its maximum and sum: merged by the head's CU
                unsigned long long* gp = tk_g_part<SH>(a, my_head, apart);
#pragma unroll
                for (int d = 0; d < ND; ++d) tk_publish(gp + d * WAVE + lane, e_att, o[d]);
                if (lane < 2) tk_publish(gp + SH::HS + lane, e_att, lane == 0 ? pm : ps);
            } else {
                // the head's CU: sum_p w_p o_p / sum_p w_p with w_p = s_p exp(m_p - M), M the maximum over all parts
                // (llama2.f90:466-478 evaluated in parts: exp(x - m_p) exp(m_p - M) for exp(x - M))
                // (the other parts are fetched G at a time: one round trip for up to G of them; head size 128 takes two rounds
                // from 6 parts on -- the q4_0 kernel has no registers for all seven at once)
                constexpr int G = ND == 1 ? NPH : 4;
                float M = pm, W = ps;
#pragma unroll
                for (int d = 0; d < ND; ++d) o[d] *= ps;
                for (int p0 = 1; p0 < plan.P && ok; p0 += G) {
                    float xv[G][ND], xm[G];
                    for (unsigned spin = 0;; ++spin) {
                        bool good = true;
#pragma unroll
                        for (int i = 0; i < G; ++i) {
                            const bool live = p0 + i < plan.P;
                            const unsigned long long* gq = tk_g_part<SH>(a, my_head, live ? p0 + i : 1);
#pragma unroll
                            for (int d = 0; d < ND; ++d) {
                                const unsigned long long x = __hip_atomic_load(gq + d * WAVE + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                good = good && (!live || (unsigned)(x >> 32) == e_att);
                                xv[i][d] = __uint_as_float((unsigned)x);
                            }
                            const unsigned long long y = __hip_atomic_load(gq + SH::HS + (lane & 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            good = good && (!live || (unsigned)(y >> 32) == e_att);
                            xm[i] = __uint_as_float((unsigned)y);
                        }
                        if (__all(good) || nosync) break;
                        if ((spin & 63) == 63) {
                            if (__hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { ok = false; break; }
                            if (spin > TK_SPIN_LIMIT) {
                                if (lane == 0) __hip_atomic_store(a.err, 0x300u + (unsigned)l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                ok = false; break;
                            }
                        }
                        __builtin_amdgcn_s_sleep(LLMK_TK_POLL_SLEEP);
                    }
#pragma unroll
                    for (int i = 0; i < G; ++i) {
                        if (p0 + i < plan.P) {
                            const float mp = __shfl(xm[i], 0, WAVE), sp = __shfl(xm[i], 1, WAVE);
                            const float Mn = fmaxf(M, mp), f0 = expf(M - Mn), f1 = expf(mp - Mn) * sp;
#pragma unroll
                            for (int d = 0; d < ND; ++d) o[d] = o[d] * f0 + xv[i][d] * f1;
                            W = W * f0 + f1;
                            M = Mn;
                        }
                    }
                }
#pragma unroll
                for (int d = 0; d < ND; ++d) tk_publish(tk_g_xb<SH>(a) + my_head * SH::HS + d * WAVE + lane, e_att, o[d] / W);
            }
        }
        // ---- P2: x += wo . xb                                                    llama2.f90:603-605
        if constexpr (SH::Q4 && LLMK_TK_ADV_HOP != 0) tk_request<SH, SC::KO, false>(r, a, l, c, TK_NS, lane);   // (as the other seven waves: tk_stream_coop)
        if constexpr (SH::COOP) {
            // (this wave's slice is the last eighth of the vector: heads (E - E/8) / HS onwards)
            if (!att_cu) ok = tk_coop_gather<SH::E, TR_E, false, SH::Q4 ? 1 : 0>(tk_g_xb<SH>(a), e_att, nullptr, xs, nullptr, red8, a.err, TK_NS, lane, nosync) && ok;
        } else {
            if (!att_cu) ok = tk_gather<SH::E, TR_E, LLMK_TK_XB_NL>(tk_g_xb<SH>(a), e_att, xs, a.err, lane, nosync, (tr && l < 32) ? tr + (32 + l) * 16 + 2 : nullptr, gflag, 4 * l + 1) && ok;
        }
        TK_STAMP(7);
        tk_barrier();
        if constexpr (SH::Q4) tk_steps<SH, SC::KO, SH::SL_O, false, LLMK_TK_ADV_HOP == 0>(r, a, l, c, TK_NS, img, part, lane);
        tk_barrier();
        TK_STAMP(8);
        if (lane < a.on) {
            // q4_0: the row sums are those of xb 2^e (TK_XSC_B): divided by the same power of two, exactly
            const float v = SH::Q4 ? tk_unit_row<SH::NCS_E>(part, 0, lane) / red8[TK_XSC_B] : part[lane];
            const int r = a.o0 + lane;
            tk_publish(tk_g_xa<SH>(a) + r, e_o, xraw[r] + v);
        }
        if constexpr (SH::Q4) {      // the next layer's xb image: its own record of the position before, else from this layer's maximum
            if (!att_cu && lane == 0) {
                const float am = tk_amax_true(red8[TK_XSC_B], red8 + TK_AMX_B);
                // (no record BEHIND a range event: from then on the images -- and every maximum taken from them -- are whatever the waves that
                // gave up left in LDS.  The vector that raised the event itself is recorded -- its maximum is what the next position needs)
                if (c == 1 && l < TK_QSC_LMAX && tk_rec_ok(red8, e_att)) {
                    float4* rec = tk_qsc<SH>(a, pos) + l; rec->x = am; rec->z = __int_as_float(pos);
                }
                red8[TK_XSC_B] = tk_scale_for(tk_predict(am, tk_hist_amax(reinterpret_cast<const float4*>(lds + LD::QSC), l + 1, L, 0, pos)), red8[TK_XSC_B]);
            }
        }
        if (GF && SH::GF_PUB) tk_flag_set(gflag + 1, 4 * l + 2, lane);
        // ---- P3: rmsnorm + w1|w3 + SwiGLU                                        llama2.f90:608-616
        float xn_ffn;
        if constexpr (SH::COOP) {
            if constexpr (GCD) __builtin_amdgcn_s_sleep(GCDN);
            ok = tk_coop_gather<SH::E, TR_E, true>(tk_g_xa<SH>(a), e_o, xraw, xs, tk_rms_ffn(a, l, SH::E), red8, a.err, TK_NS, lane, nosync) && ok;
            TK_STAMP(9);
            tk_barrier();
            xn_ffn = tk_coop_xn<SH::E>(red8, a.eps);
            if constexpr (SH::Q4) {
                sc_ffn = red8[TK_XSC];
                if (lane == 0) red8[TK_XSC] = tk_pow2_inv(xn_ffn);
                xn_ffn *= sc_ffn;
            }
            tk_steps<SH, SC::KA, SH::SL_A, false>(r, a, l, c, TK_NS, img, part, lane);
        } else {
            nrm.prefetch(tk_rms_ffn(a, l, SH::E), lane);
            if constexpr (GF && SH::GF_DELAY > 0) __builtin_amdgcn_s_sleep(SH::GF_DELAY);
            ok = tk_gather<SH::E, 0, LLMK_TK_E_NL>(tk_g_xa<SH>(a), e_o, xraw, a.err, lane, nosync, (tr && l < 32) ? tr + (32 + l) * 16 + 4 : nullptr, gflag, 4 * l + 2) && ok;
            TK_STAMP(9);
            xn_ffn = nrm.template apply<TR_E>(xraw, xs, lane, a.eps);
            tk_barrier();
        }
        TK_STAMP(10);
        tk_barrier();
        TK_STAMP(11);
        if (lane < (SH::Q4 ? tk_q4(a).hn : SH::R_A / 2)) {
            // q4_0: the gate groups' units come first, then the up groups' (tk_at)
            float gsum = SH::Q4 ? tk_unit_row<SH::NCS_E>(part, 0, lane) : part[2 * lane];
            float usum = SH::Q4 ? tk_unit_row<SH::NCS_E>(part, tk_q4(a).hn / Q16_ROWS, lane) : part[2 * lane + 1];
            gsum = gsum / xn_ffn;
            usum = usum / xn_ffn;
            const float hb = gsum * (1.0f / (1.0f + expf(-gsum)));
            tk_publish(tk_g_hb<SH>(a) + (SH::Q4 ? tk_q4(a).h0 : c * (SH::R_A / 2)) + lane, e_a, hb * usum);
        }
        if (GF && SH::GF_PUB) tk_flag_set(gflag + 1, 4 * l + 3, lane);
        // ---- P4: x += w2 . hb                                                    llama2.f90:618-620
        if constexpr (SH::COOP) {
            if constexpr (GCD) __builtin_amdgcn_s_sleep(GCDN);
            ok = tk_coop_gather<SH::H, TR_H, false, SH::Q4 ? 2 : 0>(tk_g_hb<SH>(a), e_a, nullptr, xs, nullptr, red8, a.err, TK_NS, lane, nosync) && ok;
        }
        else {
            if constexpr (GF && SH::GF_DELAY > 0) __builtin_amdgcn_s_sleep(SH::GF_DELAY);
            ok = tk_gather<SH::H, TR_H, SH::HB_NL, SH::GF_LAST>(tk_g_hb<SH>(a), e_a, xs, a.err, lane, nosync, (tr && l < 32) ? tr + (32 + l) * 16 + 6 : nullptr, gflag, 4 * l + 3) && ok;
        }
        TK_STAMP(12);
        tk_barrier();
        TK_STAMP(13);
        if constexpr (SH::Q4) tk_steps<SH, SC::KD, SC::SLP - SC::KD, false>(r, a, l, c, TK_NS, img, part, lane);
        tk_barrier();
        TK_STAMP(14);
        if (lane < (SH::Q4 ? tk_q4(a).dn : SH::R_D)) {
            float v = 0.f;
            if constexpr (SH::Q4) v = tk_unit_row<SH::NCS_H>(part, 0, lane) / red8[TK_XSC_H];      // (the hb image's power of two: TK_XSC_H)
            else {
#pragma unroll
            for (int p = 0; p < SH::TPR_H; ++p) v += part[lane * SH::TPR_H + p];
            }
            const int r = (SH::Q4 ? tk_q4(a).d0 : c * SH::R_D) + lane;
            tk_publish(tk_g_x<SH>(a) + r, e_d, xraw[r] + v);
        }
        if constexpr (SH::Q4) {
            if (lane == 0) {
                const float am = tk_amax_true(red8[TK_XSC_H], red8 + TK_AMX_H);
                if (c == 1 && l < TK_QSC_LMAX && tk_rec_ok(red8, e_a)) {
                    float4* rec = tk_qsc<SH>(a, pos) + l; rec->y = am; rec->w = __int_as_float(pos);
                }
                red8[TK_XSC_H] = tk_scale_for(tk_predict(am, tk_hist_amax(reinterpret_cast<const float4*>(lds + LD::QSC), l + 1, L, 1, pos)), red8[TK_XSC_H]);
            }
        }
        if (GF && SH::GF_PUB) tk_flag_set(gflag + 1, 4 * l + 4, lane);
        TK_STAMP(15);
    }
#undef TK_STAMP
    // ---- final rmsnorm + classifier                                             llama2.f90:627-636
    float xn_fin;
    if constexpr (SH::CLSQ6) {
        // q6_K classifier rows: x * gains in natural order (f32, unscaled) instead of the image; this wave's first rows go out first
        TkQ6<SH> q6;
        q6.prime(a, TK_NS, lane);
        if constexpr (GCD) __builtin_amdgcn_s_sleep(GCDN);
        ok = tk_coop_gather<SH::E, 0, true>(tk_g_x<SH>(a), ebase + 5u * L, xraw, xs, tk_rms_final(a, SH::E), red8, a.err, TK_NS, lane, nosync) && ok;
        tk_barrier();
        xn_fin = tk_coop_xn<SH::E>(red8, a.eps);
        q6.run(a, TK_NS, lane, xs, part);
    } else if constexpr (SH::COOP) {
        if constexpr (GCD) __builtin_amdgcn_s_sleep(GCDN);
        ok = tk_coop_gather<SH::E, TR_E, true>(tk_g_x<SH>(a), ebase + 5u * L, xraw, xs, tk_rms_final(a, SH::E), red8, a.err, TK_NS, lane, nosync) && ok;
        tk_barrier();
        xn_fin = tk_coop_xn<SH::E>(red8, a.eps);
        if constexpr (SH::Q4) xn_fin *= red8[TK_XSC];
        tk_steps<SH, 0, SH::SL_C, true>(r, a, L, c, TK_NS, img, part, lane);
    } else {
        TkNorm<SH::E> nrmf;
        nrmf.prefetch(tk_rms_final(a, SH::E), lane);
        if constexpr (GF && SH::GF_DELAY > 0) __builtin_amdgcn_s_sleep(SH::GF_DELAY);
        ok = tk_gather<SH::E>(tk_g_x<SH>(a), ebase + 5u * L, xraw, a.err, lane, nosync, nullptr, gflag, 4 * L) && ok;
        xn_fin = nrmf.template apply<TR_E>(xraw, xs, lane, a.eps);
        tk_barrier();
    }
    tk_barrier();
    const int cn = SH::CX ? a.cn : SH::R_C, c0 = SH::CX ? a.c0 : c * SH::R_C;
    auto logit = [&](int j) { return ((SH::Q4 && !SH::CLSQ6) ? tk_unit_row<SH::NCS_E>(part, 0, j) : part[j]) / xn_fin; };
    for (int j = lane; j < cn; j += WAVE) a.logits[c0 + j] = logit(j);
    if constexpr (GR) {
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        for (int j = lane; j < cn; j += WAVE) {       // the same quotient as the stored logit; ascending j per lane
            const float v = logit(j);
            if (v > bv) { bv = v; bi = c0 + j; }
        }
        tk_wave_argmax(bv, bi);
        if (lane == 0) tk_cand(a, pos & 1)[c] = make_float2(bv, __int_as_float(bi));
    }
    // q4_0: any of the CU's eight waves may have raised the word (an image slice beyond the f16 range: tk_coop_part) -- the streaming
    // waves' gathers have no other way to tell this wave, and until round 6 an event in ANOTHER wave's slice never reached the host's
    // copy of the word (direct mode): NaN logits returned as an answer.  All of this CU's gathers lie behind the barrier above.
    if constexpr (SH::Q4) { if (ok && lane == 0 && !nosync) ok = __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0; }
    if (!ok && lane == 0) {
        const unsigned cause = atomicOr(a.err, 0x1000u);      // (what the first failing wave left there: a timed-out spin's code, 0x4000 ...)
        if (a.herr) __hip_atomic_store(a.herr, cause | 0x1000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ------------------------------------------------------------------------------------------------
// STREAMING wave: a static list of 8 KB row tiles, always NB requested ahead (register ring),
// consumed between the phase's two barriers.  Nothing here depends on another CU.
// ------------------------------------------------------------------------------------------------

// slots K .. K+N-1 (compile-time) of layer l: consume ring entry K % NB, refill it with slot K + NB
template <class SH, int K, int N, bool CLS>
__device__ __forceinline__ void tk_run(TkRing<SH>& r, const TokenArgs& a, int l, int c, int sw, const TkX<SH>& x, float* part, int lane) {
    if constexpr (N > 0) {
        constexpr int R = K % SH::NB;
        tk_consume<SH>(r.b[R], r.t[R], x, part, lane);
        if constexpr (CLS) r.t[R] = tk_cls_at<SH, K + SH::NB>(a, c, sw);
        else r.t[R] = tk_at<SH, K + SH::NB>(a, l, c, sw);
        tk_issue<SH>(r.b[R], r.t[R], a.zeros, lane);
        tk_run<SH, K + 1, N - 1, CLS>(r, a, l, c, sw, x, part, lane);
    }
}
// consume only / refill only: a phase's LAST min(NB, slots) slots are dotted first, the partial sums
// handed to the service wave (barrier B), and only THEN refilled.  Issuing a refill can block for
// microseconds when the CU's memory pipeline is full of earlier prefetches; that wait must not sit
// between the dot products and the publish of the phase's result.
// N tiles at once: all per-lane dots first, then the N wave reductions (independent DPP chains the
// scheduler can interleave), then ONE lane-0 block of LDS writes
template <class SH, int K, int N>
__device__ __forceinline__ void tk_eat(const TkRing<SH>& r, const TkX<SH>& x, float* part, int lane) {
    if constexpr (N > 0) {
        float v[N][SH::RPT];
#pragma unroll
        for (int i = 0; i < N; ++i) tk_dot<SH>(r.b[(K + i) % SH::NB], x, v[i]);
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int s = 0; s < SH::RPT; ++s) v[i][s] = wave_sum(v[i][s]);
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < N; ++i)
#pragma unroll
                for (int s = 0; s < SH::RPT; ++s) part[r.t[(K + i) % SH::NB].pidx + s * r.t[(K + i) % SH::NB].pstep] = v[i][s];
        }
    }
}
template <class SH, int K, int N, bool CLS>
__device__ __forceinline__ void tk_refill(TkRing<SH>& r, const TokenArgs& a, int l, int c, int sw, int lane) {
    if constexpr (N > 0) {
        constexpr int R = K % SH::NB;
        if constexpr (CLS) r.t[R] = tk_cls_at<SH, K + SH::NB>(a, c, sw);
        else r.t[R] = tk_at<SH, K + SH::NB>(a, l, c, sw);
        tk_issue<SH>(r.b[R], r.t[R], a.zeros, lane);
        tk_refill<SH, K + 1, N - 1, CLS>(r, a, l, c, sw, lane);
    }
}
// one phase of S slots starting at slot K0: [barrier A] early slots (consume+refill), late slots
// (consume), [barrier B], late refills
template <class SH, int K0, int S, bool CLS, bool WIDE = false>
__device__ __forceinline__ void tk_phase(TkRing<SH>& r, const TokenArgs& a, int l, int c, int sw, const float4* xs4,
                                         float* part, int lane, volatile int* gflag = nullptr, int seq = 0) {
    constexpr int LATE0 = S < SH::NB ? S : SH::NB;
    constexpr int WANT = (!CLS && K0 == TkSched<SH>::KA) ? SH::EARLY_A : 0;   // leading slots refilled in-phase
    constexpr int EARLY = (S - LATE0) > (WANT < S ? WANT : S - 1) ? (S - LATE0) : (WANT < S ? WANT : S - 1), LATE = S - EARLY;
    tk_barrier();
    TkX<SH> x;
    if constexpr (SH::Q4) {
        if constexpr (WIDE) x.template load_q4<SH::NBP_H>(xs4, (sw % SH::TPR_H) * SH::LPT, SH::NBLK_H, lane);
        else x.template load_q4<SH::NBP_E>(xs4, 0, SH::NBLK_E, lane);
    } else {
        if constexpr (WIDE) x.load(xs4, (sw % SH::TPR_H) * SH::LPT, SH::LPR_H, lane);
        else x.load(xs4, 0, SH::LPR_E, lane);
    }
    tk_run<SH, K0, EARLY, CLS>(r, a, l, c, sw, x, part, lane);
    tk_eat<SH, K0 + EARLY, LATE>(r, x, part, lane);
    tk_barrier();
    if constexpr (LLMK_TK_GF && !SH::COOP && !CLS) {
        // LLMK_TK_GF: a first part of the burst now, the rest once the service wave's next sweep is in the pipeline ahead of it
        constexpr int NOW = SH::GF_NOW < LATE ? SH::GF_NOW : LATE;
        if (SH::GF_PUB && gflag) tk_flag_wait(gflag + 1, seq);
        tk_refill<SH, K0 + EARLY, NOW, CLS>(r, a, l, c, sw, lane);
        if (gflag) tk_flag_wait(gflag, seq);
        tk_refill<SH, K0 + EARLY + NOW, LATE - NOW, CLS>(r, a, l, c, sw, lane);
    } else {
        tk_refill<SH, K0 + EARLY, LATE, CLS>(r, a, l, c, sw, lane);
    }
}

template <class SH, int K>
__device__ __forceinline__ void tk_prime(TkRing<SH>& r, const TokenArgs& a, int c, int sw, int lane) {
    if constexpr (K < SH::NB) {
        r.t[K] = tk_at<SH, K>(a, 0, c, sw);
        tk_issue<SH>(r.b[K], r.t[K], a.zeros, lane);
        tk_prime<SH, K + 1>(r, a, c, sw, lane);
    }
}

template <class SH>
__device__ __forceinline__ void tk_stream(const TokenArgs& a, char* lds, int c, int sw, int lane, int tid) {
    typedef TkLds<SH> LD;
    typedef TkSched<SH> SC;
    const float4* xs4 = reinterpret_cast<const float4*>(lds + LD::XS);
    float* part = reinterpret_cast<float*>(lds + LD::PART);
    const int L = a.L;
    const int pos = a.tokpos ? a.tokpos[1] : a.pos_imm;
    constexpr int HPC = TK_NCU / SH::NH;
    // head h runs on CU h*HPC + (its kv group mod HPC): with the dispatcher placing block b on XCD b % 8 the heads
    // of one kv group share an XCD (one L2 copy of their K/V rows) and different groups use different XCDs
    const int my_head = c / HPC;                    // (its attention CU, or a part of it at long contexts: tk_att_role)

    volatile int* gflag = reinterpret_cast<volatile int*>(lds + LD::RED8 + 32);   // LLMK_TK_GF
    TkRing<SH> r;
    tk_prime<SH, 0>(r, a, c, sw, lane);

    for (int l = 0; l < L; ++l) {
        {   // QKV phase; on the attention CUs the refills wait until attention has issued ITS loads,
            // which would otherwise queue behind 100+ KB of prefetch in this CU's memory pipeline
            constexpr int LATE = SH::SL_Q < SH::NB ? SH::SL_Q : SH::NB, EARLY = SH::SL_Q - LATE;
            tk_barrier();
            TkX<SH> x;
            if constexpr (SH::Q4) x.template load_q4<SH::NBP_E>(xs4, 0, SH::NBLK_E, lane);
            else x.load(xs4, 0, SH::LPR_E, lane);
            tk_run<SH, SC::KQ, EARLY, false>(r, a, l, c, sw, x, part, lane);
            tk_eat<SH, SC::KQ + EARLY, LATE>(r, x, part, lane);
            tk_barrier();
            int at0, at1;
            if (tk_att_role<SH>(c, pos, at0, at1)) {
                TkAtt<SH> pa;
                pa.prefetch(a, l, my_head, pos, tid, at0);   // before the wait for q: K/V latency overlaps it
                tk_barrier();
                tk_attention<SH>(a, lds, l, my_head, pos, tid, pa, nullptr, at0, at1);
                tk_barrier();
            }
            if constexpr (LLMK_TK_GF && !SH::COOP) {
                constexpr int NOW = SH::GF_NOW < LATE ? SH::GF_NOW : LATE;
                if (SH::GF_PUB) tk_flag_wait(gflag + 1, 4 * l + 1);
                tk_refill<SH, SC::KQ + EARLY, NOW, false>(r, a, l, c, sw, lane);
                tk_flag_wait(gflag, 4 * l + 1);
                tk_refill<SH, SC::KQ + EARLY + NOW, LATE - NOW, false>(r, a, l, c, sw, lane);
            } else {
                tk_refill<SH, SC::KQ + EARLY, LATE, false>(r, a, l, c, sw, lane);
            }
        }
        tk_phase<SH, SC::KO, SH::SL_O, false>(r, a, l, c, sw, xs4, part, lane, gflag, 4 * l + 2);
        tk_phase<SH, SC::KA, SH::SL_A, false>(r, a, l, c, sw, xs4, part, lane, gflag, 4 * l + 3);
        tk_phase<SH, SC::KD, SC::SLP - SC::KD, false, true>(r, a, l, c, sw, xs4, part, lane, gflag, 4 * l + 4);   // w2 slots + padding
    }
    // classifier: the ring index is 0 again (SLP is a multiple of NB); refills run off the stream's end
    tk_phase<SH, 0, SH::SL_C, true>(r, a, L, c, sw, xs4, part, lane);
}

// STREAMING wave of a COOP shape (q4_0): the same static tile list with lagged refills (tk_step), and this wave (sw = 0..6;
// the service wave is slice 7) gathers its eighth of each phase's input vector right after the phase before it.  Vector and
// epoch of every gather mirror tk_service.
template <class SH>
__device__ __forceinline__ void tk_stream_coop(const TokenArgs& a, char* lds, int c, int sw, int lane_in, int tid) {
    const int lane = lane_in;
    typedef TkLds<SH> LD;
    typedef TkSched<SH> SC;
    float* xs = reinterpret_cast<float*>(lds + LD::XS);
    float* xraw = reinterpret_cast<float*>(lds + LD::XRAW);
    const char* xs4 = lds + LD::XS;                 // the x image the units are dotted with (q4_units.h)
    float* part = reinterpret_cast<float*>(lds + LD::PART);
    float* red8 = reinterpret_cast<float*>(lds + LD::RED8);
    constexpr int TR_E = SH::TR_E, TR_H = SH::TR_H;
    const int L = a.L;
    const int pos = a.tokpos ? a.tokpos[1] : a.pos_imm;
    const unsigned ebase = (unsigned)(a.tokpos ? a.tokpos[2] : a.serial_imm) * (unsigned)(5 * L + 2);
    const bool nosync = TK_DEBUG && (a.gflags & TKG_NOSYNC) != 0;
    constexpr int HPC = TK_NCU / SH::NH;
    const bool att_cu = (c % HPC) == ((c / HPC / SH::KVMUL) % HPC);
    const int my_head = c / HPC;

    TkRing<SH> r;
    tk_prime_coop<SH, 0>(r, a, c, sw, lane);
    constexpr bool ADVH = LLMK_TK_ADV_HOP != 0;

    for (int l = 0; l < L; ++l) {
        const unsigned e_q = ebase + 5u * l + 1, e_att = e_q + 1, e_o = e_q + 2, e_a = e_q + 3, e_d = e_q + 4;
        // QKV phase (its input was gathered at the end of the previous layer; layer 0: the service wave stages the embedding row)
        tk_phase_body<SH, SC::KQ, SH::SL_Q, false>(r, a, l, c, sw, xs4, part, lane);
        int at0, at1;
        if (tk_att_role<SH>(c, pos, at0, at1)) {
            TkAtt<SH> pa;
            pa.prefetch(a, l, my_head, pos, tid, at0);
            tk_barrier();
            tk_attention<SH>(a, lds, l, my_head, pos, tid, pa, nullptr, at0, at1);
            tk_barrier();
            // the service wave now folds and PUBLISHES this head's output -- the one store the other 224 CUs wait for -- and
            // stores share the CU's memory pipeline with loads: issued at once, the request below put 57 KB in front of it
            // (first version: xb arrived 0.7 us later everywhere and the whole gain was gone)
            if constexpr (ADVH) __builtin_amdgcn_s_sleep(48);
        }
        // the wo slot's request, in the window of the attention hop (behind this CU's own attention, if it has any)
        if constexpr (ADVH) tk_request<SH, SC::KO, false>(r, a, l, c, sw, lane);
        if (!att_cu) tk_coop_gather<SH::E, TR_E, false, SH::Q4 ? 1 : 0>(tk_g_xb<SH>(a), e_att, nullptr, xs, nullptr, red8, a.err, sw, lane, nosync);
        tk_phase_body<SH, SC::KO, SH::SL_O, false, !ADVH>(r, a, l, c, sw, xs4, part, lane);
        if constexpr (LLMK_TK_COOP_DELAY > 0) __builtin_amdgcn_s_sleep(LLMK_TK_COOP_DELAY);
        tk_coop_gather<SH::E, TR_E, true>(tk_g_xa<SH>(a), e_o, xraw, xs, tk_rms_ffn(a, l, SH::E), red8, a.err, sw, lane, nosync);
        tk_phase_body<SH, SC::KA, SH::SL_A, false>(r, a, l, c, sw, xs4, part, lane);
        if constexpr (LLMK_TK_COOP_DELAY > 0) __builtin_amdgcn_s_sleep(LLMK_TK_COOP_DELAY);
        tk_coop_gather<SH::H, TR_H, false, SH::Q4 ? 2 : 0>(tk_g_hb<SH>(a), e_a, nullptr, xs, nullptr, red8, a.err, sw, lane, nosync);
        tk_phase_body<SH, SC::KD, SC::SLP - SC::KD, false>(r, a, l, c, sw, xs4, part, lane);
        if constexpr (LLMK_TK_COOP_DELAY > 0) __builtin_amdgcn_s_sleep(LLMK_TK_COOP_DELAY);
        if (l + 1 < L) {
            if (!att_cu) tk_coop_gather<SH::E, TR_E, true>(tk_g_x<SH>(a), e_d, xraw, xs, tk_rms_att(a, l + 1, SH::E), red8, a.err, sw, lane, nosync);
        } else if constexpr (!SH::CLSQ6) {
            tk_coop_gather<SH::E, TR_E, true>(tk_g_x<SH>(a), e_d, xraw, xs, tk_rms_final(a, SH::E), red8, a.err, sw, lane, nosync);
        }
    }
    if constexpr (SH::CLSQ6) {      // q6_K classifier rows (tk_service mirrors this)
        TkQ6<SH> q6;
        q6.prime(a, sw, lane);
        tk_coop_gather<SH::E, 0, true>(tk_g_x<SH>(a), ebase + 5u * L, xraw, xs, tk_rms_final(a, SH::E), red8, a.err, sw, lane, nosync);
        tk_barrier();
        q6.run(a, sw, lane, xs, part);
        tk_barrier();
    } else
    tk_phase_body<SH, 0, SH::SL_C, true>(r, a, L, c, sw, xs4, part, lane);
}

template <class SH, bool GR = false>
__global__ __launch_bounds__(TK_THREADS, 2) void token_kernel(TokenArgs a_in) {
    std::conditional_t<SH::Q4, TokenArgsQ4, TokenArgs> a;
    static_cast<TokenArgs&>(a) = a_in;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = blockIdx.x;
    // every wave reads its OWN zero block: one shared block would make 1800 waves hammer one HBM channel
    a.zeros += (size_t)(c * TK_WAVES + wid) * WAVE;
    {
        constexpr int HPC = TK_NCU / SH::NH;
        const int blk = c / HPC, ap = (blk / SH::KVMUL) % HPC;       // attention CU of this block of HPC CUs (see tk_service)
        const bool att_cu = (c % HPC) == ap;
        const int n = c - blk - ((c % HPC) > ap ? 1 : 0);            // rank among the CUs that own QKV / wo rows
        a.qn = att_cu ? 0 : SH::QG * (SH::QB + (n < SH::QX ? 1 : 0));
        a.q0 = SH::QG * (n * SH::QB + min(n, SH::QX));
        a.on = att_cu ? 0 : SH::RPT * (SH::OB + (n < SH::OX ? 1 : 0));
        a.o0 = SH::RPT * (n * SH::OB + min(n, SH::OX));
        a.cn = SH::RPT * (SH::CB + (c < SH::CX ? 1 : 0));
        a.c0 = SH::RPT * (c * SH::CB + min(c, SH::CX));
        if constexpr (SH::Q4) {
            // hidden units in whole 16-row groups: the row-owning CUs first (AB or AB + 1 groups each), then AG_ATT per attention CU
            a.hn = Q16_ROWS * (att_cu ? SH::AG_ATT : SH::AB + (n < SH::AX ? 1 : 0));
            a.h0 = Q16_ROWS * (att_cu ? SH::AG_W + blk * SH::AG_ATT : n * SH::AB + min(n, SH::AX));
            a.nw = att_cu ? TK_NS : TK_WAVES;
            a.dn = Q16_ROWS * (SH::DB + (c < SH::DX ? 1 : 0));
            a.d0 = Q16_ROWS * (c * SH::DB + min(c, SH::DX));
            // the image's blocks past the end of a K = H row (whole units are read), their sums, and the line of zeros: written once
            char* img = lds + TkLds<SH>::XS;
            for (int i = SH::H / 32 * Q16_IMG_BLK + tid * 16; i < SH::NBI * Q16_IMG_BLK; i += TK_THREADS * 16)
                *reinterpret_cast<uint4*>(img + i) = make_uint4(0u, 0u, 0u, 0u);
            for (int i = Q16Img<SH::NBI>::SUM + tid * 4; i < Q16Img<SH::NBI>::BYTES; i += TK_THREADS * 4) *reinterpret_cast<unsigned*>(img + i) = 0u;
            // the previous position's per-layer records (tk_qsc) into LDS, and layer 0's xb / hb scales: from its record, else the largest
            // element assumed in [1, 2) (first read behind the QKV phase's barriers)
            {
                const int pos0 = a.tokpos ? a.tokpos[1] : a.pos_imm;
                const float4* rec = tk_qsc<SH>(a, pos0 - 1);
                float4* ql = reinterpret_cast<float4*>(lds + TkLds<SH>::QSC);
                const int nrec = min(a.L, TK_QSC_LMAX);
                for (int i = tid; i < nrec; i += TK_THREADS) ql[i] = rec[i];
                if (tid == TK_NS * WAVE) reinterpret_cast<unsigned*>(lds + TkLds<SH>::RED8)[TK_EVT] = 0u;      // (the service wave: it may set the word in layer 0, before any barrier)
                if (tid < 2) {
                    const float4 r0 = rec[0];
                    const bool hit = __float_as_int(tid ? r0.w : r0.z) == pos0 - 1;
                    reinterpret_cast<float*>(lds + TkLds<SH>::RED8)[TK_XSC_B + tid] = hit ? tk_scale_for(tid ? r0.y : r0.x, TK_IMG_TARGET) : TK_IMG_TARGET;
                }
            }
        } else if constexpr (SH::H % SH::SEGW != 0) {
            // f32 / f16, ragged K = H rows: the staged hb vector is whole segments wide, the part behind H stays zero for the whole
            // launch (the gathers write H floats); first read behind the QKV phase's barriers
            float* xs = reinterpret_cast<float*>(lds + TkLds<SH>::XS);
            for (int i = SH::H + tid; i < SH::LPR_H * SH::SEGW; i += TK_THREADS) xs[i] = 0.f;
        }
    }
    if (wid == TK_NS) { __builtin_amdgcn_s_setprio(3); tk_service<SH, GR>(a, lds, c, lane, tid); }
    else if constexpr (SH::COOP) tk_stream_coop<SH>(a, lds, c, wid, lane, tid);
    else tk_stream<SH>(a, lds, c, wid, lane, tid);
}

typedef TkShape<2048, 5632, 32, 4, 32000> TkTinyLlama;   // /root/reference/llama2.f90:102-108
typedef TkShape<256, 768, 4, 2, 1024> TkSmall;           // tests/golden/tk-small*.npz: pinned to the real reference
typedef TkShape<2048, 5632, 32, 4, 32000, WT_F16> TkTinyLlamaF16;   // BASELINE.json configs[2]: the same model, f16 matrices
typedef TkShape<512, 1536, 8, 2, 1024, WT_F16> TkSmallF16;          // parity shape for the f16 tiles (tests: tk-small16)
typedef TkShape<4096, 11008, 32, 32, 32000, WT_Q4_0> TkLlama7BQ4;   // BASELINE.json configs[3]: Llama-2-7B, q4_0 matrices, head size 128
typedef TkShape<2048, 5632, 32, 4, 32000, WT_Q4_0> TkTinyLlamaQ4;   // TinyLlama-1.1B with q4_0 matrices (GQA, head size 64): the units kernel's second shape
// the same two with the classifier in q6_K rows: what `llama-quantize ... Q4_0` writes (output.weight stays q6_K) -- round 6
typedef TkShape<4096, 11008, 32, 32, 32000, WT_Q4_0, WT_Q6_K> TkLlama7BQ4Q6;
typedef TkShape<2048, 5632, 32, 4, 32000, WT_Q4_0, WT_Q6_K> TkTinyLlamaQ4Q6;
// Llama-2-7B with f16 matrices (round 6: "any shape the reference could be recompiled for" -- K = E rows of 8 segments: one row
// per tile, a 64-register x fragment; K = H rows end inside their 22nd segment)
typedef TkShape<4096, 11008, 32, 32, 32000, WT_F16> TkLlama7BF16;
// Mistral-7B's geometry (grouped-query attention at head size 128, K = H rows of 14,336) as a stock llama.cpp Q4_0 file holds it: the most
// common 7B file after Llama-2's.  Everything else of that kind: make TK_SHAPES=... (llmk.hip LLMK_TK_SHAPES)
typedef TkShape<4096, 14336, 32, 8, 32000, WT_Q4_0, WT_Q6_K> TkMistral7BQ4Q6;

}  // namespace llmk
