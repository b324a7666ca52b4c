// llmk C-ABI shim (include/llmk.h): device memory, weight upload, the per-token launch sequence
// (captured once into a hipGraph and replayed), measurement hooks.  gfx950 only; no CPU path.
//
// Token pass (replaces `transformer`, /root/reference/llama2.f90:480-640), 5 launches per layer:
//   embed                                   x = table(:,token)                      :520
//   per layer l:
//     gemv<NORM,ROPE_KV>   rmsnorm -> fused QKV GEMV -> RoPE -> q, key/value cache   :527-565
//     attn                 scores, softmax, PV per head (GQA)                        :572-598
//     gemv<RESID>          x += wo . xb                                              :603-605
//     gemv<NORM,SWIGLU>    rmsnorm -> fused w1|w3 GEMV -> silu(g)*u                  :608-616
//     gemv<RESID>          x += w2 . hb                                              :618-620
//   gemv<NORM,STORE>       final rmsnorm -> classifier                               :627-636
//   (greedy only) argmax                                                             :388
#include "../../include/llmk.h"
#include "kernels.h"
#include "token_kernel.h"
#include "prefill.h"
#include "tp_p2p.h"

#include <rccl/rccl.h>

#include <algorithm>
#include <vector>

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <new>

using namespace llmk;

extern "C" int llmk_create_tp(const llmk_config* cfg, int tp_rank, int tp_size, llmk_ctx** out);
static int q16_build(llmk_ctx* c);        // (defined behind the upload helpers it uses)

#define HIPCHK(expr)                                          \
    do {                                                      \
        hipError_t e_ = (expr);                               \
        if (e_ != hipSuccess) return LLMK_E_HIP + (int)e_;    \
    } while (0)

namespace {

// Every device allocation of the shim goes through here.  LLMK_POISON=1 (a test aid: tests/test_tp70_gpu.py and
// tests/host_tools/gpu_job.sh poison) fills the fresh allocation with 0xFF bytes -- NaN as f32 or f16, -1 as an integer --
// BEFORE the shim's own initialisation: a kernel that reads memory the shim never wrote then yields NaN every time,
// whatever the allocator happened to hand out (a fresh process usually sees zero pages, a long-lived one does not).
template <class T>
hipError_t dev_alloc(T** p, size_t bytes) {
    static const bool poison = getenv("LLMK_POISON") && getenv("LLMK_POISON")[0] == '1';
    hipError_t e = hipMalloc((void**)p, bytes);
    // (the fill must be COMPLETE when this returns: the shim's own initialisation often goes to the ctx's non-blocking stream, which the
    // null stream does not order -- round 6's first poison run: the self-test's mismatch counter read 0xFFFFFFFF on three ranks.  And it
    // must not wait for the DEVICE: with two ranks in one process a rank's kernel may be spinning for its peer, whose host thread is
    // in here -- the second poison run.  So: a stream of its own for the calling thread, and a wait for that stream alone)
    if (e == hipSuccess && poison) {
        static thread_local hipStream_t ps = nullptr;
        if (!ps) e = hipStreamCreateWithFlags(&ps, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipMemsetAsync(*p, 0xFF, bytes, ps);
        if (e == hipSuccess) e = hipStreamSynchronize(ps);
    }
    return e;
}

struct TensorDesc {
    bool layered;   // has a leading layer dimension
    int rows;       // rows per layer (1 for vectors)
    int K;          // row length
    bool matrix;    // streamed matmul weight (may be f16/q4_0); else always f32
};

struct DevTensor {
    void* data = nullptr;    // f32/f16 rows; q4_0: rows of K/2 nibble bytes followed by the row's K/32 f16 scales
    size_t row_bytes = 0;    // bytes per row in `data` (q4_0: 9K/16 rounded up to 16)
    int type = LLMK_TYPE_F32;
    bool uploaded = false;
    size_t rows_uploaded = 0;
    bool alias = false;      // `data` points into another tensor's allocation (the rmsnorm gains share one): never freed itself
};

}  // namespace

struct llmk_ctx {
    llmk_config cfg;
    int E, H, L, nh, nkv, V, S, hs, KV, kv_mul;
    // tensor-parallel shard of this ctx (tp_size == 1: the whole model). Local dims: q columns, kv dim,
    // hidden rows, vocab rows and heads owned by this rank (Megatron split).
    int tp_rank = 0, tp_size = 1;
    int Eq, KVl, Hl, Vl, nhl;
    ncclComm_t comm = nullptr;
    // one-shot peer-memory collectives (tp_p2p.h): this rank's inbox (fine-grained HBM) and the peers' as mapped here
    unsigned long long* d_inbox = nullptr;
    unsigned* d_tp_bad = nullptr;         // the self-test's mismatch counter (allocated with the inbox: see tp_selftest_run)
    TpPeers peers = {};
    void* ipc_mapped[TP_MAX_RANKS] = {};
    bool p2p = false;
    float* d_part = nullptr;       // [E] partial sums of the row-parallel GEMVs (wo, w2) before the all-reduce
    TensorDesc gdesc[LLMK_N_TENSORS];  // the full (unsharded) tensors, what llmk_upload is handed
    TensorDesc desc[LLMK_N_TENSORS];
    DevTensor t[LLMK_N_TENSORS];
    float *d_kc = nullptr, *d_vc = nullptr;  // [L][S][KV]  (RunState, weight_module.f90:33-40)
    float *d_x = nullptr, *d_q = nullptr, *d_xb = nullptr, *d_hb = nullptr, *d_logits = nullptr, *d_rope = nullptr;
    int *d_tokpos = nullptr, *d_next = nullptr;
    int* h_tokpos = nullptr;  // pinned {token0, pos1}
    float* h_logits = nullptr;  // pinned [V] (+ error word)
    float* h_logits_dev = nullptr;   // the same buffer as the device sees it (token kernel direct mode)
    bool tk_direct = true;
    int* h_next = nullptr;      // pinned
    hipStream_t stream = nullptr;
    hipGraphExec_t graph_logits = nullptr, graph_greedy = nullptr;
    hipEvent_t ev[8] = {};
    float times[5] = {0, 0, 0, 0, 0};
    int n_cu = 256;
    float eps = 1e-5f;     // rmsnorm epsilon: the reference's constant (llama2.f90:454) unless llmk_set_rms_eps
    // persistent whole-token kernel (token_kernel.h)
    bool use_tk = false;
    bool tk_short_grid = false;   // libllmk_debug.so only (LLMK_TK_INJECT_TIMEOUT)
    bool tk_retired = false;   // the token kernel timed out once on this ctx: it stays on the multi-kernel path
    // q4_0 persistent kernels: positions whose activations did not fit the f16 image of x (sticky word 0x4000) are redone ONE AT A TIME
    // on the multi-kernel path and the kernel stays in use (advisor, round 5: one such position used to cost the context a third of
    // its rate for good); TK_RANGE_LIMIT of them in a row retire it after all (a model it cannot hold), and its unit copies are freed
    int tk_range_events = 0, tk_range_run = 0;
    int tk_shape = 0;      // 1-based index into tk_table (the instantiated shapes: LLMK_TK_SHAPES), 0 = none
    unsigned long long* d_gran = nullptr;  // exchange granules: qkv | xb | xa | hb | x | attention parts
    float4* d_zeros = nullptr;
    unsigned long long* d_trace = nullptr;  // debug stamps (LLMK_TK_TRACE=1)
    size_t tk_lds = 0;
    size_t tk_ngran = 0;   // granules in d_gran (the last 4 * TK_QSC_LMAX: the q4_0 kernels' per-layer scale records)
    // pipelined greedy decode (llmk_decode_greedy): per-CU classifier maxima of the last two launches, and the ids as they
    // are resolved (host-mapped; 0 = not there yet)
    // (the candidates sit behind the device error word, the ids behind the host error word: token_kernel.h tk_cand)
    // batched prefill (prefill.h), allocated by the first llmk_prefill.  Two LANES = workspace set + stream: consecutive
    // 128-position batches of a prompt alternate between them, so one batch's small kernels (epilogues, attention) and
    // launch gaps run under the other's GEMMs.  The only cross-batch dependency is the KV cache: batch k+1's attention in
    // layer l waits for batch k's K/V rows of layer l (event kv[l]).
    struct PfLane {
        float *X = nullptr, *Xs = nullptr, *Q = nullptr, *XB = nullptr, *HB = nullptr, *P = nullptr, *xn = nullptr;
        unsigned* lowcnt = nullptr;        // [2][PF_TMAX]: the GEMM workgroups' votes on positions that are small as a whole (prefill.h pf_low_check)
        hipStream_t stream = nullptr;      // lane 0: the ctx stream
        std::vector<hipEvent_t> kv;        // per layer: this lane's batch has written its K/V rows
        hipEvent_t done = nullptr;         // this lane's batch has left the last layer
    } pf[2];
    int* pf_tok = nullptr;
    hipEvent_t pf_start = nullptr;         // tokens are on the device (and everything before the prefill call is done)
    bool pf_ready = false;                 // pf_setup ran to its end
    // uploads are staged through the shim's OWN pinned memory (upload_block): two buffers, an event each
    void* up_stage[2] = {nullptr, nullptr};
    void* up_stage_dev[2] = {nullptr, nullptr};      // the same buffers as the device addresses them
    void* up_tmp[2] = {nullptr, nullptr};            // device scratch of the q4_0 re-packing (written and read by kernels only)
    hipEvent_t up_done[2] = {nullptr, nullptr};
    unsigned long long* up_acc = nullptr;            // device word of the upload verification's 16-bit word sums
    bool up_ready = false;                           // all of the above exist (set only after every allocation succeeded)
    // q4_0 matrices in the persistent kernel's UNIT layout (q4_units.h: 16 rows x 32 blocks as matrix-core operands), built from
    // the row layout -- which the multi-kernel path, the prefill and the fallback keep reading -- at the first token after an upload
    void* q16[LLMK_N_TENSORS] = {};
    bool q16_dirty = true;
    bool pf_hm = false;                    // GEMMs on v_mfma_f32_16x16x32_f16, activations (and f32 / q4_0 weights) as two f16 pieces (prefill.h)
    unsigned* pf_flag = nullptr;           // device word: an activation did not fit f16 (the call is redone on the f32 instruction)
};
typedef llmk_ctx::PfLane PfLane;

namespace {

constexpr size_t TENSOR_SLACK = 64 * 1024;
size_t q4_row_stride(int K) { return (size_t)K / 2 + (size_t)((K / 32 + 63) / 64) * 128; }

size_t row_bytes_for(int type, int K) {
    switch (type) {
        case LLMK_TYPE_F32: return (size_t)K * 4;
        case LLMK_TYPE_F16: return (size_t)K * 2;
        case LLMK_TYPE_Q4_0: return (size_t)K / 32 * 18;
        case LLMK_TYPE_Q6_K: return (size_t)K / Q6K_WEIGHTS * Q6K_BLOCK_BYTES;      // (the classifier only: q6k.h)
    }
    return 0;
}
// bytes between a tensor's rows ON THE DEVICE (q4_0 and q6_K rows are re-packed by the upload: kernels.h, q6k.h)
size_t dev_row_bytes(int type, int K) {
    return type == LLMK_TYPE_Q4_0 ? q4_row_stride(K) : type == LLMK_TYPE_Q6_K ? q6k_row_stride(K) : row_bytes_for(type, K);
}

// llmk_create walks the token pass once with g_prepare set: nothing is launched, but every kernel whose dynamic LDS
// request (the staged activation vector, the attention score row) exceeds the 64 KB default limit gets its limit raised,
// so a long context or a wide contraction fails at create (LLMK_E_SHAPE beyond 160 KB) and never at the first forward.
thread_local bool g_prepare = false;
constexpr size_t LDS_DEFAULT_LIMIT = 64 * 1024, LDS_MAX = 160 * 1024;
template <class F>
bool prepare_only(F kernel, size_t smem, hipError_t* e) {
    if (!g_prepare) return false;
    *e = smem > LDS_MAX ? hipErrorInvalidValue
       : smem > LDS_DEFAULT_LIMIT ? hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)
                                  : hipSuccess;
    return true;
}

template <int WT, int EPI, bool NORM, int ROWS, int NCH>
hipError_t launch_gemv(hipStream_t st, const GemvArgs& a) {
    const int njobs = (EPI == EPI_SWIGLU) ? a.H : a.rows / ROWS;
    const int blocks = (njobs + GEMV_WAVES - 1) / GEMV_WAVES;
    const size_t smem = 16 + (size_t)a.K * sizeof(float);
    hipError_t pe;
    if (prepare_only(gemv_kernel<WT, EPI, NORM, ROWS, NCH>, smem, &pe)) return pe;
    hipLaunchKernelGGL((gemv_kernel<WT, EPI, NORM, ROWS, NCH>), dim3(blocks), dim3(GEMV_THREADS), smem, st, a);
    return hipGetLastError();
}

// Tile shapes: ROWS x NCH 16-byte vectors per lane are requested up front (<= ~88 VGPRs).
//   f32  K=2048: 2 rows x 8 = the whole pair (16 KB/wave);  K=5632 (w2): 1 row x 22 = whole row
//   f16  K=2048: 2 x 4;                                      K=5632: 1 x 11
//   q4_0 K=4096: 2 x 2
template <int EPI, bool NORM>
hipError_t launch_gemv_t(int wt, hipStream_t st, const GemvArgs& a, int n_cu) {
    constexpr bool single_ok = (EPI == EPI_STORE || EPI == EPI_RESID) && !NORM;
    switch (wt) {
        case LLMK_TYPE_F32: {
            const int ncol = a.K / 4 / WAVE;
            if constexpr (single_ok)
                if (ncol % 22 == 0) return launch_gemv<WT_F32, EPI, NORM, 1, 22>(st, a);
            return launch_gemv<WT_F32, EPI, NORM, 2, 8>(st, a);
        }
        case LLMK_TYPE_F16: {
            const int ncol = a.K / 8 / WAVE;
            if constexpr (single_ok)
                if (ncol % 11 == 0) return launch_gemv<WT_F16, EPI, NORM, 1, 11>(st, a);
            return launch_gemv<WT_F16, EPI, NORM, 2, 4>(st, a);
        }
        default: {   // q4_0: dedicated kernel, up to 8 rows per wave (kernels.h); fewer when the matrix is
                     // too small to give every CU at least two workgroups that way
            const int npairs = (EPI == EPI_SWIGLU) ? a.H : a.rows / 2;
            const size_t smem = 16 + (size_t)a.K * sizeof(float) + 8 * 16 + (size_t)(a.K / 32) * sizeof(float);   // x (pitch nblk+1) + block sums
            // workgroups wanted before a wave takes more pairs: two per CU -- one and a half where x is 32 KB or more (K >= 8192: every
            // workgroup stages the whole x; the 70B rank's w1|w3, 3,584 pairs: 896 workgroups of one pair per wave 12.2 us, 448 of
            // two 10.9 us, 224 of four 14.8 us; LLMK_Q4_NP sweep, profiles/r04_tp70_rank_kernels_pairs_per_wave.txt)
            const int want = a.K >= 8192 ? 3 * n_cu / 2 : 2 * n_cu;
#define Q4_LAUNCH(NP_, KS_)                                                                                      \
            do {                                                                                                 \
                const int blocks = (npairs + (GEMV_WAVES / KS_) * NP_ - 1) / ((GEMV_WAVES / KS_) * NP_);         \
                hipError_t pe;                                                                                   \
                if (prepare_only(gemv_q4_kernel<EPI, NORM, NP_, KS_>, smem, &pe)) return pe;                     \
                hipLaunchKernelGGL((gemv_q4_kernel<EPI, NORM, NP_, KS_>), dim3(blocks), dim3(GEMV_THREADS), smem, st, a); \
            } while (0)
            // wide rows (K a multiple of 8192: the Llama-2-70B rank shapes): column slices across the block's waves
            static const int ks_env = getenv("LLMK_Q4_KS") ? atoi(getenv("LLMK_Q4_KS")) : -1;     // measurement aid: 0 = never
            // ... and only where even one pair per wave leaves CUs without a block (measured on rank 0 of 8 of the 70B shape,
            // tests/host_tools/tp_rank_time.py: QKV, 640 pairs, 7.9 -> 6.8 us; w1|w3 13.3 -> 18.8 us and the classifier shard
            // 9.3 -> 12.0 us the OTHER way: four times the blocks each stage the whole x)
            const bool wide = (a.K % 8192) == 0 && ks_env != 0 && (npairs / GEMV_WAVES < n_cu || ks_env > 0);
            static const int np_env = getenv("LLMK_Q4_NP") ? atoi(getenv("LLMK_Q4_NP")) : 0;       // measurement aid: pairs per wave, forced
            if (np_env > 0 && !wide) {          // (in the column-sliced launches below too)
                if (np_env >= 4) Q4_LAUNCH(4, 1);
                else if (np_env >= 2) Q4_LAUNCH(2, 1);
                else Q4_LAUNCH(1, 1);
            }
            else if (wide) {
                if (np_env >= 4) Q4_LAUNCH(4, 4);
                else if (np_env >= 2) Q4_LAUNCH(2, 4);
                else if (np_env == 1) Q4_LAUNCH(1, 4);
                else if (npairs / 4 >= want) Q4_LAUNCH(4, 4);
                else if (npairs / 2 >= want) Q4_LAUNCH(2, 4);
                else Q4_LAUNCH(1, 4);
            }
            else if (npairs / (GEMV_WAVES * 4) >= want) Q4_LAUNCH(4, 1);
            else if (npairs / (GEMV_WAVES * 2) >= want) Q4_LAUNCH(2, 1);
            else Q4_LAUNCH(1, 1);
#undef Q4_LAUNCH
            return hipGetLastError();
        }
    }
}

hipError_t launch_attn(llmk_ctx* c, int l) {
    const float* kc = c->d_kc + (size_t)l * c->S * c->KVl;
    const float* vc = c->d_vc + (size_t)l * c->S * c->KVl;
    const size_t smem = (516 + (size_t)c->S) * sizeof(float);
#define ATT(HS_)                                                                                                 \
    do {                                                                                                         \
        hipError_t pe;                                                                                           \
        if (prepare_only(attn_kernel<HS_>, smem, &pe)) return pe;                                                \
        hipLaunchKernelGGL((attn_kernel<HS_>), dim3(c->nhl), dim3(256), smem, c->stream, c->d_q, kc, vc, c->d_xb, \
                           c->d_tokpos, c->KVl, c->kv_mul);                                                      \
    } while (0)
    switch (c->hs) {
        case 16: ATT(16); break;
        case 32: ATT(32); break;
        case 64: ATT(64); break;
        case 128: ATT(128); break;
        default: return hipErrorInvalidValue;
    }
#undef ATT
    return hipGetLastError();
}

GemvArgs base_args(llmk_ctx* c, int tid, int l, const float* x, const float* norm_w, float* y) {
    GemvArgs a;
    memset(&a, 0, sizeof(a));
    const DevTensor& t = c->t[tid];
    const TensorDesc& d = c->desc[tid];
    const size_t lrows = d.layered ? (size_t)l * d.rows : 0;
    a.W = (const char*)t.data + lrows * t.row_bytes;
    a.row_stride = (int)t.row_bytes;
    a.eps = c->eps;
    a.x = x;
    a.norm_w = norm_w;
    a.y = y;
    a.rows = d.rows;
    a.K = d.K;
    a.rope_freqs = c->d_rope;
    a.tokpos = c->d_tokpos;
    a.E = c->Eq;
    a.KV = c->KVl;
    a.hs = c->hs;
    a.H = c->Hl;
    return a;
}

hipError_t launch_qkv(llmk_ctx* c, int l) {
    GemvArgs a = base_args(c, LLMK_WQKV, l, c->d_x, (const float*)c->t[LLMK_RMS_ATT_WEIGHT].data + (size_t)l * c->E, c->d_q);
    a.kc = c->d_kc + (size_t)l * c->S * c->KVl;
    a.vc = c->d_vc + (size_t)l * c->S * c->KVl;
    return launch_gemv_t<EPI_ROPE_KV, true>(c->cfg.weight_type, c->stream, a, c->n_cu);
}
// Row-parallel GEMVs (wo, w2): with tp_size > 1 each rank contracts over ITS slice of the input and
// writes a PARTIAL x-increment to d_part; the all-reduce + `x += part` follow (tp_reduce_add).
hipError_t launch_wo(llmk_ctx* c, int l) {
    if (c->tp_size > 1 || c->comm || c->p2p) {
        GemvArgs a = base_args(c, LLMK_WO, l, c->d_xb, nullptr, c->d_part);
        return launch_gemv_t<EPI_STORE, false>(c->cfg.weight_type, c->stream, a, c->n_cu);
    }
    GemvArgs a = base_args(c, LLMK_WO, l, c->d_xb, nullptr, c->d_x);
    return launch_gemv_t<EPI_RESID, false>(c->cfg.weight_type, c->stream, a, c->n_cu);
}
hipError_t launch_w13(llmk_ctx* c, int l) {
    GemvArgs a = base_args(c, LLMK_W13, l, c->d_x, (const float*)c->t[LLMK_RMS_FFN_WEIGHT].data + (size_t)l * c->E, c->d_hb);
    return launch_gemv_t<EPI_SWIGLU, true>(c->cfg.weight_type, c->stream, a, c->n_cu);
}
hipError_t launch_w2(llmk_ctx* c, int l) {
    if (c->tp_size > 1 || c->comm || c->p2p) {
        GemvArgs a = base_args(c, LLMK_W2, l, c->d_hb, nullptr, c->d_part);
        return launch_gemv_t<EPI_STORE, false>(c->cfg.weight_type, c->stream, a, c->n_cu);
    }
    GemvArgs a = base_args(c, LLMK_W2, l, c->d_hb, nullptr, c->d_x);
    return launch_gemv_t<EPI_RESID, false>(c->cfg.weight_type, c->stream, a, c->n_cu);
}
hipError_t launch_cls(llmk_ctx* c) {
    // vocab-parallel: this rank's V/P rows land in its slice of the full logits vector
    GemvArgs a = base_args(c, LLMK_WCLS, 0, c->d_x, (const float*)c->t[LLMK_RMS_FINAL_WEIGHT].data,
                           c->d_logits + (size_t)c->tp_rank * c->Vl);
    if (c->t[LLMK_WCLS].type == LLMK_TYPE_Q6_K) {      // raw q6_K rows (a stock q4_0 file's output.weight): q6k.h
        const size_t smem = 16 + (size_t)a.K * sizeof(float);
        const int blocks = (a.rows + GEMV_WAVES * Q6K_RPW - 1) / (GEMV_WAVES * Q6K_RPW);
        hipError_t pe;
        if (prepare_only(gemv_q6k_kernel<true>, smem, &pe)) return pe;
        hipLaunchKernelGGL((gemv_q6k_kernel<true>), dim3(blocks), dim3(GEMV_THREADS), smem, c->stream, a);
        return hipGetLastError();
    }
    return launch_gemv_t<EPI_STORE, true>(c->t[LLMK_WCLS].type, c->stream, a, c->n_cu);   // the classifier may have its own type
}
hipError_t launch_embed(llmk_ctx* c) {
    hipLaunchKernelGGL(embed_kernel, dim3((c->E + 255) / 256), dim3(256), 0, c->stream,
                       (const float*)c->t[LLMK_TOKEN_EMBEDDING_TABLE].data, c->d_tokpos, c->d_x, c->E);
    return hipGetLastError();
}

#define HIPRET(expr)                     \
    do {                                 \
        hipError_t e_ = (expr);          \
        if (e_ != hipSuccess) return e_; \
    } while (0)

// direct = true: token/pos/serial travel as kernel arguments and the logits (+ a sticky error word) are written
// straight into the pinned host buffer, so a token is ONE launch and one stream sync (no copy nodes around it)
// what a launch of the pipelined greedy decode adds to the arguments (all null otherwise)
struct TkGreedy { int gflags = 0, id_index = 0; };
template <class TK>
hipError_t launch_token_kernel_t(llmk_ctx* c, bool direct, const TkGreedy& g) {
    TokenArgs a;
    a.gflags = g.gflags | ((TK_DEBUG && getenv("LLMK_TK_NOSYNC")) ? TKG_NOSYNC : 0);   // NOSYNC: libllmk_debug.so only
    a.emb = (const float*)c->t[LLMK_TOKEN_EMBEDDING_TABLE].data;
    a.rms = (const float*)c->t[LLMK_RMS_ATT_WEIGHT].data;      // att | ffn | final in one allocation (llmk_create_tp)
    // q4_0: the unit layout (check_ready built it)
    a.wqkv = TK::Q4 ? c->q16[LLMK_WQKV] : c->t[LLMK_WQKV].data;
    a.wo = TK::Q4 ? c->q16[LLMK_WO] : c->t[LLMK_WO].data;
    a.w13 = TK::Q4 ? c->q16[LLMK_W13] : c->t[LLMK_W13].data;
    a.w2 = TK::Q4 ? c->q16[LLMK_W2] : c->t[LLMK_W2].data;
    a.wcls = (TK::Q4 && !TK::CLSQ6) ? c->q16[LLMK_WCLS] : c->t[LLMK_WCLS].data;      // (q6_K classifier rows stay rows: q6k.h)
    a.kc = c->d_kc;
    a.vc = c->d_vc;
    a.rope = c->d_rope;
    a.tokpos = (direct || g.gflags) ? nullptr : c->d_tokpos;   // immediates: direct mode, and every launch of a greedy pipeline (its own pos and serial)
    a.tok_imm = (g.gflags & TKG_CAND_IN) ? g.id_index : c->h_tokpos[0];   // the token comes from the candidates: the word carries the id's slot
    a.pos_imm = c->h_tokpos[1];
    a.serial_imm = c->h_tokpos[2];
    a.g_qkv = c->d_gran;         // qkv | xb | xa | hb | x (token_kernel.h tk_g_xb ...)
    a.logits = direct ? c->h_logits_dev : c->d_logits;
    a.err = reinterpret_cast<unsigned*>(c->d_logits + c->V);
    a.herr = (direct || g.gflags) ? reinterpret_cast<unsigned*>(c->h_logits_dev + c->V) : nullptr;
    a.zeros = c->d_zeros;
#ifdef LLMK_TK_DEBUG
    a.trace = c->d_trace;
#endif
    a.L = c->L;
    a.S = c->S;
    a.eps = c->eps;
    const dim3 grid((TK_DEBUG && c->tk_short_grid) ? TK_NCU - 1 : TK_NCU);
    if (g.gflags) hipLaunchKernelGGL((token_kernel<TK, true>), grid, dim3(TK_THREADS), c->tk_lds, c->stream, a);
    else hipLaunchKernelGGL((token_kernel<TK>), grid, dim3(TK_THREADS), c->tk_lds, c->stream, a);
    return hipGetLastError();
}
// ---- the shapes the persistent kernel is instantiated for -------------------------------------------------------------------
// The reference's dims are seven `parameter` lines (llama2.f90:102-108: edit, make, run); the persistent kernel's are template
// arguments, so a shape is one line here: the built-in list below (BASELINE.json's configurations, the parity shapes, the
// stock-file q6_K variants, Llama-2-7B f16) plus whatever the build adds --
//     make -C llm.f90_amd TK_SHAPES="4096,14336,32,8,32000,WT_F16 5120,13824,40,40,32000,WT_F32"
// turns every "E,H,NH,NKV,V,WT[,CLS]" into X(TkShape<...>) (Makefile: LLMK_TK_EXTRA_SHAPES); TkShape's static_asserts say at
// compile time whether a shape fits the kernel (token_kernel.h).  llmk_tk_shapes() lists them; `llm --vx` prints that list.
#ifndef LLMK_TK_EXTRA_SHAPES
#define LLMK_TK_EXTRA_SHAPES(X)
#endif
#define LLMK_TK_SHAPES(X)                                                                                          \
    X(TkTinyLlama) X(TkSmall) X(TkTinyLlamaF16) X(TkSmallF16) X(TkLlama7BQ4) X(TkTinyLlamaQ4) X(TkLlama7BQ4Q6)     \
    X(TkTinyLlamaQ4Q6) X(TkLlama7BF16) X(TkMistral7BQ4Q6) LLMK_TK_EXTRA_SHAPES(X)
struct TkEntry {
    int E, H, NH, NKV, V, WT, CLS;
    bool q4;                                        // units layout (q16_build) before the first token
    hipError_t (*launch)(llmk_ctx*, bool, const TkGreedy&);
    int (*setup)(llmk_ctx*, int);
};
template <class TK> int tk_setup(llmk_ctx* c, int id);
#define TK_ENTRY(...) {__VA_ARGS__::E, __VA_ARGS__::H, __VA_ARGS__::NH, __VA_ARGS__::NKV, __VA_ARGS__::V, __VA_ARGS__::WT, __VA_ARGS__::CLS, \
                       __VA_ARGS__::Q4, launch_token_kernel_t<__VA_ARGS__>, tk_setup<__VA_ARGS__>},
static const TkEntry tk_table[] = {LLMK_TK_SHAPES(TK_ENTRY)};
#undef TK_ENTRY
constexpr int TK_NSHAPES = (int)(sizeof(tk_table) / sizeof(tk_table[0]));
hipError_t launch_token_kernel(llmk_ctx* c, bool direct = false, const TkGreedy& g = TkGreedy()) {
    if (c->tk_shape < 1 || c->tk_shape > TK_NSHAPES) return hipErrorInvalidValue;
    return tk_table[c->tk_shape - 1].launch(c, direct, g);
}
// the last position of a pipelined greedy run has no next launch to fold its candidates: this does (1 wave)
// (no id while the sticky error word is set, and none from candidates without a finite maximum: see tk_token)
__global__ __launch_bounds__(64) void cand_resolve_kernel(const float2* __restrict__ cand, int* id_out, int* next, unsigned* err, int V) {
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int k = threadIdx.x; k < TK_NCU; k += 64) {
        const float2 cd = cand[k];
        const int ci = __float_as_int(cd.y);
        if (cd.x > bv || (cd.x == bv && ci < bi)) { bv = cd.x; bi = ci; }
    }
    tk_wave_argmax(bv, bi);
    if (threadIdx.x == 0) {
        const bool none = (unsigned)bi >= (unsigned)V;
        if (none) atomicOr(err, 0x2000u);
        if (!none && *err == 0) {
            next[0] = bi + 1;
            __hip_atomic_store(id_out, bi + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// Allocate the exchange state of the persistent kernel if cfg matches the instantiated shape TK.
template <class TK>
int tk_setup(llmk_ctx* c, int id) {
    const llmk_config& g = c->cfg;
    if (c->use_tk || g.emb_dim != TK::E || g.hidden_dim != TK::H || g.n_heads != TK::NH || g.n_kv_heads != TK::NKV ||
        g.vocab_size != TK::V || g.weight_type != TK::WT || c->t[LLMK_WCLS].type != TK::CLS)
        return LLMK_OK;
    // scores + exp(scores)
    const size_t lds = ((size_t)TkLds<TK>::ATT_S + 2 * (size_t)c->S * sizeof(float) + 15) & ~(size_t)15;
    c->tk_lds = lds < 96 * 1024 ? 96 * 1024 : lds;   // > 80 KB: never two workgroups on one CU
    if (c->tk_lds > 160 * 1024) return LLMK_OK;      // context too long for the in-LDS score row: multi-kernel path
    // The kernel spins on its peers, so all TK_NCU workgroups must be co-resident: one per CU by construction (the LDS
    // request excludes a second one), which needs n_cu >= TK_NCU (checked by the caller) and a register/LDS budget the
    // hardware admits.  Assert it with the occupancy query instead of assuming it; a part that cannot host the grid takes
    // the multi-kernel path.  (hipLaunchCooperativeKernel would run the same check per launch for +15-19 us each.)
    HIPCHK(hipFuncSetAttribute((const void*)token_kernel<TK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->tk_lds));
    HIPCHK(hipFuncSetAttribute((const void*)token_kernel<TK, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->tk_lds));
    int per_cu = 0;
    HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, token_kernel<TK>, TK_THREADS, c->tk_lds));
    if (per_cu < 1 || (long long)per_cu * c->n_cu < TK_NCU) return LLMK_OK;
    // qkv | xb | xa | hb | x | per head: the (PMAX - 1) other parts of a long context's attention (HS values + maximum + sum each)
    // (+ 4 granule slots per layer behind them (two buffers by position parity): the q4_0 kernels' per-layer records of the previous position's largest xb / hb elements,
    // token_kernel.h tk_qsc; cleared by llmk_reset)
    const size_t ngran = (size_t)TK::QKV + 3 * (size_t)TK::E + TK::H + (size_t)TK::NH * (TkAttPlan<TK>::PMAX - 1) * (TK::HS + 2) + 4 * (size_t)TK_QSC_LMAX;
    c->tk_ngran = ngran;
    if (!c->d_gran) HIPCHK(dev_alloc(&c->d_gran, ngran * sizeof(unsigned long long)));      // (kept over a re-type of the classifier: same shape)
    if (!c->d_zeros) HIPCHK(dev_alloc(&c->d_zeros, (size_t)TK_NCU * TK_WAVES * 1024));
    if (TK_DEBUG && getenv("LLMK_TK_TRACE")) HIPCHK(dev_alloc(&c->d_trace, (size_t)TK_NCU * TK_TRACE_N * 8));   // libllmk_debug.so only
    HIPCHK(hipMemset(c->d_gran, 0, ngran * sizeof(unsigned long long)));
    HIPCHK(hipMemset(c->d_zeros, 0, (size_t)TK_NCU * TK_WAVES * 1024));
    c->use_tk = true;
    c->tk_shape = id;
    return LLMK_OK;
}

// The whole-token persistent kernel serves the shapes it is instantiated for, on a full 256-CU part.  Called at create and again
// when the classifier gets a type of its own (llmk_set_tensor_type: the q6_K instantiations).
int tk_setup_all(llmk_ctx* c) {
    if ((c->cfg.flags & (LLMK_FLAG_MULTI_KERNEL | LLMK_FLAG_TIMINGS)) || c->n_cu != TK_NCU || c->tp_size != 1 || c->tk_retired) return LLMK_OK;
    int rc = LLMK_OK;
    for (int i = 0; i < TK_NSHAPES && rc == LLMK_OK; ++i) rc = tk_table[i].setup(c, i + 1);      // (the first match takes the ctx)
    return rc;
}

__global__ void bump_serial_kernel(int* tokpos) { tokpos[2] += 1; }

hipError_t enqueue_tail(llmk_ctx* c, bool greedy) {
    if (greedy) {
        hipLaunchKernelGGL(argmax_kernel, dim3(1), dim3(1024), 0, c->stream, c->d_logits, c->V, c->d_next);
        HIPRET(hipGetLastError());
        HIPRET(hipMemcpyAsync(c->h_next, c->d_next, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPRET(hipMemcpyAsync(c->h_next + 1, c->d_logits + c->V, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    } else {
        HIPRET(hipMemcpyAsync(c->h_logits, c->d_logits, ((size_t)c->V + 1) * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    }
    return hipSuccess;
}

__global__ void add_kernel(float* __restrict__ x, const float* __restrict__ part, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] += part[i];
}
hipError_t launch_add_partial(llmk_ctx* c) {   // x += all-reduced partial (the residual of wo / w2)
    hipLaunchKernelGGL(add_kernel, dim3((c->E + 255) / 256), dim3(256), 0, c->stream, c->d_x, c->d_part, c->E);
    return hipGetLastError();
}

// Tensor-parallel token pass over RCCL: two all-reduces of E floats per layer (ring order is fixed by the
// communicator, so every rank sees bit-identical sums), one all-gather of the logits.
int enqueue_token_tp(llmk_ctx* c) {
    if (!c->comm) return LLMK_E_COMM;
#define NC(expr) do { if ((expr) != ncclSuccess) return LLMK_E_COMM; } while (0)
    HIPCHK(hipMemcpyAsync(c->d_tokpos, c->h_tokpos, 4 * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIPCHK(launch_embed(c));
    for (int l = 0; l < c->L; ++l) {
        HIPCHK(launch_qkv(c, l));
        HIPCHK(launch_attn(c, l));
        HIPCHK(launch_wo(c, l));
        NC(ncclAllReduce(c->d_part, c->d_part, c->E, ncclFloat, ncclSum, c->comm, c->stream));
        HIPCHK(launch_add_partial(c));
        HIPCHK(launch_w13(c, l));
        HIPCHK(launch_w2(c, l));
        NC(ncclAllReduce(c->d_part, c->d_part, c->E, ncclFloat, ncclSum, c->comm, c->stream));
        HIPCHK(launch_add_partial(c));
    }
    HIPCHK(launch_cls(c));
    NC(ncclAllGather(c->d_logits + (size_t)c->tp_rank * c->Vl, c->d_logits, c->Vl, ncclFloat, c->comm, c->stream));
#undef NC
    return LLMK_OK;
}

// Enqueue one token pass on c->stream.  timed: bracket the reference's five sections with events.
// jseed != 0: the jittered instantiation (llmk_tp_p2p_stress only)
hipError_t launch_tp_allreduce_add(llmk_ctx* c, int call, unsigned jseed = 0) {   // x += sum over ranks of d_part   (:603-605, :618-620)
    const dim3 grid((c->E + 255) / 256), block(256);
    unsigned* err = reinterpret_cast<unsigned*>(c->d_logits + c->V);
    if (jseed) hipLaunchKernelGGL(tp_allreduce_add_kernel<true>, grid, block, 0, c->stream, c->peers, c->d_part, c->d_x, c->d_tokpos, call,
                                  2 * c->L, c->tp_rank, c->tp_size, c->E, err, jseed);
    else hipLaunchKernelGGL(tp_allreduce_add_kernel<false>, grid, block, 0, c->stream, c->peers, c->d_part, c->d_x, c->d_tokpos, call,
                            2 * c->L, c->tp_rank, c->tp_size, c->E, err, 0u);
    return hipGetLastError();
}
hipError_t launch_tp_allgather(llmk_ctx* c, unsigned jseed = 0) {   // every rank's classifier rows into every rank's logits   (:634-636)
    const dim3 grid((c->V + 255) / 256), block(256);
    unsigned* err = reinterpret_cast<unsigned*>(c->d_logits + c->V);
    if (jseed) hipLaunchKernelGGL(tp_allgather_kernel<true>, grid, block, 0, c->stream, c->peers, c->d_logits, c->d_tokpos, 2 * c->L,
                                  c->tp_rank, c->tp_size, c->E, c->V, err, jseed);
    else hipLaunchKernelGGL(tp_allgather_kernel<false>, grid, block, 0, c->stream, c->peers, c->d_logits, c->d_tokpos, 2 * c->L,
                            c->tp_rank, c->tp_size, c->E, c->V, err, 0u);
    return hipGetLastError();
}

hipError_t enqueue_token(llmk_ctx* c, bool greedy, bool timed) {
    HIPRET(hipMemcpyAsync(c->d_tokpos, c->h_tokpos, 4 * sizeof(int), hipMemcpyHostToDevice, c->stream));
    if (c->p2p) {   // tensor-parallel over peer memory: the whole token is stream work (replayable from a hipGraph)
        HIPRET(launch_embed(c));
        for (int l = 0; l < c->L; ++l) {
            HIPRET(launch_qkv(c, l));
            HIPRET(launch_attn(c, l));
            HIPRET(launch_wo(c, l));
            HIPRET(launch_tp_allreduce_add(c, 2 * l));
            HIPRET(launch_w13(c, l));
            HIPRET(launch_w2(c, l));
            HIPRET(launch_tp_allreduce_add(c, 2 * l + 1));
        }
        HIPRET(launch_cls(c));
        HIPRET(launch_tp_allgather(c));
        return enqueue_tail(c, greedy);
    }
    if (c->use_tk) {
        HIPRET(launch_token_kernel(c));
        return enqueue_tail(c, greedy);
    }
    HIPRET(launch_embed(c));
    for (int l = 0; l < c->L; ++l) {
        if (timed) HIPRET(hipEventRecord(c->ev[0], c->stream));
        HIPRET(launch_qkv(c, l));
        if (timed) HIPRET(hipEventRecord(c->ev[1], c->stream));
        HIPRET(launch_attn(c, l));
        if (timed) HIPRET(hipEventRecord(c->ev[2], c->stream));
        HIPRET(launch_wo(c, l));
        HIPRET(launch_w13(c, l));
        HIPRET(launch_w2(c, l));
        if (timed) {
            HIPRET(hipEventRecord(c->ev[3], c->stream));
            HIPRET(hipEventSynchronize(c->ev[3]));
            float ms;
            // section 1 = rmsnorm+QKV (:526-538); 2 = RoPE (:541-561) is fused into 1's epilogue;
            // 3 = attention (:570-599); 4 = wo+FFN (:601-622)
            HIPRET(hipEventElapsedTime(&ms, c->ev[0], c->ev[1])); c->times[0] += ms;
            HIPRET(hipEventElapsedTime(&ms, c->ev[1], c->ev[2])); c->times[2] += ms;
            HIPRET(hipEventElapsedTime(&ms, c->ev[2], c->ev[3])); c->times[3] += ms;
        }
    }
    if (timed) HIPRET(hipEventRecord(c->ev[4], c->stream));
    HIPRET(launch_cls(c));
    if (timed) {
        HIPRET(hipEventRecord(c->ev[5], c->stream));
        HIPRET(hipEventSynchronize(c->ev[5]));
        float ms;
        HIPRET(hipEventElapsedTime(&ms, c->ev[4], c->ev[5])); c->times[4] += ms;  // 5 = final norm + classifier
    }
    return enqueue_tail(c, greedy);
}

int build_graph(llmk_ctx* c, bool greedy, hipGraphExec_t* out) {
    hipGraph_t g = nullptr;
    HIPCHK(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
    hipError_t e = enqueue_token(c, greedy, false);
    hipError_t e2 = hipStreamEndCapture(c->stream, &g);
    if (e != hipSuccess) { if (g) hipGraphDestroy(g); return LLMK_E_HIP + (int)e; }
    HIPCHK(e2);
    e = hipGraphInstantiate(out, g, nullptr, nullptr, 0);
    hipGraphDestroy(g);
    HIPCHK(e);
    return LLMK_OK;
}

int check_ready(llmk_ctx* c) {
    if (!c) return LLMK_E_ARG;
    for (int i = 0; i < LLMK_N_TENSORS; ++i)
        if (!c->t[i].uploaded) return LLMK_E_STATE;
    if (c->use_tk && tk_table[c->tk_shape - 1].q4 && c->q16_dirty) return q16_build(c);
    return LLMK_OK;
}

// The persistent kernel reported a timed-out exchange (its workgroups were not all running: the GPU was shared, or a
// wedged peer).  Its error word is sticky by design (every CU drains instead of spinning on), so: clear it, retire the
// token kernel for this context and tell the user once.  The caller re-runs the SAME position on the multi-kernel path,
// which rewrites that position's KV rows and recomputes x from the embedding: nothing of the failed launch survives.
constexpr int TK_RANGE_LIMIT = 4;
int tk_clear_err(llmk_ctx* c) {
    HIPCHK(hipMemsetAsync(c->d_logits + c->V, 0, sizeof(float), c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    reinterpret_cast<unsigned*>(c->h_logits)[c->V] = 0;
    c->h_next[1] = 0;
    return LLMK_OK;
}
// 0x4000 and nothing else: this position's activations did not fit the q4_0 kernels' f16 image; the kernel itself is sound
bool tk_range_only(unsigned code) { return (code & 0x4000u) != 0 && (code & 0x2f00u) == 0; }
int tk_retire(llmk_ctx* c, unsigned code, int pos) {
    int rc = tk_clear_err(c);
    if (rc) return rc;
    c->use_tk = false;
    c->tk_retired = true;
    if (c->graph_logits) { hipGraphExecDestroy(c->graph_logits); c->graph_logits = nullptr; }
    if (c->graph_greedy) { hipGraphExecDestroy(c->graph_greedy); c->graph_greedy = nullptr; }
    for (int i = 0; i < LLMK_N_TENSORS; ++i)       // the q4_0 kernels' second copy of the matrices (3.8 GB at 7B): nobody reads it again
        if (c->q16[i]) { hipFree(c->q16[i]); c->q16[i] = nullptr; }
    c->q16_dirty = true;
    // what the sticky word says (token_kernel.h): 0x2000 no finite candidate among the classifier maxima (pipelined greedy decode),
    // 0x4000 an activation beyond the f16 range of the q4_0 kernels' x image, anything else a bounded spin that ran out
    const char* what = (code & 0x2000u) ? "found no finite logit among its candidates"
                     : (code & 0x4000u) ? "met an activation beyond the f16 range of its matrix-core operands"
                                        : "timed out waiting for its peer workgroups";
    fprintf(stderr, "llmk: the persistent token kernel %s (code 0x%x, position %d); this context continues on the multi-kernel path\n", what, code, pos);
    return LLMK_OK;
}

// the sticky word of a timed-out peer-memory exchange (tp_p2p.h tp_err_code), in words
const char* tp_describe_err(const llmk_ctx* c, unsigned code, char* buf, size_t n) {
    if ((code >> 28) != 3u) { snprintf(buf, n, "code 0x%x", code); return buf; }
    const unsigned per = 2u * (unsigned)c->L + 1u, e24 = code & 0xffffffu;
    // the epoch's low 24 bits, completed with the host's serial (the device's can be at most a graph replay ahead)
    const unsigned cur = (unsigned)c->h_tokpos[2] * per + per;
    unsigned epoch = (cur & ~0xffffffu) | e24;
    if (epoch > cur) epoch -= 0x1000000u;
    const unsigned serial = (epoch - 1u) / per, call = (epoch - 1u) % per;
    if (call == per - 1u) snprintf(buf, n, "code 0x%x: rank %u's logits slice never arrived (all-gather of serial %u)", code, (code >> 24) & 15u, serial);
    else snprintf(buf, n, "code 0x%x: rank %u's partial never arrived (all-reduce %u of serial %u: layer %u, %s)", code, (code >> 24) & 15u,
                  call, serial, call / 2u, (call & 1u) ? "w2" : "wo");
    return buf;
}
int run_token(llmk_ctx* c, int token, int pos, bool greedy) {
    int rc = check_ready(c);
    if (rc) return rc;
    if (token < 1 || token > c->V || pos < 1 || pos > c->S) return LLMK_E_ARG;
    HIPCHK(hipSetDevice(c->cfg.device));
    c->h_tokpos[0] = token - 1;
    c->h_tokpos[1] = pos;
    const bool timed = (c->cfg.flags & LLMK_FLAG_TIMINGS) != 0;
    for (int attempt = 0; attempt < 2; ++attempt) {
        c->h_tokpos[2] += 1;  // token serial: makes every exchange epoch of this pass unique
        // debug library only: launch the token kernel one workgroup short at this position, so its peers really time out
        c->tk_short_grid = TK_DEBUG && c->use_tk && getenv("LLMK_TK_INJECT_TIMEOUT") && atoi(getenv("LLMK_TK_INJECT_TIMEOUT")) == pos;
        if ((c->tp_size > 1 || c->comm) && !c->p2p) {   // tensor-parallel over RCCL: eager launches with the collectives in between
            rc = enqueue_token_tp(c);
            if (rc) return rc;
            HIPCHK(enqueue_tail(c, greedy));
        } else if (timed || (c->cfg.flags & LLMK_FLAG_NO_GRAPH)) {
            HIPCHK(enqueue_token(c, greedy, timed));
        } else if (c->use_tk && !greedy && c->tk_direct) {
            reinterpret_cast<unsigned*>(c->h_logits)[c->V] = 0;
            HIPCHK(launch_token_kernel(c, true));
        } else {
            hipGraphExec_t* g = greedy ? &c->graph_greedy : &c->graph_logits;
            if (!*g) {
                rc = build_graph(c, greedy, g);
                if (rc) return rc;
            }
            HIPCHK(hipGraphLaunch(*g, c->stream));
        }
        HIPCHK(hipStreamSynchronize(c->stream));
        if (c->p2p) {   // a peer never delivered its granules: the sticky word says so (cleared by llmk_reset)
            const unsigned perr = greedy ? (unsigned)c->h_next[1] : reinterpret_cast<unsigned*>(c->h_logits)[c->V];
            char what[160];
            if (perr) fprintf(stderr, "llmk: rank %d of %d: a peer's granules never arrived (%s; position %d, serial %d)\n",
                              c->tp_rank, c->tp_size, tp_describe_err(c, perr, what, sizeof(what)), pos, c->h_tokpos[2]);
            return perr ? LLMK_E_TIMEOUT : LLMK_OK;
        }
        if (!c->use_tk) return LLMK_OK;
        const unsigned err = greedy ? (unsigned)c->h_next[1] : reinterpret_cast<unsigned*>(c->h_logits)[c->V];
        if (err == 0) { c->tk_range_run = 0; return LLMK_OK; }
        if (tk_range_only(err) && c->tk_range_run + 1 < TK_RANGE_LIMIT) {
            // THIS position on the multi-kernel path (f32 activations throughout; eager launches: the ctx's graphs hold the token
            // kernel), the persistent kernel again from the next one
            ++c->tk_range_run;
            if (c->tk_range_events++ == 0)
                fprintf(stderr, "llmk: position %d holds an activation beyond the f16 range of the persistent q4_0 kernel's operands (code 0x%x): that "
                                "position is redone on the multi-kernel path, the kernel stays in use\n", pos, err);
            rc = tk_clear_err(c);
            if (rc) return rc;
            c->use_tk = false;
            const hipError_t e = enqueue_token(c, greedy, false);
            c->use_tk = true;
            HIPCHK(e);
            HIPCHK(hipStreamSynchronize(c->stream));
            return LLMK_OK;
        }
        rc = tk_retire(c, err, pos);   // then once more, on the multi-kernel path
        if (rc) return rc;
    }
    return LLMK_E_TIMEOUT;
}

// ---- batched prefill (prefill.h) ------------------------------------------------------------------------
// GEMM plan: row groups per wave, units per block and workgroups per CU (prefill.h PfGemmArgs).  Every candidate
// is priced with the measured step times (tests/host_tools/pf_trace.py, round 2): a step of NR row groups at 128
// positions keeps a SIMD's matrix core busy for 1.7*NR us per resident wave, plus ~0.5 us of barrier / staging per step;
// the first weights take ~4 us to arrive and every partial tile costs a write and a read in the epilogue.
// measured step times of pf_gemm_h_kernel at 128 positions, one / two row groups per wave (us; tests/host_tools/pf_trace.py
// --type f16 | q4_0, round 3): the activation tile costs the same either way
#ifndef LLMK_PF_H_STEP1
#define LLMK_PF_H_STEP1 1.00
#define LLMK_PF_H_STEP2 1.32
#define LLMK_PF_HQ_STEP1 1.20
#define LLMK_PF_HQ_STEP2 1.92
#endif
struct PfPlan { int nr, nk, total, U, grid; };
PfPlan pf_plan(const llmk_ctx* c, int rows, int K, int T = PF_TMAX) {
    // ONE workgroup per CU (pinned by the LDS request), one or two 16-row groups per wave.  Step times measured at 128
    // positions (tests/host_tools/pf_trace.py, profiles/README.md round 2): 2.45 us with one row group, 4.25 us with two;
    // ~4 us until the first weights arrive; every partial tile is written once and read once by the epilogue.  Also
    // measured and dropped: two workgroups per CU (their waves share the SIMDs' matrix cores: same step rate per CU, but
    // the pairs drift apart and the kernel waits for the slower one) and eight waves per workgroup (two per SIMD in step).
    // Four row groups per wave (128 accumulator registers, 256-row strips): step 8.2 us for 512 MFMAs -- 87 % of the matrix rate
    // in the loop against 84 % with two, eaten by the coarser units (TinyLlama w1|w3 67.9 vs 62.2 us; Llama-2-7B q4_0 +0.4 %).
    // f16 weights on the f16 instruction (pf_gemm_h_kernel): a step is 8x less matrix work
    static const double step_f32[2] = {2.45, 4.25}, step_h[2] = {LLMK_PF_H_STEP1, LLMK_PF_H_STEP2}, step_hq[2] = {LLMK_PF_HQ_STEP1, LLMK_PF_HQ_STEP2};
    // (f32 weights: three instructions per chunk, as q4_0; q4_0 at whole batches: the two-group strip runs on eight waves, pf_gemm_launch)
    static const double step_q8[2] = {LLMK_PF_HQ_STEP1, LLMK_PF_HQ_STEP2 * 0.925};
    const double* step_us = !c->pf_hm ? step_f32 : c->cfg.weight_type == LLMK_TYPE_F16 ? step_h
                            : (c->cfg.weight_type == LLMK_TYPE_Q4_0 && (T + 15) / 16 == 8) ? step_q8 : step_hq;
    PfPlan best{};
    double best_t = 1e30;
    for (int nr = 1; nr <= 2; ++nr) {
        const int sr = 64 * nr;
        if (nr == 2 && rows % sr) continue;
#ifdef LLMK_PF_TRACE
        if (const char* f = getenv("LLMK_PF_PLAN"))          // debug build: force the row groups per wave (when the shape allows it)
            if (atoi(f) != nr && !(rows % 128)) continue;
#endif
        PfPlan p;
        p.nr = nr;
        p.nk = K / PF_KSTEP;
        p.total = (rows + sr - 1) / sr * p.nk;
        p.U = (p.total + c->n_cu - 1) / c->n_cu;
        p.grid = (p.total + p.U - 1) / p.U;
        // before the first step and after the last: the f16-instruction kernel's ring takes 2.8 / 4.6 us to fill and its last
        // step (the flush) runs 0.5 / 1.4 us over, one / two row groups (block timelines, round 3); the f32 kernel: ~4 us
        const double fixed = !c->pf_hm ? 4.0 : nr == 1 ? 3.3 : 6.0;
        const double t = fixed + p.U * step_us[nr - 1] + 0.15 * ((p.nk - 1) / p.U + 1);
        if (t < best_t) { best_t = t; best = p; }
    }
    return best;
}
int pf_max_slots(const PfPlan& p) { return (p.nk - 1) / p.U + 2; }
hipError_t pf_prepare(const llmk_ctx* c);
void pf_teardown(llmk_ctx* c) {
    for (PfLane& w : c->pf) {
        float** bufs[] = {&w.X, &w.Xs, &w.Q, &w.XB, &w.HB, &w.P, &w.xn};
        for (float** b : bufs) { if (*b) hipFree(*b); *b = nullptr; }
        if (w.lowcnt) { hipFree(w.lowcnt); w.lowcnt = nullptr; }
        for (hipEvent_t e : w.kv) if (e) hipEventDestroy(e);
        w.kv.clear();
        if (w.done) { hipEventDestroy(w.done); w.done = nullptr; }
    }
    if (c->pf[1].stream) { hipStreamDestroy(c->pf[1].stream); c->pf[1].stream = nullptr; }
    c->pf[0].stream = nullptr;
    if (c->pf_start) { hipEventDestroy(c->pf_start); c->pf_start = nullptr; }
    if (c->pf_tok) { hipFree(c->pf_tok); c->pf_tok = nullptr; }
    if (c->pf_flag) { hipFree(c->pf_flag); c->pf_flag = nullptr; }
    c->pf_ready = false;
}
// Workspaces, the second lane's stream, the KV events and the kernels' dynamic-LDS limits.  pf_ready is set only after
// the LAST step succeeded; a failure part-way frees everything again, so a later llmk_prefill retries the setup instead
// of launching on null buffers.
int pf_setup_inner(llmk_ctx* c) {
    const size_t T = PF_TMAX;
    HIPCHK(dev_alloc(&c->pf_flag, sizeof(unsigned)));
    HIPCHK(hipMemset(c->pf_flag, 0, sizeof(unsigned)));
    const int rows[4] = {c->E + 2 * c->KV, c->E, 2 * c->H, c->E};
    size_t pcap = 0;
    const int Ks[4] = {c->E, c->E, c->E, c->H};
    for (int i = 0; i < 4; ++i)
        for (int t : {(int)PF_TMAX, 96}) pcap = std::max(pcap, (size_t)pf_max_slots(pf_plan(c, rows[i], Ks[i], t)) * T * rows[i]);   // (the plan may depend on the batch length)
    HIPCHK(pf_prepare(c));
    for (int i = 0; i < 2; ++i) {
        PfLane& w = c->pf[i];
        HIPCHK(dev_alloc(&w.X, T * c->E * sizeof(float)));
        HIPCHK(dev_alloc(&w.Xs, T * c->E * sizeof(float)));
        HIPCHK(dev_alloc(&w.Q, T * c->E * sizeof(float)));
        HIPCHK(dev_alloc(&w.XB, T * c->E * sizeof(float)));
        HIPCHK(dev_alloc(&w.HB, T * c->H * sizeof(float)));
        HIPCHK(dev_alloc(&w.P, pcap * sizeof(float)));
        HIPCHK(dev_alloc(&w.xn, T * sizeof(float)));
        HIPCHK(dev_alloc(&w.lowcnt, 2 * T * sizeof(unsigned)));
        HIPCHK(hipMemset(w.lowcnt, 0, 2 * T * sizeof(unsigned)));
        if (i == 0) w.stream = c->stream;
        else HIPCHK(hipStreamCreateWithFlags(&w.stream, hipStreamNonBlocking));
        w.kv.assign(c->L, nullptr);
        for (int l = 0; l < c->L; ++l) HIPCHK(hipEventCreateWithFlags(&w.kv[l], hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&w.done, hipEventDisableTiming));
    }
    HIPCHK(hipEventCreateWithFlags(&c->pf_start, hipEventDisableTiming));
    HIPCHK(dev_alloc(&c->pf_tok, (size_t)c->S * sizeof(int)));
    return LLMK_OK;
}
// choose: a setup from scratch also picks the matrix instruction (the f16 one unless LLMK_PF_F32_MFMA=1) -- llmk_prefill and the
// timing hook both say so, so that a hook called first (or after a redo that tore the workspaces down) cannot leave the context on
// the slow instruction (advisor, round 5); the redo in llmk_prefill sets pf_hm = false itself and does not
int pf_setup(llmk_ctx* c, bool choose = false) {
    if (c->pf_ready) return LLMK_OK;
    if (choose) c->pf_hm = !(getenv("LLMK_PF_F32_MFMA") && getenv("LLMK_PF_F32_MFMA")[0] == '1');
    const int rc = pf_setup_inner(c);
    if (rc) { pf_teardown(c); return rc; }
    c->pf_ready = true;
    return LLMK_OK;
}
// dynamic-LDS request of pf_gemm_kernel<NG, *, NR> (> half of the CU's 160 KB: pins one workgroup per CU, so every CU
// gets one block of equal length)
template <int NG, int NR>
constexpr size_t pf_gemm_smem() {
    const size_t need = ((size_t)2 * NG * 16 * PF_LDW + (size_t)NG * 16 * (64 * NR + PF_TPAD)) * sizeof(float);
    return need > (size_t)84 * 1024 ? need : (size_t)84 * 1024;
}
template <int NG, int NR>
constexpr size_t pf_gemm_h_smem() {
    const size_t need = (size_t)4 * NG * 16 * PF_HP * sizeof(_Float16) + (size_t)NG * 16 * (64 * NR + PF_TPAD) * sizeof(float);
    return need > (size_t)84 * 1024 ? need : (size_t)84 * 1024;
}
constexpr size_t pf_attn_smem(int hs) { return ((size_t)PF_ATT_WAVES * 16 * hs + 2 * PF_ATT_WAVES * 16) * sizeof(float); }
// The limits are raised ONCE, in pf_setup (as g_prepare does for the decode kernels): a launch is a launch, with no
// runtime call in the 5 us gaps between the prefill's dependent kernels, and an LDS failure surfaces at setup.
template <int NG, int WT>
hipError_t pf_gemm_prepare_one() {
    HIPRET(hipFuncSetAttribute((const void*)pf_gemm_kernel<NG, WT, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pf_gemm_smem<NG, 1>()));
    return hipFuncSetAttribute((const void*)pf_gemm_kernel<NG, WT, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pf_gemm_smem<NG, 2>());
}
template <int NG, int WT>
hipError_t pf_gemm_h_prepare_one() {
    HIPRET(hipFuncSetAttribute((const void*)pf_gemm_h_kernel<NG, 1, WT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pf_gemm_h_smem<NG, 1>()));
    return hipFuncSetAttribute((const void*)pf_gemm_h_kernel<NG, 2, WT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pf_gemm_h_smem<NG, 2>());
}
template <int WT>
hipError_t pf_gemm_h_prepare() {
    if constexpr (WT == WT_Q4_0)      // (the eight-wave form of the 128-row strip: pf_gemm_launch)
        HIPRET(hipFuncSetAttribute((const void*)pf_gemm_h_kernel<8, 1, WT, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pf_gemm_h_smem<8, 2>()));
    HIPRET((pf_gemm_h_prepare_one<1, WT>())); HIPRET((pf_gemm_h_prepare_one<2, WT>())); HIPRET((pf_gemm_h_prepare_one<3, WT>())); HIPRET((pf_gemm_h_prepare_one<4, WT>()));
    HIPRET((pf_gemm_h_prepare_one<5, WT>())); HIPRET((pf_gemm_h_prepare_one<6, WT>())); HIPRET((pf_gemm_h_prepare_one<7, WT>()));
    return pf_gemm_h_prepare_one<8, WT>();
}
template <int WT>
hipError_t pf_gemm_prepare() {
    HIPRET((pf_gemm_prepare_one<1, WT>())); HIPRET((pf_gemm_prepare_one<2, WT>())); HIPRET((pf_gemm_prepare_one<3, WT>()));
    HIPRET((pf_gemm_prepare_one<4, WT>())); HIPRET((pf_gemm_prepare_one<5, WT>())); HIPRET((pf_gemm_prepare_one<6, WT>()));
    HIPRET((pf_gemm_prepare_one<7, WT>()));
    return pf_gemm_prepare_one<8, WT>();
}
hipError_t pf_prepare(const llmk_ctx* c) {
    switch (c->cfg.weight_type) {
        case LLMK_TYPE_Q4_0: HIPRET(pf_gemm_prepare<WT_Q4_0>()); HIPRET(pf_gemm_h_prepare<WT_Q4_0>()); break;
        case LLMK_TYPE_F16: HIPRET(pf_gemm_prepare<WT_F16>()); HIPRET(pf_gemm_h_prepare<WT_F16>()); break;
        default: HIPRET(pf_gemm_prepare<WT_F32>()); HIPRET(pf_gemm_h_prepare<WT_F32>()); break;
    }
    const int smem = (int)pf_attn_smem(c->hs);
    switch (c->hs) {
        case 16: return hipFuncSetAttribute((const void*)pf_attn_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        case 32: return hipFuncSetAttribute((const void*)pf_attn_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        case 64: return hipFuncSetAttribute((const void*)pf_attn_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        case 128: return hipFuncSetAttribute((const void*)pf_attn_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    }
    return hipErrorInvalidValue;
}
template <int NG, int NR>
hipError_t pf_gemm_launch(llmk_ctx* c, const PfLane& w, const PfGemmArgs& a, const PfPlan& p) {
    constexpr size_t smem = pf_gemm_smem<NG, NR>();
    const dim3 grid(p.grid), block(PF_WAVES * WAVE);
    if (c->pf_hm) {
        constexpr size_t smem_h = pf_gemm_h_smem<NG, NR>();
        if constexpr (NG == 8 && NR == 2) {
            // q4_0, whole batches: the same 128-row strips on EIGHT waves of one row group each -- two waves per SIMD, one wave's nibble
            // arithmetic under the other's matrix instructions (round 5, profiles/r05_prefill_eight_waves.txt: w1|w3 GEMM 33.0 -> 30.9 us on
            // TinyLlama q4_0, 96.2 -> 88.8 at 7B; f16 and f32 weights, which have no such arithmetic, lose 4-6 % to the doubled LDS reads)
            if (c->cfg.weight_type == LLMK_TYPE_Q4_0) {
                hipLaunchKernelGGL((pf_gemm_h_kernel<8, 1, WT_Q4_0, 8>), grid, dim3(8 * WAVE), smem_h, w.stream, a, c->pf_flag, w.lowcnt);
                return hipGetLastError();
            }
        }
        if (c->cfg.weight_type == LLMK_TYPE_Q4_0) hipLaunchKernelGGL((pf_gemm_h_kernel<NG, NR, WT_Q4_0>), grid, block, smem_h, w.stream, a, c->pf_flag, w.lowcnt);
        else if (c->cfg.weight_type == LLMK_TYPE_F16) hipLaunchKernelGGL((pf_gemm_h_kernel<NG, NR, WT_F16>), grid, block, smem_h, w.stream, a, c->pf_flag, w.lowcnt);
        else hipLaunchKernelGGL((pf_gemm_h_kernel<NG, NR, WT_F32>), grid, block, smem_h, w.stream, a, c->pf_flag, w.lowcnt);
        return hipGetLastError();
    }
    if (c->cfg.weight_type == LLMK_TYPE_Q4_0) hipLaunchKernelGGL((pf_gemm_kernel<NG, WT_Q4_0, NR>), grid, block, smem, w.stream, a);
    else if (c->cfg.weight_type == LLMK_TYPE_F16) hipLaunchKernelGGL((pf_gemm_kernel<NG, WT_F16, NR>), grid, block, smem, w.stream, a);
    else hipLaunchKernelGGL((pf_gemm_kernel<NG, WT_F32, NR>), grid, block, smem, w.stream, a);
    return hipGetLastError();
}
hipError_t pf_gemm(llmk_ctx* c, const PfLane& w, const void* W, int row_stride, const float* X, int rows, int K, int T, PfEpiArgs* e) {
    const PfPlan p = pf_plan(c, rows, K, T);
    PfGemmArgs a;
    a.W = W; a.X = X; a.P = w.P; a.rows = rows; a.K = K; a.T = T; a.RS = row_stride;
    a.nk = p.nk; a.U = p.U; a.total = p.total;
#ifdef LLMK_PF_TRACE
    a.trace = (unsigned long long*)(X == w.HB ? w.Q : w.HB);   // debug build: stamps land in the SwiGLU buffer (llmk_peek 7); the w2 GEMM reads that one: its stamps go to the (dead) q buffer
#endif
    e->U = p.U; e->nk = p.nk; e->sh = p.nr == 2 ? 7 : 6;
    e->lowcnt = c->pf_hm ? w.lowcnt : nullptr; e->flag = c->pf_flag; e->gemm_blocks = p.grid;
#define PF_CASE(NG_)                                                                         \
    case NG_: return p.nr == 2 ? pf_gemm_launch<NG_, 2>(c, w, a, p) : pf_gemm_launch<NG_, 1>(c, w, a, p)
    switch ((T + 15) / 16) {
        PF_CASE(1); PF_CASE(2); PF_CASE(3); PF_CASE(4); PF_CASE(5); PF_CASE(6); PF_CASE(7);
        default: return p.nr == 2 ? pf_gemm_launch<8, 2>(c, w, a, p) : pf_gemm_launch<8, 1>(c, w, a, p);
    }
#undef PF_CASE
}
// one batch of T <= PF_TMAX prompt positions pos0 .. pos0+T-1 (1-based) through all layers; X[T-1] ends up in d_x
// `prev`: the lane of the batch before this one (null for the first batch of a call); `last`: this batch ends the prompt
hipError_t pf_batch(llmk_ctx* c, PfLane& w, const PfLane* prev, const int* tok, int T, int pos0, bool last) {
    const int E = c->E, H = c->H, KV = c->KV, QKV = E + 2 * KV, Tp = (T + 15) / 16 * 16;
    const float* emb = (const float*)c->t[LLMK_TOKEN_EMBEDDING_TABLE].data;
    hipLaunchKernelGGL(pf_embed_kernel, dim3((E + 255) / 256, T), dim3(256), 0, w.stream, emb, tok, w.X, E);
    HIPRET(hipGetLastError());
    PfEpiArgs e;
    memset(&e, 0, sizeof(e));
    e.P = w.P; e.xn = w.xn; e.rope = c->d_rope; e.Tp = Tp; e.T = T; e.pos0 = pos0;
    e.E = E; e.KV = KV; e.hs = c->hs; e.H = H;
    for (int l = 0; l < c->L; ++l) {
        // layer l of tensor tid: data plane (and the q4_0 scale plane), rows_per_layer rows
        auto gemm = [&](int tid, int rows_per_layer, const float* X, int K) -> hipError_t {
            const DevTensor& dt = c->t[tid];
            const char* wt = (const char*)dt.data + (size_t)l * rows_per_layer * dt.row_bytes;
            return pf_gemm(c, w, wt, (int)dt.row_bytes, X, rows_per_layer, K, T, &e);
        };
        float* kc = c->d_kc + (size_t)l * c->S * KV;
        float* vc = c->d_vc + (size_t)l * c->S * KV;
        // rmsnorm (layer 0 here, later layers in the previous residual's kernel) + QKV + RoPE + KV write   llama2.f90:527-565
        if (l == 0) {
            hipLaunchKernelGGL(pf_norm_kernel, dim3(T), dim3(256), 0, w.stream, w.X,
                               (const float*)c->t[LLMK_RMS_ATT_WEIGHT].data, w.Xs, w.xn, E, c->eps);
            HIPRET(hipGetLastError());
        }
        HIPRET(gemm(LLMK_WQKV, QKV, w.Xs, E));
        e.rows = QKV; e.out = w.Q; e.kc = kc; e.vc = vc;
        hipLaunchKernelGGL(pf_epi_qkv_kernel, dim3((QKV / 4 + 255) / 256, T), dim3(256), 0, w.stream, e);
        HIPRET(hipGetLastError());
        if (!last) HIPRET(hipEventRecord(w.kv[l], w.stream));             // the next batch's attention reads these rows
        if (prev) HIPRET(hipStreamWaitEvent(w.stream, prev->kv[l], 0));   // ... as this one reads the previous batch's
        // causal attention: position pos0+t sees cache rows 0 .. pos0+t-1                 :572-598
#define ATT(HS_)                                                                                                         \
    do {                                                                                                                 \
        const size_t smem = pf_attn_smem(HS_);                                                                           \
        hipLaunchKernelGGL((pf_attn_kernel<HS_>), dim3(c->nh, (T + 15) / 16), dim3(PF_ATT_WAVES * WAVE), smem, w.stream, \
                           w.Q, kc, vc, w.XB, KV, c->kv_mul, pos0, T, E);                                        \
    } while (0)
        switch (c->hs) {
            case 16: ATT(16); break;
            case 32: ATT(32); break;
            case 64: ATT(64); break;
            case 128: ATT(128); break;
            default: return hipErrorInvalidValue;
        }
#undef ATT
        HIPRET(hipGetLastError());
        // x += wo . xb                                                                    :603-605
        HIPRET(gemm(LLMK_WO, E, w.XB, E));
        e.rows = E; e.out = w.X;
        hipLaunchKernelGGL(pf_epi_resid_norm_kernel, dim3(T), dim3(1024), 0, w.stream, e,
                           (const float*)c->t[LLMK_RMS_FFN_WEIGHT].data + (size_t)l * E, w.Xs, w.xn, c->eps);
        HIPRET(hipGetLastError());
        // (rmsnorm above) + w1|w3 + SwiGLU                                                :608-616
        HIPRET(gemm(LLMK_W13, 2 * H, w.Xs, E));
        e.rows = 2 * H; e.out = w.HB;
        hipLaunchKernelGGL(pf_epi_swiglu_kernel, dim3((H / 4 + 255) / 256, T), dim3(256), 0, w.stream, e);
        HIPRET(hipGetLastError());
        // x += w2 . hb                                                                    :618-620
        HIPRET(gemm(LLMK_W2, E, w.HB, H));
        e.rows = E; e.out = w.X;
        hipLaunchKernelGGL(pf_epi_resid_norm_kernel, dim3(T), dim3(1024), 0, w.stream, e,
                           l + 1 < c->L ? (const float*)c->t[LLMK_RMS_ATT_WEIGHT].data + (size_t)(l + 1) * E : nullptr, w.Xs,
                           w.xn, c->eps);
        HIPRET(hipGetLastError());
    }
    if (last) HIPRET(hipMemcpyAsync(c->d_x, w.X + (size_t)(T - 1) * E, (size_t)E * sizeof(float), hipMemcpyDeviceToDevice, w.stream));
    return hipEventRecord(w.done, w.stream);
}

}  // namespace

extern "C" {

int llmk_version(void) { return 100; }

const char* llmk_strerror(int code) {
    switch (code) {
        case LLMK_OK: return "ok";
        case LLMK_E_ARG: return "llmk: bad argument";
        case LLMK_E_SHAPE: return "llmk: unsupported model shape";
        case LLMK_E_SIZE: return "llmk: byte count does not match tensor size";
        case LLMK_E_TYPE: return "llmk: unsupported ggml tensor type";
        case LLMK_E_STATE: return "llmk: forward called before all weights were uploaded";
        case LLMK_E_NODEVICE: return "llmk: no usable HIP device (there is no CPU fallback)";
        case LLMK_E_NOMEM: return "llmk: out of memory";
        case LLMK_E_TIMEOUT: return "llmk: device-side exchange timed out (persistent kernel not fully resident?)";
        case LLMK_E_COMM: return "llmk: tensor-parallel communicator missing or RCCL error";
        case LLMK_E_VERIFY: return "llmk: uploaded weights did not arrive intact on the device (three attempts)";
        case LLMK_E_NONFINITE: return "llmk: no finite logit at this position (the greedy pick has no answer)";
    }
    if (code >= LLMK_E_HIP) return hipGetErrorString((hipError_t)(code - LLMK_E_HIP));
    return "llmk: unknown error";
}

int llmk_create(const llmk_config* cfg, llmk_ctx** out) { return llmk_create_tp(cfg, 0, 1, out); }

int llmk_create_tp(const llmk_config* cfg, int tp_rank, int tp_size, llmk_ctx** out) {
    if (!cfg || !out) return LLMK_E_ARG;
    *out = nullptr;
    if (tp_size < 1 || tp_rank < 0 || tp_rank >= tp_size) return LLMK_E_ARG;
    const int E = cfg->emb_dim, H = cfg->hidden_dim, L = cfg->n_layers, nh = cfg->n_heads, nkv = cfg->n_kv_heads;
    const int V = cfg->vocab_size, S = cfg->seq_len;
    if (E <= 0 || H <= 0 || L <= 0 || nh <= 0 || nkv <= 0 || V <= 0 || S <= 0) return LLMK_E_SHAPE;
    if (E % nh || nh % nkv) return LLMK_E_SHAPE;
    const int hs = E / nh;
    if (hs != 16 && hs != 32 && hs != 64 && hs != 128) return LLMK_E_SHAPE;
    if (E % 32 || H % 32 || (V & 1)) return LLMK_E_SHAPE;  // 16-byte vectors, q4_0 blocks, row pairs
    // tensor parallelism splits by kv head (each rank: nkv/P kv heads + their nh/P query heads), hidden
    // rows and vocab rows; every local extent must keep the alignment rules above
    const int kalign = cfg->weight_type == LLMK_TYPE_Q4_0 ? 32 : cfg->weight_type == LLMK_TYPE_F16 ? 8 : 4;
    if (nkv % tp_size || H % tp_size || V % tp_size || (H / tp_size) % kalign || ((V / tp_size) & 1)) return LLMK_E_SHAPE;
    // the contraction slices of wo (the local heads' columns) and of w2 must start and end on the weight type's
    // column granule (q4_0: a 32-weight block), and the local K/V rows must keep their RoPE pairs
    if (((nh / tp_size) * hs) % kalign || (((nkv / tp_size) * hs) & 1)) return LLMK_E_SHAPE;
    if (cfg->weight_type != LLMK_TYPE_F32 && cfg->weight_type != LLMK_TYPE_F16 && cfg->weight_type != LLMK_TYPE_Q4_0)
        return LLMK_E_TYPE;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return LLMK_E_NODEVICE;
    if (cfg->device < 0 || cfg->device >= ndev) return LLMK_E_NODEVICE;
    HIPCHK(hipSetDevice(cfg->device));

    llmk_ctx* c = new (std::nothrow) llmk_ctx();
    if (!c) return LLMK_E_NOMEM;
    c->cfg = *cfg;
    c->E = E; c->H = H; c->L = L; c->nh = nh; c->nkv = nkv; c->V = V; c->S = S;
    c->hs = hs; c->KV = nkv * hs; c->kv_mul = nh / nkv;
    c->tp_rank = tp_rank; c->tp_size = tp_size;
    c->nhl = nh / tp_size; c->Eq = c->nhl * hs; c->KVl = (nkv / tp_size) * hs; c->Hl = H / tp_size; c->Vl = V / tp_size;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, cfg->device) == hipSuccess && prop.multiProcessorCount > 0)
        c->n_cu = prop.multiProcessorCount;

    const int KV = c->KV;
    c->gdesc[LLMK_TOKEN_EMBEDDING_TABLE] = {false, V, E, false};
    c->gdesc[LLMK_RMS_ATT_WEIGHT] = {true, 1, E, false};
    c->gdesc[LLMK_RMS_FFN_WEIGHT] = {true, 1, E, false};
    c->gdesc[LLMK_WQKV] = {true, E + 2 * KV, E, true};
    c->gdesc[LLMK_WO] = {true, E, E, true};
    c->gdesc[LLMK_W13] = {true, 2 * H, E, true};
    c->gdesc[LLMK_W2] = {true, E, H, true};
    c->gdesc[LLMK_RMS_FINAL_WEIGHT] = {false, 1, E, false};
    c->gdesc[LLMK_WCLS] = {false, V, E, true};
    for (int i = 0; i < LLMK_N_TENSORS; ++i) c->desc[i] = c->gdesc[i];
    // what THIS rank keeps: column-parallel qkv / w1|w3 / classifier (rows), row-parallel wo / w2 (input slice)
    c->desc[LLMK_WQKV] = {true, c->Eq + 2 * c->KVl, E, true};
    c->desc[LLMK_WO] = {true, E, c->Eq, true};
    c->desc[LLMK_W13] = {true, 2 * c->Hl, E, true};
    c->desc[LLMK_W2] = {true, E, c->Hl, true};
    c->desc[LLMK_WCLS] = {false, c->Vl, E, true};

    int rc = LLMK_OK;
#define CK(expr)                                                        \
    do {                                                                \
        hipError_t e_ = (expr);                                         \
        if (e_ != hipSuccess && rc == LLMK_OK) rc = LLMK_E_HIP + (int)e_; \
    } while (0)
    for (int i = 0; i < LLMK_N_TENSORS && rc == LLMK_OK; ++i) {
        const TensorDesc& d = c->desc[i];
        DevTensor& t = c->t[i];
        t.type = d.matrix ? cfg->weight_type : LLMK_TYPE_F32;
        const size_t rows = (size_t)d.rows * (d.layered ? L : 1);
        // TENSOR_SLACK bytes past the last row: the persistent kernel's last tile of a CU's row range may cover a few rows
        // beyond it (token_kernel.h, NT_Q / NG_A), which for the last rows of the last layer lie past the tensor
        // q4_0 device row: K/2 nibble bytes (16-byte vectors, one per 32-weight block), then the K/32 f16 block scales,
        // zero-padded to whole groups of 64 (a wave-wide scale load past a ragged row end must read finite zeros)
        t.row_bytes = dev_row_bytes(t.type, d.K);
        // the three rmsnorm gain tensors share ONE allocation, att [L][E] | ffn [L][E] | final [E]: the persistent kernel
        // addresses them from one pointer (token_kernel.h TokenArgs::rms)
        if (i == LLMK_RMS_FFN_WEIGHT || i == LLMK_RMS_FINAL_WEIGHT) {
            t.alias = true;
            t.data = (char*)c->t[LLMK_RMS_ATT_WEIGHT].data + (size_t)(i == LLMK_RMS_FFN_WEIGHT ? L : 2 * L) * E * sizeof(float);
            continue;
        }
        const size_t extra = i == LLMK_RMS_ATT_WEIGHT ? ((size_t)L + 1) * E * sizeof(float) : 0;
        CK(dev_alloc(&t.data, rows * t.row_bytes + extra + TENSOR_SLACK));
        if (rc == LLMK_OK) {
            if (t.type == LLMK_TYPE_Q4_0) CK(hipMemset(t.data, 0, rows * t.row_bytes + TENSOR_SLACK));
            else CK(hipMemset((char*)t.data + rows * t.row_bytes + extra, 0, TENSOR_SLACK));
        }
    }
    const size_t kvn = (size_t)L * S * c->KVl;
    CK(dev_alloc(&c->d_kc, kvn * sizeof(float)));
    CK(dev_alloc(&c->d_vc, kvn * sizeof(float)));
    CK(dev_alloc(&c->d_x, (size_t)E * sizeof(float)));
    CK(dev_alloc(&c->d_q, (size_t)E * sizeof(float)));
    CK(dev_alloc(&c->d_xb, (size_t)E * sizeof(float)));
    CK(dev_alloc(&c->d_hb, (size_t)H * sizeof(float)));
    CK(dev_alloc(&c->d_part, (size_t)E * sizeof(float)));
    // [V] = sticky device error word; behind it (at V + 4) the two candidate buffers of the pipelined greedy decode
    CK(dev_alloc(&c->d_logits, ((size_t)V + 4 + 4 * TK_NCU) * sizeof(float)));
    CK(dev_alloc(&c->d_rope, (size_t)(hs / 2) * sizeof(float)));
    CK(dev_alloc(&c->d_tokpos, 4 * sizeof(int)));
    CK(dev_alloc(&c->d_next, 2 * sizeof(int)));
    CK(hipHostMalloc(&c->h_tokpos, 4 * sizeof(int), hipHostMallocDefault));
    // [V] = the error word as the host sees it; behind it (at V + 4) the ids of llmk_decode_greedy, S ints
    CK(hipHostMalloc(&c->h_logits, ((size_t)V + 4 + S) * sizeof(float), hipHostMallocMapped));
    CK(hipHostGetDevicePointer((void**)&c->h_logits_dev, c->h_logits, 0));
    c->tk_direct = !(getenv("LLMK_TK_DIRECT") && getenv("LLMK_TK_DIRECT")[0] == '0');
    CK(hipHostMalloc(&c->h_next, 2 * sizeof(int), hipHostMallocDefault));
    CK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    for (int i = 0; i < 8; ++i) CK(hipEventCreate(&c->ev[i]));
    // The whole-token persistent kernel serves the shapes it is instantiated for, on a full 256-CU part
    if (rc == LLMK_OK) rc = tk_setup_all(c);
    if (rc == LLMK_OK) {   // raise the dynamic-LDS limits the token pass needs, or reject the shape here (see g_prepare)
        g_prepare = true;
        hipError_t pe = launch_qkv(c, 0);
        if (pe == hipSuccess) pe = launch_attn(c, 0);
        if (pe == hipSuccess) pe = launch_wo(c, 0);
        if (pe == hipSuccess) pe = launch_w13(c, 0);
        if (pe == hipSuccess) pe = launch_w2(c, 0);
        if (pe == hipSuccess) pe = launch_cls(c);
        g_prepare = false;
        if (pe == hipErrorInvalidValue) rc = LLMK_E_SHAPE;   // context or contraction width needs more than 160 KB of LDS
        else CK(pe);
    }
    if (rc == LLMK_OK) {
        CK(hipMemset(c->d_logits, 0, ((size_t)V + 4 + 4 * TK_NCU) * sizeof(float)));
        c->h_tokpos[0] = 0; c->h_tokpos[1] = 0; c->h_tokpos[2] = 0; c->h_tokpos[3] = 0;
        CK(hipMemset(c->d_kc, 0, kvn * sizeof(float)));  // s%key_cache(:,:,:) = 0   llama2.f90:317
        CK(hipMemset(c->d_vc, 0, kvn * sizeof(float)));
        CK(hipMemset(c->d_x, 0, (size_t)E * sizeof(float)));
        CK(hipMemset(c->d_q, 0, (size_t)E * sizeof(float)));
        CK(hipMemset(c->d_xb, 0, (size_t)E * sizeof(float)));
        CK(hipMemset(c->d_hb, 0, (size_t)H * sizeof(float)));
        // default RoPE table: freq_j = 1/10000**((2j+1)/hs)   (llama2.f90:544-545)
        float fr[64];
        for (int j = 0; j < hs / 2; ++j) fr[j] = 1.0f / powf(10000.0f, (float)(2 * j + 1) / (float)hs);
        CK(hipMemcpy(c->d_rope, fr, (size_t)(hs / 2) * sizeof(float), hipMemcpyHostToDevice));
        CK(hipDeviceSynchronize());
    }
#undef CK
    if (rc != LLMK_OK) {
        llmk_destroy(c);
        return rc;
    }
    *out = c;
    return LLMK_OK;
}

// ---- every upload is staged through the shim's own pinned memory, and verified ------------------------------------------------
// Round 4, found by checksums: twice in ~40 runs of the 8-rank 70B tests ONE block of ONE rank reached the device with
// other bytes than the host held -- once from a read-only mmap of a /dev/shm file (logits 3.6e-3 off on every rank), once from
// an anonymous numpy temporary, and that time THREE successive hipMemcpy2D calls from the same pointer delivered the same wrong
// bytes (profiles/r04_tp70_upload_damage.txt): the address had just been freed and mapped again by the allocator (one 264 MB
// temporary per layer), and a copy engine that reads pageable memory through a cached registration of the virtual range reads
// the pages that USED to be there.  Weights are copied once and read for the life of the ctx, so the shim no longer lets
// the device read caller memory at all: the CPU copies each chunk into one of two pinned staging buffers (8 MB, allocated with
// the first upload), summing its 16-bit words on the way.  THAT was not it either: the third event came with the staging in
// place, and all three have the same signature -- the device's sum is 7/8 of the host's to four digits, on every retry: an
// eighth of a freshly copied 16.5 MB scratch buffer reads as zeros to the kernel that follows the copy (one of the chip's
// eight XCDs, each with its own L2, seeing the buffer as it was before a copy ENGINE wrote it).  So now no kernel ever reads what
// a copy engine wrote: a kernel moves each chunk out of the (device-mapped) staging buffer into the tensor (q4_0: via a scratch
// the shim keeps, re-packed on the way), and the 64-bit sum of the tensor's rows AS THEY THEN ARE is compared with the host's.
// A mismatch is printed and the block staged again (three attempts, then LLMK_E_VERIFY).  ~0.15 s per GB on one host core;
// LLMK_VERIFY_UPLOAD=0 skips the comparison, never the staging.
constexpr size_t UP_STAGE_BYTES = (size_t)8 << 20;
__global__ void sum16_kernel(const unsigned* __restrict__ w, size_t nbytes, unsigned long long* out) {
    const size_t nwords = nbytes / 4;
    unsigned long long t = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += (size_t)gridDim.x * blockDim.x) {
        const unsigned v = w[i];
        t += (v & 0xffffu) + (v >> 16);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && (nbytes & 2)) t += reinterpret_cast<const unsigned short*>(w)[nbytes / 2 - 1];
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, t);
}
// rows [0, nrows) x col_bytes at pitch spitch -> dst (contiguous); returns the 64-bit sum of the 16-bit words copied
static unsigned long long stage_rows(uint8_t* dst, const uint8_t* src, size_t spitch, size_t col_bytes, size_t nrows) {
    unsigned long long t = 0;
    for (size_t r = 0; r < nrows; ++r) {
        const uint16_t* p = reinterpret_cast<const uint16_t*>(src + r * spitch);    // rows of every encoding start 2-byte aligned
        uint16_t* q = reinterpret_cast<uint16_t*>(dst + r * col_bytes);
        unsigned long long a = 0;
        for (size_t i = 0; i < col_bytes / 2; ++i) { const uint16_t v = p[i]; q[i] = v; a += v; }
        t += a;
    }
    return t;
}
// staging buffer (host memory, device-mapped) -> tensor rows: f32 / f16 rows are contiguous on both sides
__global__ void stage_copy_kernel(const unsigned* __restrict__ src, unsigned* __restrict__ dst, size_t nwords) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
static bool verify_uploads() {
    static const bool on = !(getenv("LLMK_VERIFY_UPLOAD") && getenv("LLMK_VERIFY_UPLOAD")[0] == '0');
    return on;
}
// device sum of `nbytes` contiguous bytes (null stream, like the copies before it); 0 + error code on failure
static hipError_t device_sum16(const void* dev, size_t nbytes, unsigned long long* d_acc, unsigned long long* out) {
    hipError_t e = hipMemsetAsync(d_acc, 0, sizeof(*d_acc), 0);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(sum16_kernel, dim3(512), dim3(256), 0, 0, (const unsigned*)dev, nbytes, d_acc);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpy(out, d_acc, sizeof(*out), hipMemcpyDeviceToHost);
    return e;
}

// q4_0 + persistent kernel: (re)build the unit layout of the five matrices from their rows; with the upload verification on,
// the 16-bit word sum of every unit image must equal that of the rows it was built from (the padding blocks are zeros)
static int q16_build(llmk_ctx* c) {
    static const int mats[5] = {LLMK_WQKV, LLMK_WO, LLMK_W13, LLMK_W2, LLMK_WCLS};
    for (int i = 0; i < 5; ++i) {
        const int tid = mats[i];
        const TensorDesc& d = c->desc[tid];
        const DevTensor& t = c->t[tid];
        if (tid == LLMK_WCLS && t.type == LLMK_TYPE_Q6_K) continue;      // q6_K classifier rows are dotted as rows (q6k.h)
        if (t.type != LLMK_TYPE_Q4_0 || d.rows % Q16_ROWS) return LLMK_E_SHAPE;
        const size_t rows = (size_t)d.rows * (d.layered ? c->L : 1), bytes = q16_bytes(rows, d.K);
        if (!c->q16[tid]) {
            HIPCHK(dev_alloc(&c->q16[tid], bytes + TENSOR_SLACK));
            HIPCHK(hipMemset((char*)c->q16[tid] + bytes, 0, TENSOR_SLACK));
        }
        hipLaunchKernelGGL(q16_units_kernel, dim3((unsigned)(bytes / Q16_UNIT_BYTES)), dim3(64), 0, c->stream, (const char*)t.data, t.row_bytes, d.K,
                           q16_ncs(d.K), (char*)c->q16[tid]);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    if (verify_uploads()) {
        unsigned long long* d_acc = nullptr;
        HIPCHK(dev_alloc(&d_acc, sizeof(*d_acc)));
        int rc = LLMK_OK;
        for (int i = 0; i < 5 && rc == LLMK_OK; ++i) {
            const int tid = mats[i];
            if (!c->q16[tid] || c->t[tid].type != LLMK_TYPE_Q4_0) continue;
            const TensorDesc& d = c->desc[tid];
            const size_t rows = (size_t)d.rows * (d.layered ? c->L : 1);
            unsigned long long a = 0, b = 0;
            hipError_t e = device_sum16(c->t[tid].data, rows * c->t[tid].row_bytes, d_acc, &a);
            if (e == hipSuccess) e = device_sum16(c->q16[tid], q16_bytes(rows, d.K), d_acc, &b);
            if (e != hipSuccess) rc = LLMK_E_HIP + (int)e;
            else if (a != b) {
                fprintf(stderr, "llmk: the unit layout of tensor %d differs from its rows (16-bit word sums 0x%llx / 0x%llx)\n", tid, b, a);
                rc = LLMK_E_VERIFY;
            }
        }
        hipFree(d_acc);
        if (rc) return rc;
    }
    c->q16_dirty = false;
    return LLMK_OK;
}

// Copy `nrows` rows of a HOST tensor (row pitch `spitch` bytes in `type` encoding; for each row only the
// bytes [col_off, col_off + col_bytes) -- a contraction slice for the row-parallel wo / w2) into local rows
// dst_row0.. of layer `layer` of tensor `tid`.  q4_0 is re-packed (nibble plane + scale plane) on the way.
static int upload_block(llmk_ctx* c, int tid, int layer, int dst_row0, int nrows, const uint8_t* src, size_t spitch,
                        size_t col_off, size_t col_bytes) {
    const TensorDesc& d = c->desc[tid];
    DevTensor& t = c->t[tid];
    const size_t first_row = (size_t)layer * d.rows + dst_row0;
    const bool verify = verify_uploads();
    int rc = LLMK_OK;
    if (!c->up_ready) {
        // the staging objects live as long as the ctx (llmk_destroy frees whatever exists); `up_ready` only after ALL of them do, so a
        // failure half-way is retried from where it stopped instead of dereferencing what is missing (advisor, round 4)
        for (int b = 0; b < 2; ++b) {
            if (!c->up_stage[b]) HIPCHK(hipHostMalloc(&c->up_stage[b], UP_STAGE_BYTES, hipHostMallocMapped));
            if (!c->up_stage_dev[b]) HIPCHK(hipHostGetDevicePointer(&c->up_stage_dev[b], c->up_stage[b], 0));
            if (!c->up_done[b]) HIPCHK(hipEventCreateWithFlags(&c->up_done[b], hipEventDisableTiming));
            if (!c->up_tmp[b]) HIPCHK(dev_alloc(&c->up_tmp[b], UP_STAGE_BYTES));
        }
        if (!c->up_acc) HIPCHK(dev_alloc(&c->up_acc, sizeof(*c->up_acc)));
        c->up_ready = true;
    }
    unsigned long long* const d_acc = c->up_acc;
    const bool q6 = t.type == LLMK_TYPE_Q6_K;
    const bool q4 = t.type == LLMK_TYPE_Q4_0 || q6;      // re-packed on the way (q6_K: its own kernel, q6k.h)
    // whole 16-bit words on the host side; the non-q4_0 path moves 32-bit words into rows row_bytes apart (advisor, round 4)
    if (col_bytes > UP_STAGE_BYTES || col_bytes % 2 || (!q4 && (col_bytes % 4 || t.row_bytes % 4))) return LLMK_E_ARG;
    if (q6 && (col_off != 0 || col_bytes != row_bytes_for(LLMK_TYPE_Q6_K, d.K))) return LLMK_E_ARG;      // whole rows only (never a contraction slice)
    c->q16_dirty = true;
    const size_t blocks_per_row = col_bytes / 18;                       // (q4_0)
    const size_t rows_per_chunk = UP_STAGE_BYTES / col_bytes;
    char* dst0 = (char*)t.data + first_row * t.row_bytes;              // the block's rows in the tensor (contiguous: row_bytes apart)
    for (int attempt = 1; rc == LLMK_OK; ++attempt) {
        unsigned long long want = 0;
        int b = 0;
        for (size_t r = 0; r < (size_t)nrows && rc == LLMK_OK; r += rows_per_chunk, b ^= 1) {
            const size_t n = (size_t)nrows - r < rows_per_chunk ? (size_t)nrows - r : rows_per_chunk;
            hipError_t e = hipEventSynchronize(c->up_done[b]);             // the kernel that read this buffer's previous chunk has finished
            if (e != hipSuccess) { rc = LLMK_E_HIP + (int)e; break; }
            want += stage_rows((uint8_t*)c->up_stage[b], src + r * spitch + col_off, spitch, col_bytes, n);
            // a KERNEL moves the chunk from the (device-mapped) staging buffer into the tensor -- re-packing q4_0 blocks on the
            // way -- so nothing a copy engine wrote is ever read by a kernel
            if (q4) {   // (wide loads over PCIe into a device scratch first: the re-packing reads 2 bytes at a time)
                hipLaunchKernelGGL(stage_copy_kernel, dim3(1024), dim3(256), 0, 0, (const unsigned*)c->up_stage_dev[b], (unsigned*)c->up_tmp[b],
                                   (n * col_bytes + 3) / 4);
                if (q6) hipLaunchKernelGGL(q6k_repack_kernel, dim3(1024), dim3(256), 0, 0, (const uint8_t*)c->up_tmp[b], dst0 + r * t.row_bytes,
                                           n * (size_t)(d.K / Q6K_WEIGHTS), d.K / Q6K_WEIGHTS, t.row_bytes);
                else
                hipLaunchKernelGGL(q4_repack_kernel, dim3(1024), dim3(256), 0, 0, (const uint8_t*)c->up_tmp[b], dst0 + r * t.row_bytes,
                                   n * blocks_per_row, (int)blocks_per_row, t.row_bytes);
            }
            else hipLaunchKernelGGL(stage_copy_kernel, dim3(1024), dim3(256), 0, 0, (const unsigned*)c->up_stage_dev[b], (unsigned*)(dst0 + r * t.row_bytes),
                                    n * col_bytes / 4);
            e = hipGetLastError();
            if (e == hipSuccess) e = hipEventRecord(c->up_done[b], 0);
            if (e != hipSuccess) rc = LLMK_E_HIP + (int)e;
        }
        if (rc != LLMK_OK || !verify) break;
#ifdef LLMK_TK_DEBUG
        {   // test aid of the debug library (tests/test_upload_verify_gpu.py): damage the first block the process uploads on its first
            // LLMK_UPLOAD_INJECT attempts -- what the verification below exists to catch
            static const int inject = getenv("LLMK_UPLOAD_INJECT") ? atoi(getenv("LLMK_UPLOAD_INJECT")) : 0;
            static bool first_block = true;
            if (first_block && attempt <= inject) (void)hipMemsetAsync(dst0, 0xA5, 64, 0);
            if (attempt >= inject) first_block = false;
        }
#endif
        // the tensor's rows as they now are: a q4_0 row's 16-bit words are its blocks' words in another order (8 nibble words and
        // the scale per block, zero padding behind), so the sum of the re-packed image equals the sum of the bytes handed over
        unsigned long long got = 0;
        const hipError_t e = device_sum16(dst0, (size_t)nrows * t.row_bytes, d_acc, &got);
        if (e != hipSuccess) { rc = LLMK_E_HIP + (int)e; break; }
        if (got == want) break;
        fprintf(stderr, "llmk: upload of tensor %d, layer %d, local rows %zu..%zu (%zu bytes) did not arrive intact: 16-bit word sum 0x%llx on the "
                        "device, 0x%llx on the host (attempt %d of 3)%s\n", tid, layer, first_row, first_row + (size_t)nrows - 1, (size_t)nrows * col_bytes,
                got, want, attempt, attempt < 3 ? " -- staging it again" : "");
        if (attempt == 3) rc = LLMK_E_VERIFY;
    }
    if (rc == LLMK_OK) { const hipError_t e = hipDeviceSynchronize(); if (e != hipSuccess) rc = LLMK_E_HIP + (int)e; }
    if (rc) return rc;
    t.rows_uploaded += (size_t)nrows;
    if (t.rows_uploaded >= (size_t)d.rows * (d.layered ? c->L : 1)) t.uploaded = true;
    return LLMK_OK;
}

// Rows [row0, row0 + nrows) of one layer of a full (unsharded) host tensor -> the part of them this rank's shard holds.
// The shard is a few runs of global rows (its query heads / K rows / V rows; its gate / up rows; its vocabulary rows) or,
// for the row-parallel wo / w2, every row but only the input columns of the local heads / hidden slice: each run is
// intersected with the rows handed over, so a loader may stream any row range -- in particular ONLY the rows of its own
// shard (host/gguf_loader.f90 stream_ggml_matrices) -- and whole layers keep working.
static int upload_rows_sharded(llmk_ctx* c, int tid, int layer, int row0, int nrows, const uint8_t* src, int type) {
    const TensorDesc& g = c->gdesc[tid];
    const size_t pitch = row_bytes_for(type, g.K);
    const int r = c->tp_rank, E = c->E, KV = c->KV, H = c->H;
    struct Run { int gfirst, n, lfirst; };            // global first row, count, local first row
    Run runs[3];
    int nruns = 0;
    size_t col_off = 0, col_bytes = pitch;
    switch (tid) {
        case LLMK_WQKV:  // rows: this rank's query heads, then its kv heads' K rows, then their V rows
            runs[nruns++] = {r * c->Eq, c->Eq, 0};
            runs[nruns++] = {E + r * c->KVl, c->KVl, c->Eq};
            runs[nruns++] = {E + KV + r * c->KVl, c->KVl, c->Eq + c->KVl};
            break;
        case LLMK_WO:    // all E rows, the input columns of this rank's heads
            runs[nruns++] = {0, E, 0};
            col_off = row_bytes_for(type, r * c->Eq); col_bytes = row_bytes_for(type, c->Eq);
            break;
        case LLMK_W13:   // gate rows then up rows of this rank's hidden slice
            runs[nruns++] = {r * c->Hl, c->Hl, 0};
            runs[nruns++] = {H + r * c->Hl, c->Hl, c->Hl};
            break;
        case LLMK_W2:    // all E rows, the input columns of this rank's hidden slice
            runs[nruns++] = {0, E, 0};
            col_off = row_bytes_for(type, r * c->Hl); col_bytes = row_bytes_for(type, c->Hl);
            break;
        case LLMK_WCLS:
            runs[nruns++] = {r * c->Vl, c->Vl, 0};
            break;
        default:         // replicated: embedding table and norm gains
            runs[nruns++] = {0, g.rows, 0};
            break;
    }
    for (int i = 0; i < nruns; ++i) {
        const int a = std::max(runs[i].gfirst, row0), b = std::min(runs[i].gfirst + runs[i].n, row0 + nrows);
        if (a >= b) continue;
        const int rc = upload_block(c, tid, layer, runs[i].lfirst + (a - runs[i].gfirst), b - a, src + (size_t)(a - row0) * pitch, pitch,
                                    col_off, col_bytes);
        if (rc) return rc;
    }
    return LLMK_OK;
}

int llmk_upload_rows(llmk_ctx* c, int tid, int layer, int row_offset, int rows, const void* host, size_t nbytes,
                     int ggml_type) {
    if (!c || !host || tid < 0 || tid >= LLMK_N_TENSORS) return LLMK_E_ARG;
    const TensorDesc& g = c->gdesc[tid];
    DevTensor& t = c->t[tid];
    const int nl = g.layered ? c->L : 1;
    if (layer < 0 || layer >= nl || row_offset < 0 || rows <= 0 || row_offset + rows > g.rows) return LLMK_E_ARG;
    if (ggml_type != t.type) return LLMK_E_TYPE;
    if (nbytes != (size_t)rows * row_bytes_for(ggml_type, g.K)) return LLMK_E_SIZE;
    HIPCHK(hipSetDevice(c->cfg.device));
    if (c->tp_size > 1) return upload_rows_sharded(c, tid, layer, row_offset, rows, (const uint8_t*)host, ggml_type);
    const size_t pitch = row_bytes_for(ggml_type, g.K);
    return upload_block(c, tid, layer, row_offset, rows, (const uint8_t*)host, pitch, 0, pitch);
}

int llmk_upload(llmk_ctx* c, int tid, const void* host, size_t nbytes, int ggml_type) {
    if (!c || !host || tid < 0 || tid >= LLMK_N_TENSORS) return LLMK_E_ARG;
    const TensorDesc& g = c->gdesc[tid];
    const int nl = g.layered ? c->L : 1;
    if (ggml_type != c->t[tid].type) return LLMK_E_TYPE;
    const size_t per_layer = (size_t)g.rows * row_bytes_for(ggml_type, g.K);
    if (nbytes != per_layer * nl) return LLMK_E_SIZE;
    c->t[tid].rows_uploaded = 0;
    c->t[tid].uploaded = false;
    for (int l = 0; l < nl; ++l) {
        int rc = llmk_upload_rows(c, tid, l, 0, g.rows, (const char*)host + (size_t)l * per_layer, per_layer, ggml_type);
        if (rc) return rc;
    }
    return LLMK_OK;
}

// The classifier's encoding may differ from the other matrices' (stock llama.cpp q4_0 files keep output.weight in
// q6_K; the host loader hands it over dequantised): re-type the tensor BEFORE uploading it.  Such a ctx runs the
// multi-kernel path (the persistent kernel is instantiated for one weight type).
int llmk_set_tensor_type(llmk_ctx* c, int tid, int ggml_type) {
    if (!c || tid != LLMK_WCLS) return LLMK_E_ARG;
    if (ggml_type != LLMK_TYPE_F32 && ggml_type != LLMK_TYPE_F16 && ggml_type != LLMK_TYPE_Q4_0 && ggml_type != LLMK_TYPE_Q6_K) return LLMK_E_TYPE;
    DevTensor& t = c->t[tid];
    if (t.type == ggml_type) return LLMK_OK;
    const TensorDesc& d = c->desc[tid];
    const int kalign = ggml_type == LLMK_TYPE_Q6_K ? Q6K_WEIGHTS : ggml_type == LLMK_TYPE_Q4_0 ? 32 : ggml_type == LLMK_TYPE_F16 ? 8 : 4;
    if (d.K % kalign) return LLMK_E_SHAPE;
    HIPCHK(hipSetDevice(c->cfg.device));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (t.data) HIPCHK(hipFree(t.data));
    t.data = nullptr;
    t.type = ggml_type;
    t.uploaded = false;
    t.rows_uploaded = 0;
    const size_t rows = (size_t)d.rows * (d.layered ? c->L : 1);
    t.row_bytes = dev_row_bytes(ggml_type, d.K);
    HIPCHK(dev_alloc(&t.data, rows * t.row_bytes + TENSOR_SLACK));
    HIPCHK(hipMemset(t.data, 0, rows * t.row_bytes + TENSOR_SLACK));
    if (c->q16[tid]) { HIPCHK(hipFree(c->q16[tid])); c->q16[tid] = nullptr; }      // the classifier's unit copy, if one was built
    c->q16_dirty = true;
    if (c->use_tk) {
        c->use_tk = false;
        if (c->graph_logits) { hipGraphExecDestroy(c->graph_logits); c->graph_logits = nullptr; }
        if (c->graph_greedy) { hipGraphExecDestroy(c->graph_greedy); c->graph_greedy = nullptr; }
    }
    // the persistent kernel again, if one is instantiated for this shape with a classifier of this type (round 6: q6_K rows beside
    // q4_0 matrices -- a stock llama.cpp q4_0 file keeps the fast path); otherwise the multi-kernel path
    { const int rc = tk_setup_all(c); if (rc) return rc; }
    g_prepare = true;                      // the classifier GEMV may need a larger dynamic-LDS limit in its new type
    const hipError_t pe = launch_cls(c);
    g_prepare = false;
    if (pe == hipErrorInvalidValue) return LLMK_E_SHAPE;
    HIPCHK(pe);
    return LLMK_OK;
}

// rmsnorm epsilon.  The reference hard-codes 1e-5 (llama2.f90:454) and ignores the file's
// llama.attention.layer_norm_rms_epsilon; a host that opts into the file's value (llm --gguf-eps) sets it here.
int llmk_set_rms_eps(llmk_ctx* c, float eps) {
    if (!c || !(eps > 0.f) || !(eps < 1.f)) return LLMK_E_ARG;
    HIPCHK(hipSetDevice(c->cfg.device));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->eps = eps;
    if (c->graph_logits) { hipGraphExecDestroy(c->graph_logits); c->graph_logits = nullptr; }   // kernel arguments are baked in
    if (c->graph_greedy) { hipGraphExecDestroy(c->graph_greedy); c->graph_greedy = nullptr; }
    return LLMK_OK;
}

int llmk_set_rope_freqs(llmk_ctx* c, const float* freqs, int n) {
    if (!c || !freqs || n != c->hs / 2) return LLMK_E_ARG;
    HIPCHK(hipSetDevice(c->cfg.device));
    HIPCHK(hipMemcpy(c->d_rope, freqs, (size_t)n * sizeof(float), hipMemcpyHostToDevice));
    return LLMK_OK;
}

int llmk_forward(llmk_ctx* c, int token, int pos, float* logits_out) {
    if (!c || !logits_out) return LLMK_E_ARG;
    int rc = run_token(c, token, pos, false);
    if (rc) return rc;
    memcpy(logits_out, c->h_logits, (size_t)c->V * sizeof(float));
    return LLMK_OK;
}

// The prompt loop of llama2.f90:376-402 as ONE call: tokens[0..n) (1-based ids) sit at positions pos0 .. pos0+n-1,
// the KV cache rows of those positions are written, and logits_out receives the logits of the LAST position --
// exactly what n llmk_forward calls leave behind.  f32 single-GPU contexts run the batched MFMA path (prefill.h);
// every other configuration falls back to the token-by-token pass.
int llmk_prefill(llmk_ctx* c, const int* tokens, int n, int pos0, float* logits_out) {
    if (!c || !tokens || !logits_out || n < 1 || pos0 < 1 || pos0 + n - 1 > c->S) return LLMK_E_ARG;
    int rc = check_ready(c);
    if (rc) return rc;
    for (int i = 0; i < n; ++i)
        if (tokens[i] < 1 || tokens[i] > c->V) return LLMK_E_ARG;
    const int pf_step = PF_KSTEP;
    const bool batched = c->tp_size == 1 && !c->comm && c->E % pf_step == 0 && c->H % pf_step == 0 && c->KV % 16 == 0 && !(getenv("LLMK_PREFILL") && getenv("LLMK_PREFILL")[0] == '0');
    if (!batched) {
        for (int i = 0; i < n; ++i) {
            rc = run_token(c, tokens[i], pos0 + i, false);
            if (rc) return rc;
        }
        memcpy(logits_out, c->h_logits, (size_t)c->V * sizeof(float));
        return LLMK_OK;
    }
    HIPCHK(hipSetDevice(c->cfg.device));
    rc = pf_setup(c, true);
    if (rc) return rc;
    std::vector<int> tok0(tokens, tokens + n);
    for (int& t : tok0) --t;
    HIPCHK(hipMemcpyAsync(c->pf_tok, tok0.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, c->stream));
    // these positions are being rewritten by other kernels: whatever an earlier sequence's decode left in the q4_0 persistent kernel's
    // scale records (token_kernel.h tk_qsc; tagged by position only) must not pass for "the position before" of the decode that follows
    if (c->d_gran && c->tk_ngran)
        HIPCHK(hipMemsetAsync(c->d_gran + c->tk_ngran - 4 * TK_QSC_LMAX, 0, 4 * TK_QSC_LMAX * sizeof(unsigned long long), c->stream));
    HIPCHK(hipEventRecord(c->pf_start, c->stream));
    HIPCHK(hipStreamWaitEvent(c->pf[1].stream, c->pf_start, 0));
    int k = 0;
    for (int i = 0; i < n; i += PF_TMAX, ++k) {
        // batch k runs on lane k % 2; its lane's previous batch (k - 2) is ordered by the stream, batch k - 1 by the KV events
        const bool last = i + PF_TMAX >= n;
        const hipError_t pe = pf_batch(c, c->pf[k & 1], k > 0 ? &c->pf[(k - 1) & 1] : nullptr, c->pf_tok + i, std::min(PF_TMAX, n - i), pos0 + i, last);
        if (pe != hipSuccess) {   // nothing of this call may still be running (on either lane) when it returns
            hipStreamSynchronize(c->pf[1].stream);
            hipStreamSynchronize(c->stream);
            return LLMK_E_HIP + (int)pe;
        }
    }
    // the classifier runs on the ctx stream (= lane 0's): after everything lane 1 was given, too -- nothing of this call is
    // still running when it returns
    if (k >= 2) HIPCHK(hipStreamWaitEvent(c->stream, c->pf[1].done, 0));
    HIPCHK(launch_cls(c));                          // final rmsnorm + classifier of the last position   :627-636
    HIPCHK(hipMemcpyAsync(c->h_logits, c->d_logits, (size_t)c->V * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    unsigned* h_flag = reinterpret_cast<unsigned*>(c->h_logits + c->V + 1);
    *h_flag = 0;
    if (c->pf_hm) HIPCHK(hipMemcpyAsync(h_flag, c->pf_flag, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->pf_hm && *h_flag) {
        // bit 0: an activation of this prompt does not fit an f16 (|x| >= 65504, or not finite): this context's GEMMs go back to
        // the f32 matrix instruction for good.  Bit 1 alone: a position whose whole row is below 2^-7 (the lo piece of the split
        // is an f16 subnormal there): THIS call is redone on the f32 instruction, the next one tries the f16 one again (advisor,
        // round 4: a BOS or early-layer row with genuinely small values must not cost the context the fast path).
        // Either way the call is redone (the K/V rows it wrote are rewritten) and the first event says so once.
        const bool for_good = (*h_flag & 1u) != 0;
        static bool told = false;
        if (!told) {
            told = true;
            fprintf(stderr, "llmk: llmk_prefill met %s; %s\n", for_good ? "an activation beyond the f16 range" : "a position whose activations are all below 2^-7",
                    for_good ? "this context's prompt GEMMs run on the f32 matrix instruction from now on" : "this call is redone on the f32 matrix instruction");
        }
        HIPCHK(hipMemsetAsync(c->pf_flag, 0, sizeof(unsigned), c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        pf_teardown(c);
        c->pf_hm = false;
        rc = pf_setup(c);
        if (rc) return rc;
        rc = llmk_prefill(c, tokens, n, pos0, logits_out);
        if (!for_good) pf_teardown(c);          // (pf_ready = false: the next call sets the f16 instruction up again)
        return rc;
    }
    memcpy(logits_out, c->h_logits, (size_t)c->V * sizeof(float));
    return LLMK_OK;
}

int llmk_forward_greedy(llmk_ctx* c, int token, int pos, int* next_token) {
    if (!c || !next_token) return LLMK_E_ARG;
    int rc = run_token(c, token, pos, true);
    if (rc) return rc;
    // the device argmax answers 0 ("no token") when no logit is finite (a damaged upload, an overflow): an error, not an id --
    // a host that indexes its vocabulary with it reads out of bounds (advisor, round 4)
    if (*c->h_next < 1 || *c->h_next > c->V) return LLMK_E_NONFINITE;
    *next_token = *c->h_next;
    return LLMK_OK;
}

// The temperature-0 generation loop of llama2.f90:379-396 for n positions with no host round trip between them.
int llmk_decode_greedy(llmk_ctx* c, int token, int pos0, int n, int* ids_out, llmk_token_fn on_token, void* user) {
    if (!c || !ids_out || n < 1 || pos0 < 1 || pos0 + n - 1 > c->S || token < 1 || token > c->V) return LLMK_E_ARG;
    int rc = check_ready(c);
    if (rc) return rc;
    const bool timed = (c->cfg.flags & (LLMK_FLAG_TIMINGS | LLMK_FLAG_NO_GRAPH)) != 0;
    int done = 0;
    if (c->use_tk && !timed && !c->p2p && !c->comm && c->tp_size == 1) {
        // n launches enqueued back to back: launch i takes its token from launch i-1's candidates (device memory) and leaves
        // its own; ids reach the host through mapped memory as CU 0 of the NEXT launch resolves them
        HIPCHK(hipSetDevice(c->cfg.device));
        int* h_ids = reinterpret_cast<int*>(c->h_logits + c->V + 4);
        int* h_ids_dev = reinterpret_cast<int*>(c->h_logits_dev + c->V + 4);
        float2* d_cand = reinterpret_cast<float2*>(c->d_logits + c->V + 4);
        memset(h_ids, 0, (size_t)n * sizeof(int));
        reinterpret_cast<unsigned*>(c->h_logits)[c->V] = 0;
        for (int i = 0; i < n; ++i) {
            c->h_tokpos[0] = token - 1;
            c->h_tokpos[1] = pos0 + i;
            c->h_tokpos[2] += 1;
            TkGreedy g;
            g.gflags = TKG_GREEDY | (i ? TKG_CAND_IN | TKG_ID : 0);
            g.id_index = i - 1;
            // debug library only: this launch one workgroup short, so its peers really time out INSIDE the pipeline
            c->tk_short_grid = TK_DEBUG && getenv("LLMK_TK_INJECT_TIMEOUT") && atoi(getenv("LLMK_TK_INJECT_TIMEOUT")) == pos0 + i;
            HIPCHK(launch_token_kernel(c, false, g));
        }
        hipLaunchKernelGGL(cand_resolve_kernel, dim3(1), dim3(64), 0, c->stream, d_cand + (size_t)((pos0 + n - 1) & 1) * TK_NCU,
                           h_ids_dev + (n - 1), c->d_next, reinterpret_cast<unsigned*>(c->d_logits + c->V), c->V);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(c->h_next + 1, c->d_logits + c->V, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        // drain: ids are 1-based, 0 = not resolved yet
        volatile int* ids = h_ids;
        while (done < n) {
            if (ids[done] != 0) {
                ids_out[done] = ids[done];
                if (on_token) on_token(done, ids_out[done], user);
                ++done;
                continue;
            }
            const hipError_t q = hipStreamQuery(c->stream);
            if (q == hipSuccess) { if (ids[done] == 0) break; continue; }   // stream drained without the id: an error below
            if (q != hipErrorNotReady) return LLMK_E_HIP + (int)q;
        }
        HIPCHK(hipStreamSynchronize(c->stream));
        for (; done < n && ids[done] != 0; ++done) {
            ids_out[done] = ids[done];
            if (on_token) on_token(done, ids_out[done], user);
        }
        const unsigned err = (unsigned)c->h_next[1];
        if (err == 0 && done == n) return LLMK_OK;
        // a timed-out exchange: every later launch drained on the sticky word.  Retire the token kernel and redo the rest,
        // from the first position whose id never arrived, on the multi-kernel path (it rewrites those KV rows).
        // (0x4000 alone -- an activation beyond the q4_0 kernel's f16 image at ONE position: the rest of the call goes position by
        // position through run_token, which redoes just the positions that need it and keeps the kernel: see there)
        rc = tk_range_only(err) ? tk_clear_err(c) : tk_retire(c, err, pos0 + done);
        if (rc) return rc;
        if (done > 0) token = ids_out[done - 1];
        c->tk_short_grid = false;
    }
    for (int i = done; i < n; ++i) {
        rc = run_token(c, token, pos0 + i, true);
        if (rc) return rc;
        if (*c->h_next < 1 || *c->h_next > c->V) return LLMK_E_NONFINITE;      // (before the callback sees it)
        token = ids_out[i] = *c->h_next;
        if (on_token) on_token(i, token, user);
    }
    return LLMK_OK;
}

int llmk_reset(llmk_ctx* c) {
    if (!c) return LLMK_E_ARG;
    HIPCHK(hipSetDevice(c->cfg.device));
    const size_t kvn = (size_t)c->L * c->S * c->KVl * sizeof(float);
    HIPCHK(hipMemsetAsync(c->d_kc, 0, kvn, c->stream));
    HIPCHK(hipMemsetAsync(c->d_vc, 0, kvn, c->stream));
    HIPCHK(hipMemsetAsync(c->d_logits + c->V, 0, sizeof(float), c->stream));   // the token kernel's sticky error word
    if (c->d_gran && c->tk_ngran)      // a new sequence: no position before it (the q4_0 kernels' scale records, token_kernel.h tk_qsc)
        HIPCHK(hipMemsetAsync(c->d_gran + c->tk_ngran - 4 * TK_QSC_LMAX, 0, 4 * TK_QSC_LMAX * sizeof(unsigned long long), c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    reinterpret_cast<unsigned*>(c->h_logits)[c->V] = 0;
    c->h_next[1] = 0;
    for (int i = 0; i < 5; ++i) c->times[i] = 0.f;
    return LLMK_OK;
}

int llmk_timings(llmk_ctx* c, float ms[5]) {
    if (!c || !ms) return LLMK_E_ARG;
    for (int i = 0; i < 5; ++i) ms[i] = c->times[i];
    return LLMK_OK;
}

int llmk_time_kernel(llmk_ctx* c, int kernel, int iters, float* avg_ms, double* bytes_per_launch) {
    int rc = check_ready(c);
    if (rc) return rc;
    if (kernel < 0 || kernel > 11 || iters <= 0 || !avg_ms) return LLMK_E_ARG;
    if (kernel == 6 && !c->use_tk) return LLMK_E_ARG;
    HIPCHK(hipSetDevice(c->cfg.device));
    if (kernel == 11) {
        // the five per-layer kernels of the multi-kernel path (a tensor-parallel rank's too: without its exchanges) for all L
        // layers as ONE hipGraph, replayed `iters` times: microseconds per LAYER as the token pass pays them -- a dependent
        // kernel boundary inside a graph is ~1 us cheaper than between eager launches, which is what kernels 0..4 time
        if (c->h_tokpos[1] < 1) { c->h_tokpos[0] = 0; c->h_tokpos[1] = 1; }
        HIPCHK(hipMemcpyAsync(c->d_tokpos, c->h_tokpos, 4 * sizeof(int), hipMemcpyHostToDevice, c->stream));
        hipGraph_t g = nullptr;
        hipGraphExec_t ge = nullptr;
        HIPCHK(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
        hipError_t e = hipSuccess;
        for (int l = 0; l < c->L && e == hipSuccess; ++l) {
            e = launch_qkv(c, l);
            if (e == hipSuccess) e = launch_attn(c, l);
            if (e == hipSuccess) e = launch_wo(c, l);
            if (e == hipSuccess) e = launch_w13(c, l);
            if (e == hipSuccess) e = launch_w2(c, l);
        }
        const hipError_t e2 = hipStreamEndCapture(c->stream, &g);
        if (e != hipSuccess || e2 != hipSuccess) { if (g) hipGraphDestroy(g); return LLMK_E_HIP + (int)(e != hipSuccess ? e : e2); }
        e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipGraphDestroy(g);
        HIPCHK(e);
        for (int i = 0; i < 3 && e == hipSuccess; ++i) e = hipGraphLaunch(ge, c->stream);
        if (e == hipSuccess) e = hipEventRecord(c->ev[6], c->stream);
        for (int i = 0; i < iters && e == hipSuccess; ++i) e = hipGraphLaunch(ge, c->stream);
        if (e == hipSuccess) e = hipEventRecord(c->ev[7], c->stream);
        if (e == hipSuccess) e = hipEventSynchronize(c->ev[7]);
        float ms = 0.f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, c->ev[6], c->ev[7]);
        hipGraphExecDestroy(ge);
        HIPCHK(e);
        *avg_ms = ms / (float)iters / (float)c->L;
        if (bytes_per_launch) {
            double b = 0;
            for (int m : {LLMK_WQKV, LLMK_WO, LLMK_W13, LLMK_W2}) b += (double)c->desc[m].rows * (double)row_bytes_for(c->t[m].type, c->desc[m].K);
            *bytes_per_launch = b;
        }
        return LLMK_OK;
    }
    if (kernel >= 7) {   // the prefill GEMMs at PF_TMAX positions: 7 w1|w3, 8 wqkv, 9 wo, 10 w2 (whatever the workspaces hold: timing only)
        const int pf_step = PF_KSTEP;
        if (c->tp_size != 1 || c->E % pf_step || c->H % pf_step) return LLMK_E_ARG;
        rc = pf_setup(c, true);
        if (rc) return rc;
        // activations of ordinary size (every float 0x3c3c3c3c = 0.0115): an all-zero workspace would make every workgroup of the f16-instruction
        // GEMM cast its low-end votes (prefill.h pf_low_check: 32k atomics per launch, +16 us) -- a path real activations do not take
        HIPCHK(hipMemsetAsync(c->pf[0].Xs, 0x3c, (size_t)PF_TMAX * c->E * sizeof(float), c->stream));
        HIPCHK(hipMemsetAsync(c->pf[0].HB, 0x3c, (size_t)PF_TMAX * c->H * sizeof(float), c->stream));
    }
    // Successive launches walk the layers so the weight stream never re-hits the 256 MiB Infinity
    // Cache (the classifier has one matrix: its figure is cache-assisted beyond the first launch).
    // NOTE: this overwrites x / caches at h_tokpos' position; call llmk_reset afterwards.
    if (c->h_tokpos[1] < 1) { c->h_tokpos[0] = 0; c->h_tokpos[1] = 1; }
    HIPCHK(hipMemcpyAsync(c->d_tokpos, c->h_tokpos, 4 * sizeof(int), hipMemcpyHostToDevice, c->stream));
    auto one = [&](int i) -> hipError_t {
        int l = i % c->L;
#ifdef LLMK_PF_TRACE
        if (getenv("LLMK_PF_ONE_LAYER")) l = 0;      // debug build: weights from the Infinity Cache instead of HBM
#endif
        if (kernel == 6) {  // whole-token kernel: fresh exchange epochs for every launch
            hipLaunchKernelGGL(bump_serial_kernel, dim3(1), dim3(1), 0, c->stream, c->d_tokpos);
            return launch_token_kernel(c);
        }
        if (kernel >= 7) {
            PfEpiArgs e;
            const int tid = kernel == 7 ? LLMK_W13 : kernel == 8 ? LLMK_WQKV : kernel == 9 ? LLMK_WO : LLMK_W2;
            const int rows = kernel == 7 ? 2 * c->H : kernel == 8 ? c->E + 2 * c->KV : c->E;
            const char* w = (const char*)c->t[tid].data + (size_t)l * rows * c->t[tid].row_bytes;
            const float* X = kernel == 10 ? c->pf[0].HB : c->pf[0].Xs;
            const int K = kernel == 10 ? c->H : c->E;
            return pf_gemm(c, c->pf[0], w, (int)c->t[tid].row_bytes, X, rows, K, PF_TMAX, &e);
        }
        switch (kernel) {
            case 0: return launch_qkv(c, l);
            case 1: return launch_attn(c, l);
            case 2: return launch_wo(c, l);
            case 3: return launch_w13(c, l);
            case 4: return launch_w2(c, l);
            default: return launch_cls(c);
        }
    };
    for (int i = 0; i < 3; ++i) HIPCHK(one(i));
    HIPCHK(hipEventRecord(c->ev[6], c->stream));
    for (int i = 0; i < iters; ++i) HIPCHK(one(i + 3));
    HIPCHK(hipEventRecord(c->ev[7], c->stream));
    HIPCHK(hipEventSynchronize(c->ev[7]));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, c->ev[6], c->ev[7]));
    *avg_ms = ms / (float)iters;
    if (kernel == 6) c->h_tokpos[2] += iters + 3;   // the device-side serial was bumped once per launch: keep the host's in step
                                                     // (exchange epochs must stay unique per serial)
    if (bytes_per_launch) {
        const int tids[11] = {LLMK_WQKV, -1, LLMK_WO, LLMK_W13, LLMK_W2, LLMK_WCLS, -1, LLMK_W13, LLMK_WQKV, LLMK_WO, LLMK_W2};
        double b = 0;
        if (kernel == 6) {
            double w = 0;
            const int mats[5] = {LLMK_WQKV, LLMK_WO, LLMK_W13, LLMK_W2, LLMK_WCLS};
            for (int m : mats) w += (double)c->desc[m].rows * (c->desc[m].layered ? c->L : 1) * (double)row_bytes_for(c->t[m].type, c->desc[m].K);
            b = w + (2.0 * c->L + 1) * c->E * 4 + c->E * 4 + 2.0 * c->L * c->KV * 4.0 * c->h_tokpos[1] +
                2.0 * c->L * c->KV * 4 + c->V * 4.0;
        } else if (kernel == 1) {
            b = 2.0 * c->KV * 4.0 * c->h_tokpos[1];  // K and V rows 1..pos of one layer
        } else {
            const TensorDesc& d = c->desc[tids[kernel]];
            b = (double)d.rows * (double)row_bytes_for(c->t[tids[kernel]].type, d.K);
        }
        *bytes_per_launch = b;
    }
    return LLMK_OK;
}

int llmk_peek(llmk_ctx* c, int which, int layer, int pos, float* out, int n) {
    if (!c || !out || n <= 0) return LLMK_E_ARG;
    HIPCHK(hipSetDevice(c->cfg.device));
    const float* src = nullptr;
    int len = 0;
    switch (which) {
        case 0: src = c->d_x; len = c->E; break;
        case 1: src = c->d_q; len = c->Eq; break;
        case 2: src = c->d_xb; len = c->Eq; break;
        case 3: src = c->d_hb; len = c->Hl; break;
        case 4:
        case 5:
            if (layer < 0 || layer >= c->L || pos < 1 || pos > c->S) return LLMK_E_ARG;
            src = (which == 4 ? c->d_kc : c->d_vc) + ((size_t)layer * c->S + (pos - 1)) * c->KVl;
            len = c->KVl;
            break;
#ifdef LLMK_PF_TRACE
        case 7: src = c->pf[0].HB; len = PF_TMAX * c->H; break;
#endif
        case 8:  // the q4_0 persistent kernels' per-layer scale records (token_kernel.h tk_qsc): [2 buffers][TK_QSC_LMAX] x {|xb|, |hb|, pos, pos}
            if (!c->d_gran || !c->tk_ngran) return LLMK_E_ARG;
            src = reinterpret_cast<const float*>(c->d_gran + c->tk_ngran - 4 * TK_QSC_LMAX); len = 8 * TK_QSC_LMAX;
            break;
        case 6:  // debug: raw trace stamps reinterpret as floats (2 per stamp)
            if (!c->d_trace) return LLMK_E_ARG;
            src = (const float*)c->d_trace; len = TK_NCU * TK_TRACE_N * 2;
            break;
        default: return LLMK_E_ARG;
    }
    if (n > len) return LLMK_E_ARG;
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(out, src, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    return LLMK_OK;
}

int llmk_tp_unique_id(char id_out[128]) {
    if (!id_out) return LLMK_E_ARG;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return LLMK_E_COMM;
    memcpy(id_out, &id, sizeof(id));
    return LLMK_OK;
}

int llmk_tp_init_comm(llmk_ctx* c, const char id_in[128]) {
    if (!c || !id_in || c->comm) return LLMK_E_ARG;
    if (c->p2p) return LLMK_E_STATE;      // llmk_tp_p2p_disable first: a ctx runs ONE kind of collective
    HIPCHK(hipSetDevice(c->cfg.device));
    ncclUniqueId id;
    memcpy(&id, id_in, sizeof(id));
    if (ncclCommInitRank(&c->comm, c->tp_size, id, c->tp_rank) != ncclSuccess) { c->comm = nullptr; return LLMK_E_COMM; }
    return LLMK_OK;
}

// ---- one-shot peer-memory collectives (tp_p2p.h) ----------------------------------------------------------------------
static int tp_inbox_alloc(llmk_ctx* c) {
    if (c->d_inbox) return LLMK_OK;
    if (c->tp_size < 2 || c->tp_size > TP_MAX_RANKS) return LLMK_E_ARG;
    HIPCHK(hipSetDevice(c->cfg.device));
    const size_t bytes = tp_inbox_granules(c->tp_size, c->E, c->V) * sizeof(unsigned long long);
    // fine-grained: peers' stores over xGMI and this device's system-scope polls meet in memory, not in a stale L2 line
    HIPCHK(hipExtMallocWithFlags((void**)&c->d_inbox, bytes, hipDeviceMallocFinegrained));
    HIPCHK(hipMemset(c->d_inbox, 0, bytes));      // tag 0 is never a valid epoch
    HIPCHK(dev_alloc(&c->d_tp_bad, sizeof(unsigned)));
    HIPCHK(hipDeviceSynchronize());
    c->peers.inbox[c->tp_rank] = c->d_inbox;
    // bound of every spin of the exchange kernels in WALL-CLOCK ticks (tp_p2p.h): 20 s unless LLMK_TP_TIMEOUT_MS says otherwise
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->cfg.device) != hipSuccess || khz <= 0) khz = 100000;   // 100 MHz on gfx9
    const long long ms = getenv("LLMK_TP_TIMEOUT_MS") ? atoll(getenv("LLMK_TP_TIMEOUT_MS")) : 20000;
    c->peers.timeout_ticks = (unsigned long long)(ms > 0 ? ms : 20000) * (unsigned long long)khz;
    return LLMK_OK;
}

int llmk_tp_p2p_handle(llmk_ctx* c, char handle_out[64]) {
    if (!c || !handle_out) return LLMK_E_ARG;
    int rc = tp_inbox_alloc(c);
    if (rc) return rc;
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
    hipIpcMemHandle_t h;
    HIPCHK(hipIpcGetMemHandle(&h, c->d_inbox));
    memcpy(handle_out, &h, sizeof(h));
    return LLMK_OK;
}

int llmk_tp_p2p_connect(llmk_ctx* c, const char* handles) {
    if (!c || !handles || c->p2p) return LLMK_E_ARG;
    int rc = tp_inbox_alloc(c);
    if (rc) return rc;
    for (int r = 0; r < c->tp_size; ++r) {
        if (r == c->tp_rank) continue;
        hipIpcMemHandle_t h;
        memcpy(&h, handles + (size_t)r * 64, sizeof(h));
        void* p = nullptr;
        HIPCHK(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
        c->ipc_mapped[r] = p;
        c->peers.inbox[r] = (unsigned long long*)p;
    }
    c->p2p = true;
    return LLMK_OK;
}

int llmk_tp_p2p_connect_local(llmk_ctx* c, llmk_ctx* const* ranks) {
    if (!c || !ranks || c->p2p) return LLMK_E_ARG;
    int rc = tp_inbox_alloc(c);
    if (rc) return rc;
    for (int r = 0; r < c->tp_size; ++r) {
        if (r == c->tp_rank) continue;
        llmk_ctx* o = ranks[r];
        if (!o || o->tp_size != c->tp_size || o->tp_rank != r || o->E != c->E || o->V != c->V) return LLMK_E_ARG;
        rc = tp_inbox_alloc(o);
        if (rc) return rc;
        HIPCHK(hipSetDevice(c->cfg.device));
        if (o->cfg.device != c->cfg.device) {
            const hipError_t e = hipDeviceEnablePeerAccess(o->cfg.device, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return LLMK_E_HIP + (int)e;
            (void)hipGetLastError();
        }
        c->peers.inbox[r] = o->d_inbox;
    }
    c->p2p = true;
    return LLMK_OK;
}

// ---- self-test of the peer-memory collectives ----------------------------------------------------------------------------
// What differs between one GPU and eight is exactly what a 1-GPU box cannot show: peers' system-scope stores landing in
// this rank's fine-grained inbox over xGMI, lazy peer access through the IPC mapping, polls of memory a remote device
// writes.  So every rank proves it on the hardware it runs on BEFORE the first token: `iters` rounds of both all-reduce
// halves and the all-gather on known integers (exact in f32), with the bounded spins of the real kernels.  The host
// collects the ranks' verdicts over its side channel and keeps the peer-memory path only if ALL passed; otherwise every
// rank calls llmk_tp_p2p_disable and the token pass runs over RCCL (llmk_tp_init_comm).
__global__ void tp_selftest_fill_kernel(float* part, float* x, float* logits, int E, int V, int me, int P, int it) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < E) { part[i] = (float)((i * 7 + me * 13 + it) % 1021); x[i] = 0.f; }
    if (i < V) logits[i] = (i / (V / P) == me) ? (float)((i * 3 + it) % 4093) : -1.f;
}
__global__ void tp_selftest_check_kernel(const float* x, const float* logits, int E, int V, int P, int it, unsigned* bad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < E) {
        float want = 0.f;
        for (int r = 0; r < P; ++r) want += (float)((i * 7 + r * 13 + it) % 1021);
        if (x[i] != 2.f * want) atomicAdd(bad, 1u);      // two all-reduces were added into x
    }
    if (i < V && logits[i] != (float)((i * 3 + it) % 4093)) atomicAdd(bad, 1u);
}
static int tp_selftest_run(llmk_ctx* c, int iters, unsigned jseed) {
    if (!c || iters < 1) return LLMK_E_ARG;
    if (!c->p2p) return LLMK_E_COMM;
    HIPCHK(hipSetDevice(c->cfg.device));
    // (no allocation and no hipFree in here: with two ranks in one process a peer's kernel may already be spinning for this rank, and
    // an allocation's fill -- LLMK_POISON -- or a free's device-wide wait queues behind it: round 6's poison runs, 2 x 20 s of timeouts)
    unsigned* d_bad = c->d_tp_bad;
    if (!d_bad) return LLMK_E_COMM;
    HIPCHK(hipMemsetAsync(d_bad, 0, sizeof(unsigned), c->stream));
    const int n = std::max(c->E, c->V);
    int rc = LLMK_OK;
    // The serial that makes the epochs unique is bumped ON THE DEVICE between rounds (the host's copy is brought in step at
    // the end): an async copy out of the pinned h_tokpos per round would read whatever the host loop, running ahead, had
    // already written there -- ranks would disagree on the epochs (seen: 4 rounds pass, 64 time out).
    hipError_t e0 = hipMemcpyAsync(c->d_tokpos, c->h_tokpos, 4 * sizeof(int), hipMemcpyHostToDevice, c->stream);
    if (e0 == hipSuccess) e0 = hipStreamSynchronize(c->stream);
    if (e0 != hipSuccess) rc = LLMK_E_HIP + (int)e0;
    for (int it = 0; it < iters && rc == LLMK_OK; ++it) {
        const unsigned js = jseed ? (jseed + (unsigned)it * 2654435761u) | 1u : 0u;
        hipLaunchKernelGGL(bump_serial_kernel, dim3(1), dim3(1), 0, c->stream, c->d_tokpos);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) {
            hipLaunchKernelGGL(tp_selftest_fill_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->d_part, c->d_x, c->d_logits, c->E,
                               c->V, c->tp_rank, c->tp_size, it);
            e = launch_tp_allreduce_add(c, 0, js);
        }
        if (e == hipSuccess) e = launch_tp_allreduce_add(c, 1, js);
        if (e == hipSuccess) e = launch_tp_allgather(c, js);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(tp_selftest_check_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->d_x, c->d_logits, c->E, c->V,
                               c->tp_size, it, d_bad);
            e = hipGetLastError();
        }
        if (e != hipSuccess) rc = LLMK_E_HIP + (int)e;
        // the jittered rounds are milliseconds each: do not queue thousands of spinning launches ahead of the device
        if (jseed && (it & 63) == 63 && rc == LLMK_OK) { e = hipStreamSynchronize(c->stream); if (e != hipSuccess) rc = LLMK_E_HIP + (int)e; }
    }
    c->h_tokpos[2] += iters;                     // every rank ran the same number of rounds: the serials stay in step
    unsigned bad = 0, err = 0;
    if (rc == LLMK_OK) {
        hipError_t e = hipStreamSynchronize(c->stream);
        if (e == hipSuccess) e = hipMemcpy(&bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(&err, c->d_logits + c->V, sizeof(err), hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = LLMK_E_HIP + (int)e;
        else if (err) rc = LLMK_E_TIMEOUT;
        else if (bad) rc = LLMK_E_COMM;
    }
    // leave the ctx as a fresh one: x, the logits and the sticky word
    hipMemsetAsync(c->d_x, 0, (size_t)c->E * sizeof(float), c->stream);
    hipMemsetAsync(c->d_logits, 0, ((size_t)c->V + 4) * sizeof(float), c->stream);
    hipStreamSynchronize(c->stream);
    if (rc != LLMK_OK) {
        char what[160];
        fprintf(stderr, "llmk: rank %d of %d: the peer-memory collectives failed their %s (%s, %u mismatches, %s)\n",
                c->tp_rank, c->tp_size, jseed ? "stress run" : "self-test",
                rc == LLMK_E_TIMEOUT ? "a peer's granules never arrived" : rc == LLMK_E_COMM ? "wrong sums" : "HIP error", bad,
                tp_describe_err(c, err, what, sizeof(what)));
    }
    return rc;
}
int llmk_tp_p2p_selftest(llmk_ctx* c, int iters) { return tp_selftest_run(c, iters, 0u); }
// The same rounds with wave-uniform pseudo-random delays before and between the sends and the reads of every exchange
// (tp_p2p.h JIT): ranks and waves drift apart by up to a whole exchange.  Verification only (tests/test_tp_gpu.py).
int llmk_tp_p2p_stress(llmk_ctx* c, int iters, unsigned seed) { return tp_selftest_run(c, iters, seed | 1u); }

// Stop using the peer-memory collectives on this ctx (after a failed self-test on ANY rank): the token pass then needs
// the RCCL communicator (llmk_tp_init_comm).  The inbox stays mapped until llmk_destroy.
int llmk_tp_p2p_disable(llmk_ctx* c) {
    if (!c) return LLMK_E_ARG;
    HIPCHK(hipSetDevice(c->cfg.device));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->p2p = false;
    if (c->graph_logits) { hipGraphExecDestroy(c->graph_logits); c->graph_logits = nullptr; }
    if (c->graph_greedy) { hipGraphExecDestroy(c->graph_greedy); c->graph_greedy = nullptr; }
    return LLMK_OK;
}

int llmk_tp_begin(llmk_ctx* c, int token, int pos) {
    int rc = check_ready(c);
    if (rc) return rc;
    if (token < 1 || token > c->V || pos < 1 || pos > c->S) return LLMK_E_ARG;
    HIPCHK(hipSetDevice(c->cfg.device));
    c->h_tokpos[0] = token - 1;
    c->h_tokpos[1] = pos;
    c->h_tokpos[2] += 1;
    HIPCHK(hipMemcpyAsync(c->d_tokpos, c->h_tokpos, 4 * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIPCHK(launch_embed(c));
    HIPCHK(hipStreamSynchronize(c->stream));
    return LLMK_OK;
}

int llmk_tp_segment(llmk_ctx* c, int seg, int layer) {
    int rc = check_ready(c);
    if (rc) return rc;
    if (seg < 0 || seg > 2 || layer < 0 || layer >= c->L) return LLMK_E_ARG;
    HIPCHK(hipSetDevice(c->cfg.device));
    if (seg == 0) {
        if (layer > 0) HIPCHK(launch_add_partial(c));
        HIPCHK(launch_qkv(c, layer));
        HIPCHK(launch_attn(c, layer));
        HIPCHK(launch_wo(c, layer));
    } else if (seg == 1) {
        HIPCHK(launch_add_partial(c));
        HIPCHK(launch_w13(c, layer));
        HIPCHK(launch_w2(c, layer));
    } else {
        HIPCHK(launch_add_partial(c));
        HIPCHK(launch_cls(c));
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    return LLMK_OK;
}

int llmk_tp_read_partial(llmk_ctx* c, float* out) {
    if (!c || !out) return LLMK_E_ARG;
    HIPCHK(hipSetDevice(c->cfg.device));
    HIPCHK(hipMemcpy(out, c->d_part, (size_t)c->E * sizeof(float), hipMemcpyDeviceToHost));
    return LLMK_OK;
}
int llmk_tp_write_partial(llmk_ctx* c, const float* in) {
    if (!c || !in) return LLMK_E_ARG;
    HIPCHK(hipSetDevice(c->cfg.device));
    HIPCHK(hipMemcpy(c->d_part, in, (size_t)c->E * sizeof(float), hipMemcpyHostToDevice));
    return LLMK_OK;
}
int llmk_tp_read_logits(llmk_ctx* c, float* out_slice) {
    if (!c || !out_slice) return LLMK_E_ARG;
    HIPCHK(hipSetDevice(c->cfg.device));
    HIPCHK(hipMemcpy(out_slice, c->d_logits + (size_t)c->tp_rank * c->Vl, (size_t)c->Vl * sizeof(float), hipMemcpyDeviceToHost));
    return LLMK_OK;
}

// 64-bit sum of the 32-bit words of a tensor's DEVICE image (this rank's shard, in the device layout): equal images give
// equal sums, so two uploads of the same weights, two contexts, or one context before and after a run can be compared
// without reading gigabytes back (tests: upload determinism, "weights untouched by the token pass").
__global__ void checksum_kernel(const unsigned* __restrict__ w, size_t nwords, unsigned long long* out) {
    unsigned long long t = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += (size_t)gridDim.x * blockDim.x) t += w[i];
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, t);
}
int llmk_tensor_checksum(llmk_ctx* c, int tid, unsigned long long* out) {
    if (!c || !out || tid < 0 || tid >= LLMK_N_TENSORS) return LLMK_E_ARG;
    const TensorDesc& d = c->desc[tid];
    const DevTensor& t = c->t[tid];
    if (!t.data) return LLMK_E_STATE;
    HIPCHK(hipSetDevice(c->cfg.device));
    const size_t nwords = (size_t)d.rows * (d.layered ? c->L : 1) * t.row_bytes / 4;
    unsigned long long* d_out = nullptr;
    HIPCHK(dev_alloc(&d_out, sizeof(*d_out)));
    hipError_t e = hipMemsetAsync(d_out, 0, sizeof(*d_out), c->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(checksum_kernel, dim3(1024), dim3(256), 0, c->stream, (const unsigned*)t.data, nwords, d_out);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, sizeof(*d_out), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    hipFree(d_out);
    return e == hipSuccess ? LLMK_OK : LLMK_E_HIP + (int)e;
}

// "E,H,NH,NKV,V,type[+cls];..." of every shape the persistent kernel is instantiated for in THIS build (LLMK_TK_SHAPES)
int llmk_tk_shapes(char* buf, size_t n) {
    if (!buf || n == 0) return LLMK_E_ARG;
    static const char* tn[] = {"f32", "f16", "q4_0"};
    size_t o = 0;
    buf[0] = 0;
    for (int i = 0; i < TK_NSHAPES; ++i) {
        const TkEntry& e = tk_table[i];
        char one[96];
        const int k = snprintf(one, sizeof(one), "%s%d,%d,%d,%d,%d,%s%s", i ? ";" : "", e.E, e.H, e.NH, e.NKV, e.V, tn[e.WT], e.CLS != e.WT ? "+q6_K" : "");
        if (o + (size_t)k + 1 > n) return LLMK_E_SIZE;
        memcpy(buf + o, one, (size_t)k + 1);
        o += (size_t)k;
    }
    return LLMK_OK;
}

int llmk_path(llmk_ctx* c) {
    if (!c) return -LLMK_E_ARG;
    if (c->p2p) return LLMK_PATH_TP_P2P;
    if (c->comm) return LLMK_PATH_TP_RCCL;
    if (c->tp_size > 1) return LLMK_PATH_TP_UNCONNECTED;
    return c->use_tk ? LLMK_PATH_TOKEN_KERNEL : LLMK_PATH_MULTI_KERNEL;
}

int llmk_tp_ranks_seen(llmk_ctx* c) {
    if (!c) return -LLMK_E_ARG;
    if (c->comm) {
        int n = 0;
        if (ncclCommCount(c->comm, &n) != ncclSuccess) return -LLMK_E_COMM;
        return n;                       // what the RCCL communicator itself says
    }
    if (c->p2p) {                       // peers whose inboxes are mapped here (+ this rank)
        int n = 0;
        for (int r = 0; r < c->tp_size; ++r) n += c->peers.inbox[r] != nullptr;
        return n;
    }
    return 1;
}

int llmk_destroy(llmk_ctx* c) {
    if (!c) return LLMK_E_ARG;
    hipSetDevice(c->cfg.device);
    if (c->stream) hipStreamSynchronize(c->stream);
    if (c->comm) ncclCommDestroy(c->comm);
    for (int r = 0; r < TP_MAX_RANKS; ++r)
        if (c->ipc_mapped[r]) hipIpcCloseMemHandle(c->ipc_mapped[r]);
    if (c->d_inbox) hipFree(c->d_inbox);
    if (c->d_tp_bad) hipFree(c->d_tp_bad);
    if (c->graph_logits) hipGraphExecDestroy(c->graph_logits);
    if (c->graph_greedy) hipGraphExecDestroy(c->graph_greedy);
    for (int i = 0; i < LLMK_N_TENSORS; ++i) {
        if (c->t[i].data && !c->t[i].alias) hipFree(c->t[i].data);
    }
    pf_teardown(c);
    void* dev[] = {c->d_kc, c->d_vc, c->d_x, c->d_q, c->d_xb, c->d_hb, c->d_logits, c->d_rope, c->d_tokpos, c->d_next,
                   c->d_gran, c->d_zeros, c->d_trace, c->d_part};
    for (void* p : dev)
        if (p) hipFree(p);
    for (int b = 0; b < 2; ++b) {
        if (c->up_stage[b]) hipHostFree(c->up_stage[b]);
        if (c->up_tmp[b]) hipFree(c->up_tmp[b]);
        if (c->up_done[b]) hipEventDestroy(c->up_done[b]);
    }
    if (c->up_acc) hipFree(c->up_acc);
    for (int i = 0; i < LLMK_N_TENSORS; ++i)
        if (c->q16[i]) hipFree(c->q16[i]);
    if (c->h_tokpos) hipHostFree(c->h_tokpos);
    if (c->h_logits) hipHostFree(c->h_logits);
    if (c->h_next) hipHostFree(c->h_next);
    for (int i = 0; i < 8; ++i)
        if (c->ev[i]) hipEventDestroy(c->ev[i]);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
    return LLMK_OK;
}

}  // extern "C"
