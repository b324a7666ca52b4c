/*
 * llmk -- C-ABI of the MI355X-native (gfx950) decode hot path of rbitr/llm.f90.
 *
 * This is the drop-in boundary.  The reference has no FFI of its own: the
 * seam is the single Fortran call
 *
 *     logits = transformer(token,pos,s,weights)            /root/reference/llama2.f90:380
 *     function transformer(token, pos, s, w) result(logits) /root/reference/llama2.f90:480-485
 *
 * plus the weights it reads (type TransformerWeights, /root/reference/weight_module.f90:13-26)
 * and the per-sequence state it mutates (type RunState, weight_module.f90:33-40, allocated at
 * llama2.f90:311-319).  A Fortran host binds these entry points with ISO_C_BINDING
 * (llm.f90_amd/host/llmk_binding.f90; the stub a reference maintainer would add is in
 * INTEGRATION.md); tests/ and bench.py bind the same symbols with ctypes.
 *
 * Conventions
 *   - plain pointers and sizes only; every function returns 0 on success, nonzero on error
 *     (a LLMK_E_* code or, above 1000, 1000 + the hipError_t).  No exceptions, no exit().
 *     The reference's own convention is print + stop (read_ggml.f90:122-125); the host does that.
 *   - token and pos are 1-BASED exactly as at llama2.f90:380 (BOS is token 2, llama2.f90:376).
 *   - weights are COPIED to the device by llmk_upload; host arrays may be freed afterwards.
 *   - array layout is the reference's: Fortran (in, rows, layer) column-major == C
 *     [layer][row][in].  No transposition anywhere.
 *   - one ctx == one sequence (KV cache inside); calls on a ctx are serialised by the caller,
 *     like the reference's non-reentrant `transformer` (it mutates `s`).
 *   - there is NO CPU fallback: without a usable HIP device llmk_create fails.
 */
#ifndef LLMK_H
#define LLMK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ggml tensor types accepted for the matmul weights (GGUF tensor-info `type`,
 * read_ggml.f90:613-635 reads 0 and 1; 2 is the q4_0 of the four_bit_dev branch). */
#define LLMK_TYPE_F32 0
#define LLMK_TYPE_F16 1
#define LLMK_TYPE_Q4_0 2
#define LLMK_TYPE_Q6_K 14 /* ggml block_q6_K (210 bytes per 256 weights): LLMK_WCLS only, see llmk_set_tensor_type */

/* tensor ids for llmk_upload: one per component of TransformerWeights (weight_module.f90:13-26) */
#define LLMK_TOKEN_EMBEDDING_TABLE 0 /* (E,V)        C [V][E]         always f32            */
#define LLMK_RMS_ATT_WEIGHT 1        /* (E,L)        C [L][E]         f32                   */
#define LLMK_RMS_FFN_WEIGHT 2        /* (E,L)        C [L][E]         f32                   */
#define LLMK_WQKV 3                  /* (E,E+2KV,L)  C [L][E+2KV][E]  rows: Q | K | V       */
#define LLMK_WO 4                    /* (E,E,L)      C [L][E][E]                            */
#define LLMK_W13 5                   /* (E,2H,L)     C [L][2H][E]     rows: gate(w1) | up(w3) */
#define LLMK_W2 6                    /* (H,E,L)      C [L][E][H]                            */
#define LLMK_RMS_FINAL_WEIGHT 7      /* (E)                           f32                   */
#define LLMK_WCLS 8                  /* (E,V)        C [V][E]                               */
#define LLMK_N_TENSORS 9

#define LLMK_FLAG_NO_GRAPH 1 /* launch kernels eagerly instead of replaying a hipGraph          */
#define LLMK_FLAG_TIMINGS 2  /* record the reference's 5 section timers (implies NO_GRAPH)      */
#define LLMK_FLAG_MULTI_KERNEL 4 /* never use the persistent whole-token kernel (5 launches/layer) */

/* error codes */
#define LLMK_OK 0
#define LLMK_E_ARG 1       /* bad argument (null pointer, id out of range, token/pos out of range) */
#define LLMK_E_SHAPE 2     /* unsupported or inconsistent model shape                               */
#define LLMK_E_SIZE 3      /* nbytes does not match the tensor's size for its type                  */
#define LLMK_E_TYPE 4      /* unsupported ggml type                                                 */
#define LLMK_E_STATE 5     /* forward before all weights were uploaded                              */
#define LLMK_E_NODEVICE 6  /* no usable HIP device (there is no CPU fallback)                       */
#define LLMK_E_NOMEM 7
#define LLMK_E_TIMEOUT 8   /* an in-kernel exchange timed out (GPU shared with other work?)        */
#define LLMK_E_COMM 9      /* tensor-parallel ctx used before llmk_tp_init_comm, or an RCCL error       */
#define LLMK_E_VERIFY 10   /* an uploaded block's word sum on the device differed from the host's, three times over      */
#define LLMK_E_NONFINITE 11 /* llmk_forward_greedy / llmk_decode_greedy: no logit of the position is finite, there is no greedy token  */
#define LLMK_E_HIP 1000    /* 1000 + hipError_t                                                     */

/* Run-time replacement of the reference's compile-time dims (llama2.f90:102-108) and of
 * type Config (weight_module.f90:28-31). */
typedef struct llmk_config {
    int32_t emb_dim;     /* E  */
    int32_t hidden_dim;  /* H  */
    int32_t n_layers;    /* L  */
    int32_t n_heads;     /* nh */
    int32_t n_kv_heads;  /* nkv */
    int32_t vocab_size;  /* V  */
    int32_t seq_len;     /* S: KV-cache capacity (llama2.f90:311-313) */
    int32_t weight_type; /* LLMK_TYPE_* of wqkv/wo/w13/w2/wcls */
    int32_t device;      /* HIP device ordinal */
    int32_t flags;       /* LLMK_FLAG_* */
} llmk_config;

typedef struct llmk_ctx llmk_ctx;

/* Allocates device weights + RunState (key_cache, value_cache zeroed as at llama2.f90:316-318).
 * Replaces: the allocations at llama2.f90:311-319 and read_ggml.f90:265-410. */
int llmk_create(const llmk_config *cfg, llmk_ctx **out);

/* Tensor-parallel shard `tp_rank` of `tp_size` of the same model (for the 70B configuration,
 * where one GPU's HBM bandwidth is the limit).  Megatron split: wqkv, w13 and wcls by output rows (each rank
 * owns nkv/P kv heads and their query heads, H/P hidden rows, V/P vocabulary rows), wo and w2 by the
 * contraction dimension; per layer two all-reduces of an E-vector, one all-gather of the logits per token.
 * llmk_upload / llmk_upload_rows are handed the FULL tensors (whole layers) and keep only this rank's
 * shard.  Requires n_kv_heads, hidden_dim, vocab_size divisible by tp_size.  One process per GPU:
 * rank 0 calls llmk_tp_unique_id, ships the 128 bytes to the other ranks (any side channel), every rank
 * calls llmk_tp_init_comm; after that llmk_forward / llmk_forward_greedy run the collectives over RCCL. */
int llmk_create_tp(const llmk_config *cfg, int tp_rank, int tp_size, llmk_ctx **out);
int llmk_tp_unique_id(char id_out[128]);
int llmk_tp_init_comm(llmk_ctx *ctx, const char id[128]);

/* The same three collectives as ONE-SHOT exchanges over peer memory (xGMI is a full mesh: every rank writes its partial
 * E-vector straight into every peer's inbox and adds the P partials in rank order -- one hop instead of a ring's
 * 2(P-1); csrc/tp_p2p.h).  Each rank exports its inbox with llmk_tp_p2p_handle (64 bytes, a hipIpcMemHandle_t), the host
 * ships the P handles to every rank (any side channel: files, MPI, torch.distributed) and each rank calls
 * llmk_tp_p2p_connect with all of them, `handles` = tp_size * 64 bytes in rank order (its own entry is ignored).  Ranks
 * that live in ONE process (one host thread per GPU) connect with llmk_tp_p2p_connect_local instead: `ranks[r]` = rank r's
 * ctx.  After either call llmk_forward / llmk_forward_greedy run the token pass with these collectives (the RCCL
 * communicator, if any, is then unused).  All ranks must call llmk_forward for the same token and position. */
int llmk_tp_p2p_handle(llmk_ctx *ctx, char handle_out[64]);
int llmk_tp_p2p_connect(llmk_ctx *ctx, const char *handles);
int llmk_tp_p2p_connect_local(llmk_ctx *ctx, llmk_ctx *const *ranks);
/* Prove the peer-memory path on the hardware it runs on before the first token (all ranks call it together, after the
 * connect): `iters` rounds of both all-reduce halves and the all-gather on known integers, with the real kernels and
 * their bounded spins.  0 = this rank saw only correct sums; LLMK_E_TIMEOUT / LLMK_E_COMM / a HIP code otherwise.  The
 * host gathers the ranks' verdicts over its side channel; unless ALL are 0, every rank calls llmk_tp_p2p_disable and
 * llmk_tp_init_comm, and the token pass runs over RCCL (what `llm --ngpu` and `bench.py --tp` do). */
int llmk_tp_p2p_selftest(llmk_ctx *ctx, int iters);
/* Verification only: the self-test's rounds with pseudo-random delays (seeded by `seed`) injected before and between the
 * sends and the reads of every exchange, so that ranks and wavefronts drift apart by up to a whole exchange -- what a slower
 * link or a time-sliced GPU does to them.  Same verdicts as the self-test; all ranks call it together. */
int llmk_tp_p2p_stress(llmk_ctx *ctx, int iters, unsigned seed);
int llmk_tp_p2p_disable(llmk_ctx *ctx);

/* Single-process stepping of a tensor-parallel ctx, for verification on one GPU (no communicator): the
 * caller plays the collective.  llmk_tp_begin sets token/pos; llmk_tp_segment runs
 *   seg 0 (layer l):  [l>0: x += exchanged]  rmsnorm+qkv, attention, wo   -> partial E-vector
 *   seg 1 (layer l):  x += exchanged          rmsnorm+w1|w3, w2            -> partial E-vector
 *   seg 2:            x += exchanged          final rmsnorm + classifier   -> this rank's V/P logits
 * llmk_tp_read_partial / llmk_tp_write_partial move the E-vector (the caller sums the ranks' partials in
 * rank order and writes the sum back to every rank); llmk_tp_read_logits returns the rank's logits slice. */
int llmk_tp_begin(llmk_ctx *ctx, int token, int pos);
int llmk_tp_segment(llmk_ctx *ctx, int seg, int layer);
int llmk_tp_read_partial(llmk_ctx *ctx, float *out);
int llmk_tp_write_partial(llmk_ctx *ctx, const float *in);
int llmk_tp_read_logits(llmk_ctx *ctx, float *out_slice);

/* Copy one whole TransformerWeights component to the device.  `host` points at the first
 * element of the Fortran array (c_loc(w%wqkv) ...); nbytes must equal the full array size for
 * `ggml_type` (f32: 4 B/weight, f16: 2 B/weight, q4_0: 18 B per 32 weights along `in`).
 * Norm gains and the embedding table must be f32.  Replaces nothing in the reference (it has no
 * device); called once after load_ggml returns (llama2.f90:151). */
int llmk_upload(llmk_ctx *ctx, int tensor_id, const void *host, size_t nbytes, int ggml_type);

/* Same, for `rows` consecutive rows of layer `layer` starting at `row_offset` (lets a loader
 * stream one GGUF tensor at a time, e.g. attn_k into rows E..E+KV-1 of wqkv, read_ggml.f90:286,
 * without materialising the fused array on the host).  Rows and offsets are those of the FULL tensor on a
 * tensor-parallel ctx too: the shim keeps the part of the range its shard holds (possibly nothing), so a rank may hand
 * over whole layers or only its own rows (host/gguf_loader.f90 stream_ggml_matrices reads nothing else from the file). */
int llmk_upload_rows(llmk_ctx *ctx, int tensor_id, int layer, int row_offset, int rows, const void *host,
                     size_t nbytes, int ggml_type);

/* Extensions beyond the reference's behaviour, both OPT-IN (the defaults reproduce llama2.f90):
 *  - llmk_set_tensor_type: give LLMK_WCLS its own ggml type (f32 / f16 / q4_0 / q6_K) before it is uploaded -- stock llama.cpp
 *    q4_0 files keep output.weight in q6_K (the reference stops on it, read_ggml.f90:633-635, :682-684).  Round 6: the raw q6_K
 *    super-blocks are uploaded as they lie in the file (LLMK_TYPE_Q6_K, emb_dim a multiple of 256) and dotted on the device; a
 *    ctx with q4_0 matrices of a shape the persistent kernel serves STAYS on it (llmk_path == 1).  A host may still hand the
 *    classifier over dequantised (LLMK_TYPE_F32): that ctx runs the multi-kernel path, as before;
 *  - llmk_set_rms_eps: rmsnorm epsilon other than the reference's hard-coded 1e-5 (llama2.f90:454), for a host that
 *    honours llama.attention.layer_norm_rms_epsilon. */
int llmk_set_tensor_type(llmk_ctx *ctx, int tensor_id, int ggml_type);
int llmk_set_rms_eps(llmk_ctx *ctx, float eps);

/* RoPE frequency table, n = head_size/2 floats: freqs[j] = 1/10000**((2j+1)/head_size), computed
 * by the HOST with the reference's own expression (llama2.f90:544-545) so the device angle
 * pos*freq starts from bit-identical frequencies.  Optional: llmk_create installs the same
 * table computed with powf. */
int llmk_set_rope_freqs(llmk_ctx *ctx, const float *freqs, int n);

/* The hot path: one token through the whole stack.  Replaces
 * `logits = transformer(token,pos,s,weights)` (llama2.f90:380).  token in [1,V], pos in [1,S],
 * both 1-based; positions must be fed in order 1,2,3,... (each call appends to the KV cache at
 * pos, llama2.f90:564-565).  logits_out: V floats, caller-owned host memory. */
int llmk_forward(llmk_ctx *ctx, int token, int pos, float *logits_out);

/* The prompt loop of llama2.f90:376-402 as ONE call: tokens[0..n) (1-based ids) sit at
 * positions pos0 .. pos0+n-1 (1-based); their KV-cache rows are written and logits_out[vocab_size] receives the
 * logits of the LAST position -- what n llmk_forward calls would leave behind, within the 1e-4 parity bar (only the
 * order of the dot-product partial sums differs).  single-GPU contexts (f32, f16, q4_0) batch up to 128 positions per pass
 * through MFMA GEMMs (a layer's weights cross HBM once per batch); tensor-parallel contexts, and shapes whose emb_dim /
 * hidden_dim is not a multiple of the GEMM's 64-column step, run the token-by-token pass inside.
 * The GEMMs multiply on the f16 matrix instruction with each f32 activation (and each f32 / q4_0 weight) as two f16 pieces
 * (exact products; |error| <= 2^-20 of an operand of magnitude >= 2^-3, 2^-24 absolute below: csrc/prefill.h); a prompt with an activation of magnitude >= 65504, or with a position whose whole row is below 2^-7, is redone on the
 * f32 instruction inside the same call, and the context stays there.  LLMK_PF_F32_MFMA=1 in the environment selects the
 * f32 instruction from the start. */
int llmk_prefill(llmk_ctx* ctx, const int* tokens, int n, int pos0, float* logits_out);

/* Same pass, but the temperature-0 consumer (`token = maxloc(logits,DIM=1)`, llama2.f90:388) runs
 * on the device: returns the 1-based argmax (first maximum wins) and skips the logits copy.
 * SURVEY.md section 8(f) rank 1. */
int llmk_forward_greedy(llmk_ctx *ctx, int token, int pos, int *next_token);

/* The temperature-0 generation loop of llama2.f90:379-396 for n positions in ONE call, without a host round trip per
 * token: position pos0 is fed `token`, every later position the device argmax (first maximum wins, llama2.f90:388) of the
 * position before it.  ids_out[i] = the 1-based token chosen after position pos0+i (= what llmk_forward_greedy returns
 * there).  On the persistent-kernel path the n launches are enqueued back to back: each launch leaves the per-CU maxima
 * of its classifier rows in device memory and the next one folds them at its start, so the token never leaves the
 * device; ids reach the host through mapped memory as they are resolved, and on_token (optional) is called from the
 * calling thread, in order, as each arrives -- the host can stream the text exactly as llama2.f90:396 does.  Other
 * contexts run the same positions through llmk_forward_greedy.  The logits of every position are still computed and
 * written (device memory); only their trip to the host is dropped.  SURVEY.md section 8(f) rank 1. */
typedef void (*llmk_token_fn)(int index, int token, void *user);
int llmk_decode_greedy(llmk_ctx *ctx, int token, int pos0, int n, int *ids_out, llmk_token_fn on_token, void *user);

/* Zero the KV cache (new sequence), as llama2.f90:316-318. */
int llmk_reset(llmk_ctx *ctx);

/* The reference's five section timers s%times(1:5) (llama2.f90:538,561,599,622,638), accumulated
 * milliseconds since create/reset; all zero unless LLMK_FLAG_TIMINGS. */
int llmk_timings(llmk_ctx *ctx, float ms[5]);

/* Measurement hook for bench.py: runs `iters` launches of one kernel of the token pass on the
 * ctx's stream with HIP events around them and returns the average milliseconds per launch and
 * the algorithmic bytes one launch moves.  kernel: 0 qkv, 1 attention, 2 wo, 3 w13, 4 w2,
 * 5 classifier (successive launches walk the layers), 6 the persistent whole-token kernel
 * (LLMK_E_ARG when the ctx runs the multi-kernel path), 7..10 the w1|w3, wqkv, wo, w2 GEMMs of llmk_prefill at 128 positions
 * (bytes = that matrix of one layer; flop = 2 * 128 * rows * K), 11 the five per-layer kernels of the multi-kernel path (a
 * tensor-parallel rank's too, without its exchanges) for all layers as one hipGraph: milliseconds and bytes per LAYER.
 * STATE: a measurement hook, not part of the generation path.  Kernels 0..6 and 11 run real kernels of the pass at the ctx's
 * current position (position 1 if none was run yet): they overwrite x, that position's K/V rows and the exchange state, kernel 11
 * for every layer; kernels 7..10 overwrite the prefill workspaces.  Call llmk_reset before generating on the ctx again. */
int llmk_time_kernel(llmk_ctx *ctx, int kernel, int iters, float *avg_ms, double *bytes_per_launch);

/* Debug/verification: copy internal device vectors to the host. which: 0 = x (residual stream, E),
 * 1 = q (E), 2 = xb (attention output, E), 3 = hb (H), 4 = key_cache row [layer][pos-1] (KV),
 * 5 = value_cache row (KV); debugging aids: 6, 7 = the debug library's trace stamps and a prefill workspace, 8 = the q4_0
 * persistent kernel's per-layer scale records of the last two positions ([2][128] x {max|xb|, max|hb|, pos, pos} as floats /
 * int bits: tests/host_tools/qsc_dump.py; LLMK_E_ARG on a ctx without the persistent kernel). */
int llmk_peek(llmk_ctx *ctx, int which, int layer, int pos, float *out, int n);

/* Which implementation of the token pass this ctx runs RIGHT NOW (it can change: a timed-out persistent kernel retires
 * to the multi-kernel path; llmk_set_tensor_type does the same): one of LLMK_PATH_*, or a negative LLMK_E_* code.
 * For labels in benchmarks and logs -- no reference counterpart. */
#define LLMK_PATH_MULTI_KERNEL 0     /* 5 launches per layer (csrc/kernels.h)                                        */
#define LLMK_PATH_TOKEN_KERNEL 1     /* persistent whole-token kernel (csrc/token_kernel.h)                          */
#define LLMK_PATH_TP_P2P 2           /* tensor-parallel rank: 6 launches per layer + one-shot peer-memory exchanges  */
#define LLMK_PATH_TP_RCCL 3          /* tensor-parallel rank: eager launches + ncclAllReduce / ncclAllGather         */
#define LLMK_PATH_TP_UNCONNECTED 4   /* The model shapes the persistent whole-token kernel is instantiated for in this build, as text:
 * "E,H,NH,NKV,V,type[+q6_K];..." (type = the matrices' f32 | f16 | q4_0; +q6_K = with a q6_K classifier).  The reference's dims
 * are compile-time parameters (llama2.f90:102-108); so are the kernel's -- `make TK_SHAPES="..."` adds shapes (llm.f90_amd/Makefile).
 * Any other shape runs the multi-kernel path (llmk_path).  LLMK_E_SIZE when `buf` is too small. */
int llmk_tk_shapes(char *buf, size_t n);

/* tensor-parallel rank without collectives yet (llmk_tp_segment stepping only) */
int llmk_path(llmk_ctx *ctx);
/* How many ranks this ctx's collective actually spans: ncclCommCount of its RCCL communicator, or the number of mapped
 * peer inboxes (+ itself) on the peer-memory path, or 1.  For the benchmark line of a multi-GPU run (`ranks_seen`). */
int llmk_tp_ranks_seen(llmk_ctx *ctx);

/* Verification: 64-bit sum of the 32-bit words of a tensor's DEVICE image (this rank's shard, device layout).  Equal images
 * give equal sums: two uploads of the same weights, or one context before and after a run, compare without a read-back. */
int llmk_tensor_checksum(llmk_ctx *ctx, int tensor_id, unsigned long long *out);

int llmk_destroy(llmk_ctx *ctx);

/* static string for an error code returned by any function above */
const char *llmk_strerror(int code);

/* library/ABI version: major*10000 + minor*100 + patch */
int llmk_version(void);

#ifdef __cplusplus
}
#endif
#endif /* LLMK_H */
