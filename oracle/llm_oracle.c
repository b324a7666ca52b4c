/*
 * TEST INFRASTRUCTURE -- NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, f32) of the reference's single-token forward pass,
 * `transformer` in /root/reference/llama2.f90:480-640, with run-time dims instead of the
 * reference's compile-time parameters (llama2.f90:102-108).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's library; the
 * product path (libllmk.so + the Fortran host) never links or calls it.
 *
 * Parity pin: validated against the REAL reference (built by oracle/build_ref.sh with
 * amdflang) on every shape in tests/golden/ -- see tests/golden/make_golden.py and
 * tests/test_oracle.py.  The reference ships no tests or golden vectors of its own
 * (SURVEY.md section 4), so reference-run outputs are the pin.
 *
 * Deliberately reproduced reference behaviour:
 *   - rmsnorm = x*w / sqrt(dot(x,x)/n + 1e-5), eps hard-coded            (llama2.f90:450-457)
 *   - RoPE pairs are interleaved (i,i+1); for 1-based odd i the exponent is mod(i,hs)/hs, i.e.
 *     pair j uses 10000^-((2j+1)/hs) (not 2j/hs), and the angle is pos*freq with 1-based pos
 *                                                                        (llama2.f90:543-559)
 *   - k is rotated only while (1-based) i < kv_dim                       (llama2.f90:553)
 *   - GQA: head h reads kv head h/kv_mul (intended semantics of the slice at :581/:591)
 *   - softmax subtracts the max, divides by the sum                      (llama2.f90:468-478)
 *   - SwiGLU: hb*(1/(1+exp(-hb)))*hb2                                    (llama2.f90:615-616)
 *   - sums run sequentially in f32 like a scalar dot_product (unless built with -ffast-math).
 *
 * Weight layout is the reference's weight_module (weight_module.f90:13-26): Fortran
 * (in,rows,layer) column-major == C [layer][row][in], QKV fused, gate/up fused.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int32_t emb_dim, hidden_dim, n_layers, n_heads, n_kv_heads, vocab_size, seq_len;
    const float *token_embedding_table; /* [V][E]        */
    const float *rms_att_weight;        /* [L][E]        */
    const float *rms_ffn_weight;        /* [L][E]        */
    const float *wqkv;                  /* [L][E+2KV][E] */
    const float *wo;                    /* [L][E][E]     */
    const float *w13;                   /* [L][2H][E]    */
    const float *w2;                    /* [L][E][H]     */
    const float *rms_final_weight;      /* [E]           */
    const float *wcls;                  /* [V][E]        */
    float *key_cache;                   /* [L][S][KV]    (RunState, weight_module.f90:33-40) */
    float *value_cache;                 /* [L][S][KV]    */
} oracle_model;

static float dotf(const float *a, const float *b, int n) {
    float s = 0.0f;
    for (int i = 0; i < n; ++i) s += a[i] * b[i];
    return s;
}

/* llama2.f90:450-457 */
/* The reference's constant (llama2.f90:454).  oracle_set_eps exists ONLY to check the product's opt-in extension
 * (llmk_set_rms_eps, llm --gguf-eps); every parity test runs with the default. */
static float g_eps = 1e-5f;
void oracle_set_eps(float eps) { g_eps = eps; }

static void rmsnorm(float *out, const float *x, const float *w, int n) {
    float xn = sqrtf(dotf(x, x, n) / (float)n + g_eps);
    for (int i = 0; i < n; ++i) out[i] = x[i] * w[i] / xn;
}

/* rows [r0,r1) of y = W x, W row-major [rows][n]; optionally accumulate (residual). */
static void gemv(float *y, const float *W, const float *x, int rows, int n, int accumulate) {
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int r = 0; r < rows; ++r) {
        float d = dotf(x, W + (size_t)r * n, n);
        y[r] = accumulate ? y[r] + d : d;
    }
}

/*
 * token, pos are 1-based as in the reference call `transformer(token,pos,s,weights)`
 * (llama2.f90:380).  logits: V floats.  trace (optional, may be NULL): (L+1)*E floats, the
 * residual stream x after each layer, then the final-normed x -- debugging aid for kernels.
 */
int oracle_forward(const oracle_model *m, int token, int pos, float *logits, float *trace) {
    const int E = m->emb_dim, H = m->hidden_dim, L = m->n_layers, nh = m->n_heads, nkv = m->n_kv_heads;
    const int V = m->vocab_size, S = m->seq_len;
    const int hs = E / nh, KV = nkv * hs, kv_mul = nh / nkv;
    if (token < 1 || token > V || pos < 1 || pos > S) return 1;
    float *x = (float *)malloc(sizeof(float) * (size_t)(E + E + (E + 2 * KV) + 2 * H + S));
    if (!x) return 2;
    float *xb = x + E, *qkv = xb + E, *hb13 = qkv + E + 2 * KV, *att = hb13 + 2 * H;
    float *q = qkv, *k = qkv + E, *v = qkv + E + KV;

    memcpy(x, m->token_embedding_table + (size_t)(token - 1) * E, sizeof(float) * E); /* :520 */

    for (int l = 0; l < L; ++l) {
        rmsnorm(xb, x, m->rms_att_weight + (size_t)l * E, E);                          /* :527 */
        gemv(qkv, m->wqkv + (size_t)l * (E + 2 * KV) * E, xb, E + 2 * KV, E, 0);       /* :529-531 */

        for (int i = 1; i <= E; i += 2) {                                             /* :543-559 */
            int head_dim = i % hs;
            float freq = 1.0f / powf(10000.0f, (float)head_dim / (float)hs);
            float rval = (float)pos * freq;
            float fcr = cosf(rval), fci = sinf(rval);
            float q0 = q[i - 1], q1 = q[i];
            q[i - 1] = q0 * fcr - q1 * fci;
            q[i] = q0 * fci + q1 * fcr;
            if (i < KV) {
                float k0 = k[i - 1], k1 = k[i];
                k[i - 1] = k0 * fcr - k1 * fci;
                k[i] = k0 * fci + k1 * fcr;
            }
        }
        float *kc = m->key_cache + (size_t)l * S * KV, *vc = m->value_cache + (size_t)l * S * KV;
        memcpy(kc + (size_t)(pos - 1) * KV, k, sizeof(float) * KV);                   /* :564 */
        memcpy(vc + (size_t)(pos - 1) * KV, v, sizeof(float) * KV);                   /* :565 */

        const float inv_scale = sqrtf((float)hs);
        for (int h = 0; h < nh; ++h) {                                                /* :574-598 */
            const float *qh = q + h * hs;
            const int g = h / kv_mul;
            for (int t = 0; t < pos; ++t) att[t] = dotf(qh, kc + (size_t)t * KV + g * hs, hs) / inv_scale;
            float mx = att[0];
            for (int t = 1; t < pos; ++t) mx = att[t] > mx ? att[t] : mx;
            float sum = 0.0f;
            for (int t = 0; t < pos; ++t) { att[t] = expf(att[t] - mx); sum += att[t]; }
            for (int t = 0; t < pos; ++t) att[t] = att[t] / sum;
            float *xbh = xb + h * hs;
            for (int d = 0; d < hs; ++d) xbh[d] = 0.0f;
            for (int t = 0; t < pos; ++t) {
                const float a = att[t];
                const float *vt = vc + (size_t)t * KV + g * hs;
                for (int d = 0; d < hs; ++d) xbh[d] = xbh[d] + a * vt[d];
            }
        }

        gemv(x, m->wo + (size_t)l * E * E, xb, E, E, 1);                               /* :603-605 */
        rmsnorm(xb, x, m->rms_ffn_weight + (size_t)l * E, E);                          /* :608 */
        gemv(hb13, m->w13 + (size_t)l * 2 * H * E, xb, 2 * H, E, 0);                   /* :610-612 */
        for (int i = 0; i < H; ++i) {                                                 /* :615-616 */
            float hb = hb13[i];
            hb = hb * (1.0f / (1.0f + expf(-hb)));
            hb13[i] = hb * hb13[H + i];
        }
        gemv(x, m->w2 + (size_t)l * E * H, hb13, E, H, 1);                             /* :618-620 */
        if (trace) memcpy(trace + (size_t)l * E, x, sizeof(float) * E);
    }

    rmsnorm(x, x, m->rms_final_weight, E);                                            /* :627 */
    if (trace) memcpy(trace + (size_t)L * E, x, sizeof(float) * E);
    gemv(logits, m->wcls, x, V, E, 0);                                                /* :634-636 */
    free(x);
    return 0;
}

/* first maximum wins, 1-based, like maxloc(logits,DIM=1) at llama2.f90:388 */
int oracle_argmax1(const float *logits, int n) {
    int best = 0;
    for (int i = 1; i < n; ++i) if (logits[i] > logits[best]) best = i;
    return best + 1;
}

/*
 * Greedy generation exactly as the reference loop does it (llama2.f90:376-402): BOS is 1-based
 * token 2; positions 1..n; prompt tokens (1-based ids) are fed while pos <= n_prompt.
 * tokens_out[n] receives the token chosen after each position; logits_out (optional) n*V.
 */
int oracle_generate(const oracle_model *m, const int *prompt, int n_prompt, int n, int *tokens_out,
                    float *logits_out) {
    float *logits = (float *)malloc(sizeof(float) * (size_t)m->vocab_size);
    if (!logits) return 2;
    int token = 2;
    for (int pos = 1; pos <= n; ++pos) {
        int rc = oracle_forward(m, token, pos, logits, NULL);
        if (rc) { free(logits); return rc; }
        if (logits_out) memcpy(logits_out + (size_t)(pos - 1) * m->vocab_size, logits, sizeof(float) * m->vocab_size);
        token = pos <= n_prompt ? prompt[pos - 1] : oracle_argmax1(logits, m->vocab_size);
        tokens_out[pos - 1] = token;
    }
    free(logits);
    return 0;
}
