/*
 * icrec_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, CPU restatement of the reference's hot path
 *   SentenceTransformer.encode -> cos_sim -> argsort -> exclusion/top-k loop
 * (reference: src/inference/serve_recommendations.py:195-200, 206-225, 236-262).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the product path (libicrec.so) never does.
 *
 * The arithmetic itself lives in third-party packages the reference pins
 * (uv.lock): transformers 5.1.0 (BertModel), sentence-transformers 5.2.2
 * (Pooling / Normalize / util.cos_sim), torch 2.10.0 (kernels).  What is
 * restated here, with the file:line it follows
 * (tf = transformers/models/bert/modeling_bert.py as installed, 5.15.0):
 *   embeddings        tf:68-108   LN(word[ids] + type[0] + pos[0:L]), eps 1e-12
 *   self-attention    tf:111-136, 164-203  softmax(QK^T/sqrt(dh) + mask) V
 *   self-output       tf:282-293  LN(dense(ctx) + x)
 *   intermediate      tf:325-337  gelu_erf(dense(x))
 *   output            tf:340-351  LN(dense(h) + x)
 *   layer loop        tf:419-448
 *   pooling           sentence_transformers Pooling(mean): sum(m*h)/clamp(sum m,1e-9)
 *   Normalize         F.normalize(p=2, dim=1, eps=1e-12)
 *   cos_sim           F.normalize(a), F.normalize(b), mm(a, b^T)
 *   ranking           scores.argsort(descending=True) then skip excluded ids
 *
 * PINNING STATUS: the reference's own tests hold no
 * numeric vector for this path (tests/conftest.py:28-33 mocks the recommender)
 * and the reference module cannot be imported here (sentence_transformers,
 * dotenv, slowapi are not installed).  The encoder restatement is pinned
 * against the installed transformers.BertModel and the similarity/ranking
 * restatement against torch.nn.functional.normalize + torch.mm + torch.argsort
 * by oracle/pin_against_libs.py, whose outputs are the fixtures in
 * tests/golden/.  The sentence-transformers glue (Pooling / Normalize /
 * cos_sim wrappers) is restated from its documented behaviour: that part is
 * "parity unpinned".
 *
 * Floating-point order.  Reductions are written in one fixed, documented order
 * so that the HIP kernels can reproduce them bit for bit where they choose to:
 *   dot products / GEMM outputs : acc = 0; for k ascending: acc = fmaf(a[k], b[k], acc)
 *   row reductions over d elems : 64 strided partial sums (lane l takes l, l+64, ...),
 *                                 then a butterfly (xor 32,16,8,4,2,1)
 *   attention P.V and softmax sum : the key order of the MFMA accumulator layout (see the
 *                                 attention block below and csrc/encoder.hip:attention_kernel)
 * Ties in ranking: higher score first, then LOWER row index first (the
 * reference's torch.argsort(descending=True) is unstable on ties; this is the
 * build's own documented policy).
 *
 * Build: gcc -O2 -fPIC -shared -ffp-contract=off -fopenmp (see oracle/Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_API __attribute__((visibility("default")))

typedef struct {
    int32_t vocab_size, hidden, layers, heads, intermediate, max_position, type_vocab;
    float ln_eps;
    int32_t n_normalize;
    int32_t gemm_mode; /* HIP engine option; unused here (the oracle is always exact fp32) */
} oracle_bert_cfg; /* same layout as icrec_bert_cfg in include/icrec.h */

/* ---------------------------------------------------------------- helpers */

/* 64 strided partials + xor butterfly: the order a 64-lane wavefront uses. */
static float wave_sum(const float* v, int n, int mode, float shift) {
    /* mode 0: sum v[i]; mode 1: sum (v[i]-shift)^2 via fmaf; mode 2: sum v[i]^2 via fmaf */
    float part[64];
    for (int l = 0; l < 64; ++l) {
        float acc = 0.0f;
        for (int i = l; i < n; i += 64) {
            if (mode == 0) acc = acc + v[i];
            else if (mode == 1) { float d = v[i] - shift; acc = fmaf(d, d, acc); }
            else acc = fmaf(v[i], v[i], acc);
        }
        part[l] = acc;
    }
    for (int m = 32; m >= 1; m >>= 1) {
        float nxt[64];
        for (int l = 0; l < 64; ++l) nxt[l] = part[l] + part[l ^ m];
        memcpy(part, nxt, sizeof part);
    }
    return part[0];
}

/* LayerNorm over one row of H (tf:106-107, 292, 350 -> torch.nn.LayerNorm):
 * biased variance, y = (x-mean) * rsqrt(var+eps) * g + b. */
static void layer_norm_row(const float* x, const float* g, const float* b, float eps,
                           int H, float* y) {
    float mean = wave_sum(x, H, 0, 0.0f) / (float)H;
    float var = wave_sum(x, H, 1, mean) / (float)H;
    float rstd = 1.0f / sqrtf(var + eps);
    for (int i = 0; i < H; ++i) y[i] = fmaf((x[i] - mean) * rstd, g[i], b[i]);
}

/* out[M,N] = A[M,K] . W[N,K]^T + bias[N]; every element is the k-ascending
 * fmaf chain from 0, bias added last (torch.nn.Linear, tf:175-177, 290, 335, 348). */
static void linear(const float* A, const float* W, const float* bias, int M, int N, int K,
                   float* out) {
    /* transpose W once so the inner loop runs over output columns (SIMD across
     * independent chains, each chain still strictly k-ascending). */
    float* Wt = (float*)malloc((size_t)K * N * sizeof(float));
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < K; ++k) Wt[(size_t)k * N + n] = W[(size_t)n * K + k];
#pragma omp parallel for schedule(static)
    for (int m = 0; m < M; ++m) {
        float* o = out + (size_t)m * N;
        for (int n = 0; n < N; ++n) o[n] = 0.0f;
        const float* a = A + (size_t)m * K;
        for (int k = 0; k < K; ++k) {
            const float ak = a[k];
            const float* w = Wt + (size_t)k * N;
            for (int n = 0; n < N; ++n) o[n] = fmaf(ak, w[n], o[n]);
        }
        for (int n = 0; n < N; ++n) o[n] = o[n] + bias[n];
    }
    free(Wt);
}

static inline float gelu_erf(float x) { /* tf:336 ACT2FN["gelu"] = x*0.5*(1+erf(x/sqrt2)) */
    return x * 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
}

/* ------------------------------------------------------------ the encoder */

typedef struct {
    const float *word, *pos, *type, *eg, *eb;
} emb_w;
typedef struct {
    const float *Wq, *bq, *Wk, *bk, *Wv, *bv, *Wo, *bo, *g1, *b1n, *W1, *b1, *W2, *b2, *g2, *b2n;
} layer_w;

ORACLE_API size_t icrec_oracle_weight_count(const oracle_bert_cfg* c) {
    size_t H = c->hidden, I = c->intermediate;
    size_t n = (size_t)c->vocab_size * H + (size_t)c->max_position * H + (size_t)c->type_vocab * H + 2 * H;
    size_t per = 4 * (H * H + H) + 2 * H + (I * H + I) + (H * I + H) + 2 * H;
    return n + per * c->layers;
}

/* Encode a token-packed batch; out[n_seqs, H].  Returns 0, or -1 on bad args.
 * If `hidden_out` is non-NULL it receives the last layer's hidden states
 * [total_tokens, H] (for per-stage parity checks). */
ORACLE_API int icrec_oracle_encode(const float* w, const oracle_bert_cfg* c, const int32_t* ids,
                                   const int32_t* cu, int n_seqs, float* out, float* hidden_out) {
    const int H = c->hidden, I = c->intermediate, NH = c->heads;
    if (H % NH) return -1;
    const int DH = H / NH;
    const int T = cu[n_seqs];
    emb_w e;
    const float* p = w;
    e.word = p; p += (size_t)c->vocab_size * H;
    e.pos = p;  p += (size_t)c->max_position * H;
    e.type = p; p += (size_t)c->type_vocab * H;
    e.eg = p;   p += H;
    e.eb = p;   p += H;

    float* x = (float*)malloc((size_t)T * H * sizeof(float));
    float* q = (float*)malloc((size_t)T * H * sizeof(float));
    float* k = (float*)malloc((size_t)T * H * sizeof(float));
    float* v = (float*)malloc((size_t)T * H * sizeof(float));
    float* ctx = (float*)malloc((size_t)T * H * sizeof(float));
    float* t1 = (float*)malloc((size_t)T * H * sizeof(float));
    float* hbuf = (float*)malloc((size_t)T * I * sizeof(float));
    if (!x || !q || !k || !v || !ctx || !t1 || !hbuf) return -1;

    /* embeddings tf:98-107: (word + type[0]) + pos, then LayerNorm */
    for (int s = 0; s < n_seqs; ++s)
        for (int t = cu[s]; t < cu[s + 1]; ++t) {
            int id = ids[t], ps = t - cu[s];
            if (id < 0 || id >= c->vocab_size || ps >= c->max_position) return -1;
            float tmp[4096];
            for (int i = 0; i < H; ++i)
                tmp[i] = (e.word[(size_t)id * H + i] + e.type[i]) + e.pos[(size_t)ps * H + i];
            layer_norm_row(tmp, e.eg, e.eb, c->ln_eps, H, x + (size_t)t * H);
        }

    const float scale = 1.0f / sqrtf((float)DH); /* tf:117 scaling = head_dim**-0.5 */
    for (int l = 0; l < c->layers; ++l) {
        layer_w L;
        L.Wq = p; p += (size_t)H * H; L.bq = p; p += H;
        L.Wk = p; p += (size_t)H * H; L.bk = p; p += H;
        L.Wv = p; p += (size_t)H * H; L.bv = p; p += H;
        L.Wo = p; p += (size_t)H * H; L.bo = p; p += H;
        L.g1 = p; p += H; L.b1n = p; p += H;
        L.W1 = p; p += (size_t)I * H; L.b1 = p; p += I;
        L.W2 = p; p += (size_t)H * I; L.b2 = p; p += H;
        L.g2 = p; p += H; L.b2n = p; p += H;

        linear(x, L.Wq, L.bq, T, H, H, q);
        linear(x, L.Wk, L.bk, T, H, H, k);
        linear(x, L.Wv, L.bv, T, H, H, v);

        /* attention per (sequence, head); packed form = padded form because pad
         * keys get weight exp(-inf)=0 under the additive mask (tf:120-127). */
#pragma omp parallel for schedule(dynamic)
        for (int sh = 0; sh < n_seqs * NH; ++sh) {
            int s = sh / NH, h = sh % NH;
            int t0 = cu[s], Ls = cu[s + 1] - cu[s];
            float* sc = (float*)malloc((size_t)Ls * sizeof(float));
            for (int i = 0; i < Ls; ++i) {
                const float* qi = q + (size_t)(t0 + i) * H + h * DH;
                float mx = -INFINITY;
                for (int j = 0; j < Ls; ++j) {
                    const float* kj = k + (size_t)(t0 + j) * H + h * DH;
                    float acc = 0.0f;
                    for (int d = 0; d < DH; ++d) acc = fmaf(qi[d], kj[d], acc);
                    sc[j] = acc * scale;
                    if (sc[j] > mx) mx = sc[j];
                }
                /* softmax numerator, denominator split by key parity class
                 * ((j>>2)&1) — the two half-wave partial sums of the kernel —
                 * each ascending in j, then added. */
                float l0 = 0.0f, l1 = 0.0f;
                for (int j = 0; j < Ls; ++j) {
                    sc[j] = expf(sc[j] - mx);
                    if ((j >> 2) & 1) l1 += sc[j]; else l0 += sc[j];
                }
                float lsum = l0 + l1;
                float* ci = ctx + (size_t)(t0 + i) * H + h * DH;
                /* P.V chain: per 32-key tile the keys enter in the order the kernel's
                 * accumulator registers hold them (e = 0..15: key (e&3)+8(e>>2), then +4). */
                for (int d = 0; d < DH; ++d) {
                    float acc = 0.0f;
                    for (int kt = 0; kt * 32 < Ls; ++kt)
                        for (int e = 0; e < 16; ++e)
                            for (int hh = 0; hh < 2; ++hh) {
                                int j = kt * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                                if (j < Ls) acc = fmaf(sc[j], v[(size_t)(t0 + j) * H + h * DH + d], acc);
                            }
                    ci[d] = acc / lsum;
                }
            }
            free(sc);
        }

        /* self-output tf:289-293 */
        linear(ctx, L.Wo, L.bo, T, H, H, t1);
        for (int t = 0; t < T; ++t) {
            float tmp[4096];
            for (int i = 0; i < H; ++i) tmp[i] = t1[(size_t)t * H + i] + x[(size_t)t * H + i];
            layer_norm_row(tmp, L.g1, L.b1n, c->ln_eps, H, x + (size_t)t * H);
        }
        /* intermediate tf:334-337, output tf:347-351 */
        linear(x, L.W1, L.b1, T, I, H, hbuf);
        for (size_t i = 0; i < (size_t)T * I; ++i) hbuf[i] = gelu_erf(hbuf[i]);
        linear(hbuf, L.W2, L.b2, T, H, I, t1);
        for (int t = 0; t < T; ++t) {
            float tmp[4096];
            for (int i = 0; i < H; ++i) tmp[i] = t1[(size_t)t * H + i] + x[(size_t)t * H + i];
            layer_norm_row(tmp, L.g2, L.b2n, c->ln_eps, H, x + (size_t)t * H);
        }
    }
    if (hidden_out) memcpy(hidden_out, x, (size_t)T * H * sizeof(float));

    /* mean pooling: sum over tokens (ascending) / clamp(count, 1e-9); then
     * n_normalize times x / max(|x|, 1e-12). */
    for (int s = 0; s < n_seqs; ++s) {
        float* o = out + (size_t)s * H;
        int Ls = cu[s + 1] - cu[s];
        float cnt = (float)Ls;
        if (cnt < 1e-9f) cnt = 1e-9f;
        for (int i = 0; i < H; ++i) {
            float acc = 0.0f;
            for (int t = cu[s]; t < cu[s + 1]; ++t) acc = acc + x[(size_t)t * H + i];
            o[i] = acc / cnt;
        }
        for (int r = 0; r < c->n_normalize; ++r) {
            float nrm = sqrtf(wave_sum(o, H, 2, 0.0f));
            float den = nrm > 1e-12f ? nrm : 1e-12f;
            for (int i = 0; i < H; ++i) o[i] = o[i] / den;
        }
    }
    free(x); free(q); free(k); free(v); free(ctx); free(t1); free(hbuf);
    return 0;
}

/* ------------------------------------------------ similarity and ranking */

/* F.normalize(x, p=2, dim=1, eps): out = x / max(|x|_2, eps)  (cos_sim's first step). */
ORACLE_API void icrec_oracle_normalize_rows(const float* x, float* out, int64_t n, int d, float eps) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < n; ++r) {
        const float* xi = x + r * d;
        float nrm = sqrtf(wave_sum(xi, d, 2, 0.0f));
        float den = nrm > eps ? nrm : eps;
        for (int i = 0; i < d; ++i) out[r * d + i] = xi[i] / den;
    }
}

/* scores[Q,N] = qhat . phat^T, each the j-ascending fmaf chain (torch.mm in cos_sim). */
ORACLE_API void icrec_oracle_scores(const float* qhat, const float* phat, int Q, int64_t N, int d,
                                    float* scores) {
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < N; ++n) {
        const float* pr = phat + n * d;
        for (int qi = 0; qi < Q; ++qi) {
            const float* qr = qhat + (size_t)qi * d;
            float acc = 0.0f;
            for (int j = 0; j < d; ++j) acc = fmaf(qr[j], pr[j], acc);
            scores[(size_t)qi * N + n] = acc + 0.0f; /* -0 -> +0, as the kernel does */
        }
    }
}

typedef struct { float s; int64_t i; } hit;
static int hit_cmp(const void* a, const void* b) {
    const hit* x = (const hit*)a; const hit* y = (const hit*)b;
    if (x->s > y->s) return -1;
    if (x->s < y->s) return 1;
    return (x->i > y->i) - (x->i < y->i);
}

/* The reference's ranking loop (serve_recommendations.py:215-225): full
 * descending argsort, walk it, skip excluded rows, stop at k.
 * excl: sorted-or-not list of LOCAL row numbers (n_excl may be 0).
 * out_idx gets row_offset+row, -1 padded; out_score 0 padded. */
ORACLE_API void icrec_oracle_rank(const float* scores, int64_t N, int k, const int32_t* excl,
                                  int n_excl, int64_t row_offset, int64_t* out_idx,
                                  float* out_score) {
    hit* h = (hit*)malloc((size_t)N * sizeof(hit));
    for (int64_t n = 0; n < N; ++n) { h[n].s = scores[n]; h[n].i = n; }
    qsort(h, (size_t)N, sizeof(hit), hit_cmp);
    unsigned char* ex = (unsigned char*)calloc((size_t)N + 1, 1);
    for (int e = 0; e < n_excl; ++e)
        if (excl[e] >= 0 && excl[e] < N) ex[excl[e]] = 1;
    int got = 0;
    for (int64_t n = 0; n < N && got < k; ++n) {
        if (ex[h[n].i]) continue;
        out_idx[got] = row_offset + h[n].i;
        out_score[got] = h[n].s;
        ++got;
    }
    for (; got < k; ++got) { out_idx[got] = -1; out_score[got] = 0.0f; }
    free(h); free(ex);
}

/* cos_sim + ranking for a batch of queries against a raw (un-normalised)
 * catalog, exactly the call sequence of Recommender.recommend.
 * excl_idx/excl_off: CSR per query (may be NULL). */
ORACLE_API void icrec_oracle_search(const float* q, const float* P, int Q, int64_t N, int d, int k,
                                    const int32_t* excl_idx, const int32_t* excl_off,
                                    int64_t row_offset, int64_t* out_idx, float* out_score) {
    float* qh = (float*)malloc((size_t)Q * d * sizeof(float));
    float* ph = (float*)malloc((size_t)N * d * sizeof(float));
    float* sc = (float*)malloc((size_t)Q * N * sizeof(float));
    icrec_oracle_normalize_rows(q, qh, Q, d, 1e-12f);
    icrec_oracle_normalize_rows(P, ph, N, d, 1e-12f);
    icrec_oracle_scores(qh, ph, Q, N, d, sc);
#pragma omp parallel for schedule(dynamic)
    for (int qi = 0; qi < Q; ++qi) {
        const int32_t* ex = excl_idx && excl_off ? excl_idx + excl_off[qi] : NULL;
        int ne = excl_idx && excl_off ? excl_off[qi + 1] - excl_off[qi] : 0;
        icrec_oracle_rank(sc + (size_t)qi * N, N, k, ex, ne, row_offset,
                          out_idx + (size_t)qi * k, out_score + (size_t)qi * k);
    }
    free(qh); free(ph); free(sc);
}

/* ICREC_ROWS_BF16 storage (include/icrec.h): each value rounded to bfloat16, round-to-nearest-even,
 * returned widened to fp32 (torch's .to(torch.bfloat16).float(); pinned against it in tests/test_oracle.py). */
ORACLE_API void icrec_oracle_round_bf16(const float* x, float* out, int64_t n) {
    for (int64_t i = 0; i < n; ++i) {
        uint32_t u;
        memcpy(&u, x + i, 4);
        u = (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
        memcpy(out + i, &u, 4);
    }
}

/* icrec_oracle_search with the catalog held as bf16: cos_sim's normalisation, THEN the rounding of the
 * normalised rows, then the same fp32 fmaf chains and the same ranking loop. */
ORACLE_API void icrec_oracle_search_bf16(const float* q, const float* P, int Q, int64_t N, int d, int k,
                                         const int32_t* excl_idx, const int32_t* excl_off,
                                         int64_t row_offset, int64_t* out_idx, float* out_score) {
    float* qh = (float*)malloc((size_t)Q * d * sizeof(float));
    float* ph = (float*)malloc((size_t)N * d * sizeof(float));
    float* sc = (float*)malloc((size_t)Q * N * sizeof(float));
    icrec_oracle_normalize_rows(q, qh, Q, d, 1e-12f);
    icrec_oracle_normalize_rows(P, ph, N, d, 1e-12f);
    icrec_oracle_round_bf16(ph, ph, N * d);
    icrec_oracle_scores(qh, ph, Q, N, d, sc);
#pragma omp parallel for schedule(dynamic)
    for (int qi = 0; qi < Q; ++qi) {
        const int32_t* ex = excl_idx && excl_off ? excl_idx + excl_off[qi] : NULL;
        int ne = excl_idx && excl_off ? excl_off[qi + 1] - excl_off[qi] : 0;
        icrec_oracle_rank(sc + (size_t)qi * N, N, k, ex, ne, row_offset,
                          out_idx + (size_t)qi * k, out_score + (size_t)qi * k);
    }
    free(qh); free(ph); free(sc);
}

/* Merge per-shard (idx, score) lists [n_lists, Q, k] into the global top-k under
 * the same order; -1 entries are pads.  (Restates "rank the union".) */
ORACLE_API void icrec_oracle_merge(const int64_t* idx, const float* score, int n_lists, int Q, int k,
                                   int64_t* out_idx, float* out_score) {
    for (int qi = 0; qi < Q; ++qi) {
        hit* h = (hit*)malloc((size_t)n_lists * k * sizeof(hit));
        int m = 0;
        for (int l = 0; l < n_lists; ++l)
            for (int j = 0; j < k; ++j) {
                size_t o = ((size_t)l * Q + qi) * k + j;
                if (idx[o] >= 0) { h[m].s = score[o]; h[m].i = idx[o]; ++m; }
            }
        qsort(h, (size_t)m, sizeof(hit), hit_cmp);
        for (int j = 0; j < k; ++j) {
            out_idx[(size_t)qi * k + j] = j < m ? h[j].i : -1;
            out_score[(size_t)qi * k + j] = j < m ? h[j].s : 0.0f;
        }
        free(h);
    }
}

/* Cap the OpenMP team (bench.py sizes it to the container's CPU quota: oversubscribing a 16-CPU cgroup with
 * 128 threads makes the baseline 2x slower batched and 8x slower per request). */
ORACLE_API void icrec_oracle_set_threads(int n) {
#ifdef _OPENMP
    if (n >= 1) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

ORACLE_API int icrec_oracle_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
