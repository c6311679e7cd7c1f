/*
 * sjd_oracle.c -- CPU restatement of the Speculative-Jacobi-Decoding scheduler step.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the *checker* for the HIP kernels in
 * accelerating-t2i-ar-with-sjd_amd/csrc/.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it; the product never does.
 *
 * What it restates (reference = /root/reference, tyshiwo1/Accelerating-T2I-AR-with-SJD):
 *   sjd_o_logits_to_probs_sample  <- sampling_logits2tokens          scheduler/jacobi_iteration_lumina_mgpt.py:82-132
 *                                    + grammar masks / forced rows   scheduler/logit_processor_3dim.py:31-43,125-145
 *                                    + top-k                         scheduler/logit_processor_3dim.py:196-204
 *                                    + top-p                         scheduler/logit_processor_3dim.py:406-419
 *                                    + multinomial == argmax(p/Exp1) (ATen multinomial fast path; noise is an INPUT)
 *   sjd_o_verify_accept           <- SpeculativeSampler.__call__     scheduler/jacobi_iteration_lumina_mgpt.py:247-315
 *                                    + reject_sampling_single_token  :203-241
 *   sjd_o_first_mismatch          <- find_first_misaligned_token_inds :317-333
 *
 * Pinning: checked against fixtures produced by importing the reference in the build
 * container (tests/golden/make_golden.py -> tests/golden/ *.npz); see tests/test_oracle_golden.py.
 *
 * Canonical fp32 numerics (shared *specification* with the HIP kernels, so that HIP == oracle
 * bit-for-bit; the reference's torch softmax differs from this by a few ulp, tolerance 1e-6):
 *   - no FMA contraction except the explicit fmaf() below (build with -ffp-contract=off);
 *   - exp:   sjd_expf (Cody-Waite + degree-5 polynomial in fmaf form), 0 below -87;
 *   - sum:   4096 accumulators, column i -> accumulator (i mod 4096), added in increasing i;
 *            lane T (0..1023) owns accumulators 4T..4T+3: s_T = (a0+a1)+(a2+a3);
 *            64-lane xor-butterfly (offsets 32,16,8,4,2,1) inside each of the 16 waves,
 *            then the 16 wave totals are added in wave order;
 *   - p = e / S (IEEE division);  token = lowest-index argmax of p / noise;
 *   - top-p: the reference removes the ascending-sorted prefix whose cumulative probability is <= 1 - top_p.  Restated
 *     order-independently: with T(key) = canonical_sum{ p_i : key_i <= key } (key = the usual monotone uint32 image of
 *     the float weight e_i = exp(z_i - max), resp. d_i), K* = the largest 32-bit key with T(K*) <= thr (bitwise descent, 32 sums) and every entry with
 *     key <= K* except the row maximum is removed.  Identical to the reference except for exact value ties at the
 *     cut and ~1e-7 relative differences of the fp32 cumulative sums (reference: double running sum);
 *   - residual resample uses r = d / sum(d), d = max(p - q, 0), which equals the reference's
 *     softmax(log d) in real arithmetic (reference :203-207,232).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define SJD_MAX_RANGES 4

typedef struct {
    int32_t n_ranges;              /* 0 => every column allowed (identity grammar, LP:89-92) */
    int32_t lo[SJD_MAX_RANGES];    /* allowed columns are the union of [lo, hi) */
    int32_t hi[SJD_MAX_RANGES];
    int32_t forced;                /* >=0: row is -inf except [forced]=0 (LP:39-41); -1: none */
    int32_t top_k;                 /* <=0 or >=V: no top-k */
    float   top_p_thr;             /* float32(1 - top_p) (LP:411); < 0: no top-p */
    float   temperature;           /* HF TemperatureLogitsWarper: scores / temperature after the processors above, before top-p and
                                    * the softmax (transformers 4.47.1 generation/utils.py `_get_logits_processor`: warpers follow the
                                    * user's processors; logits_process.py TemperatureLogitsWarper.__call__); 1 or <= 0: off */
} sjd_row_rule;

/* ---------------------------------------------------------------- canonical numerics */
static float sjd_expf(float x)
{
    if (!(x >= -87.0f)) return 0.0f;         /* also catches -inf and NaN */
    const float LOG2E = 1.44269504088896341f;
    const float LN2_HI = 0.693359375f;
    const float LN2_LO = -2.12194440e-4f;
    const float MAGIC = 12582912.0f;          /* 1.5 * 2^23: round-to-nearest-even via add/sub */
    float n = fmaf(x, LOG2E, MAGIC) - MAGIC;
    float r = fmaf(-n, LN2_HI, x);
    r = fmaf(-n, LN2_LO, r);
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    float r2 = r * r;
    float y = fmaf(p, r2, r) + 1.0f;
    int32_t ni = (int32_t)n;                  /* -126 <= ni <= 0 here */
    union { uint32_t u; float f; } s;
    s.u = (uint32_t)(ni + 127) << 23;
    return y * s.f;
}

/* canonical natural logarithm of a positive finite float (musl / fdlibm logf), operations rounded separately in the order written;
 * the HIP side (csrc/sjd_device.h::sjd_logf) is the same sequence.  Used only by the temperature path of the residual resample. */
static float sjd_logf(float x)
{
    const float LN2_HI = 6.9313812256e-01f, LN2_LO = 9.0580006145e-06f;
    const float LG1 = 0.66666662693f, LG2 = 0.40000972152f, LG3 = 0.28498786688f, LG4 = 0.24279078841f;
    union { float f; uint32_t u; } c;
    c.f = x;
    uint32_t ix = c.u;
    int k = 0;
    if (ix < 0x00800000u) { x = x * 33554432.0f; k = -25; c.f = x; ix = c.u; }
    ix += 0x3f800000u - 0x3f3504f3u;
    k += (int)(ix >> 23) - 0x7f;
    ix = (ix & 0x007fffffu) + 0x3f3504f3u;
    c.u = ix;
    const float xr = c.f;
    const float f = xr - 1.0f;
    const float s = f / (2.0f + f);
    const float z = s * s;
    const float w = z * z;
    const float t1 = w * (LG2 + w * LG4);
    const float t2 = z * (LG1 + w * LG3);
    const float R = t2 + t1;
    const float hfsq = (0.5f * f) * f;
    const float dk = (float)k;
    float r = s * (hfsq + R);
    r = r + dk * LN2_LO;
    r = r - hfsq;
    r = r + f;
    r = r + dk * LN2_HI;
    return r;
}

static float sjd_canonical_sum(const float *v, int V)
{
    float acc[4096];                     /* on the stack: the row loop of the K2 restatement runs one row per OpenMP thread */
    for (int a = 0; a < 4096; ++a) acc[a] = 0.0f;
    for (int i = 0; i < V; ++i) acc[i & 4095] = acc[i & 4095] + v[i];
    float lane[1024];
    for (int t = 0; t < 1024; ++t)
        lane[t] = (acc[4 * t] + acc[4 * t + 1]) + (acc[4 * t + 2] + acc[4 * t + 3]);
    float total = 0.0f;
    for (int w = 0; w < 16; ++w) {
        float s[64], n[64];
        memcpy(s, lane + 64 * w, sizeof s);
        for (int off = 32; off >= 1; off >>= 1) {
            for (int l = 0; l < 64; ++l) n[l] = s[l] + s[l ^ off];
            memcpy(s, n, sizeof s);
        }
        total = (w == 0) ? s[0] : total + s[0];
    }
    return total;
}

static int cmp_desc(const void *a, const void *b)
{
    float x = *(const float *)a, y = *(const float *)b;
    return (x < y) - (x > y);
}

/* k-th largest value of z[0..V) (k is 1-based); torch.topk(scores,k)[0][...,-1] */
static float kth_largest(const float *z, int V, int k, float *scratch)
{
    memcpy(scratch, z, (size_t)V * sizeof(float));
    qsort(scratch, (size_t)V, sizeof(float), cmp_desc);
    return scratch[k - 1];
}

static int col_allowed(const sjd_row_rule *r, int i)
{
    if (r->n_ranges == 0) return 1;
    for (int a = 0; a < r->n_ranges; ++a)
        if (i >= r->lo[a] && i < r->hi[a]) return 1;
    return 0;
}

static uint32_t f2key(float z)
{
    union { float f; uint32_t u; } c;
    c.f = z;
    return (c.u & 0x80000000u) ? ~c.u : (c.u | 0x80000000u);
}

/* top-p cut on non-negative weights w (p_i = w_i / S): returns the key K* described in the header */
static uint32_t top_p_cut_key(const float *keyval, const float *w, float S, int V, float thr, float *scratch)
{
    uint32_t K = 0;
    for (int bit = 31; bit >= 0; --bit) {
        uint32_t cand = K | (1u << bit);
        for (int i = 0; i < V; ++i) scratch[i] = (w[i] > 0.0f && f2key(keyval[i]) <= cand) ? w[i] / S : 0.0f;
        if (sjd_canonical_sum(scratch, V) <= thr) K = cand;
    }
    return K;
}

/* grammar + top-k + top-p on one row of (already CFG-combined) scores z, in place (-inf = removed) */
static void apply_rule(float *z, int V, const sjd_row_rule *r, float *scratch)
{
    if (r->forced >= 0) {                                  /* LP:39-41 */
        for (int i = 0; i < V; ++i) z[i] = -INFINITY;
        z[r->forced] = 0.0f;
    } else if (r->n_ranges > 0) {                          /* LP:125-129 / JE:80-82 */
        for (int i = 0; i < V; ++i) if (!col_allowed(r, i)) z[i] = -INFINITY;
    }
    if (r->top_k > 0 && r->top_k < V) {                    /* LP:196-204 */
        float kth = kth_largest(z, V, r->top_k, scratch);
        for (int i = 0; i < V; ++i) if (z[i] < kth) z[i] = -INFINITY;
    }
    if (r->temperature > 0.0f && r->temperature != 1.0f)   /* TemperatureLogitsWarper: after the grammar and its top-k, before top-p */
        for (int i = 0; i < V; ++i) if (z[i] > -INFINITY) z[i] = z[i] / r->temperature;
    if (r->top_p_thr >= 0.0f) {                            /* LP:406-419 */
        float m = -INFINITY;
        for (int i = 0; i < V; ++i) if (z[i] > m) m = z[i];
        float *e = (float *)malloc((size_t)V * sizeof(float));
        for (int i = 0; i < V; ++i) e[i] = sjd_expf(z[i] - m);
        float S = sjd_canonical_sum(e, V);
        float em = 0.0f;                                  /* the cut is keyed on the weights e (monotone in z) */
        int imax = 0;
        for (int i = 0; i < V; ++i) if (e[i] > em) { em = e[i]; imax = i; }
        uint32_t K = top_p_cut_key(e, e, S, V, r->top_p_thr, scratch);
        for (int i = 0; i < V; ++i) if (i != imax && e[i] > 0.0f && f2key(e[i]) <= K) z[i] = -INFINITY;
        free(e);
    }
}

static void softmax_canonical(const float *z, int V, float *p)
{
    float m = -INFINITY;
    for (int i = 0; i < V; ++i) if (z[i] > m) m = z[i];
    for (int i = 0; i < V; ++i) p[i] = sjd_expf(z[i] - m);
    float S = sjd_canonical_sum(p, V);
    for (int i = 0; i < V; ++i) p[i] = p[i] / S;
}

static int64_t argmax_ratio(const float *p, const float *noise, int V)
{
    int64_t best = 0;
    float bv = -INFINITY;
    for (int i = 0; i < V; ++i) {
        float r = p[i] / noise[i];
        if (r > bv) { bv = r; best = i; }
    }
    return best;
}

/* ---------------------------------------------------------------- K2 restatement */
/* logits_c / logits_u: [n_rows, V] fp32 (u may be NULL => no CFG, JL:101-102).
 * noise: [n_rows, V] Exp(1) samples.  probs_out: [n_rows, V].  tokens_out: [n_rows]. */
int sjd_o_logits_to_probs_sample(const float *logits_c, const float *logits_u, float guidance,
                                 int n_rows, int V, const sjd_row_rule *rules, const float *noise,
                                 float *probs_out, int64_t *tokens_out)
{
    /* window rows are independent (the reference's tensor ops broadcast over L): with OpenMP one row per thread, which is what
     * bench.py's cpu_baseline leg times on all host cores; the result does not depend on the thread count */
    int err = 0;
#pragma omp parallel for schedule(dynamic, 1)
    for (int j = 0; j < n_rows; ++j) {
        float *z = (float *)malloc((size_t)V * sizeof(float));
        float *scratch = (float *)malloc((size_t)V * sizeof(float));
        if (!z || !scratch) {
#pragma omp atomic write
            err = -1;
            free(z);
            free(scratch);
            continue;
        }
        const float *c = logits_c + (size_t)j * V;
        if (logits_u) {
            const float *u = logits_u + (size_t)j * V;
            for (int i = 0; i < V; ++i) {                  /* JL:104: g*(c-u)+u, three roundings */
                float t = c[i] - u[i];
                t = guidance * t;
                z[i] = t + u[i];
            }
        } else {
            memcpy(z, c, (size_t)V * sizeof(float));
        }
        apply_rule(z, V, &rules[j], scratch);
        float *p = probs_out + (size_t)j * V;
        softmax_canonical(z, V, p);                        /* JL:111 */
        tokens_out[j] = argmax_ratio(p, noise + (size_t)j * V, V);   /* JL:118 */
        free(z);
        free(scratch);
    }
    return err;
}

#ifdef _OPENMP
#include <omp.h>
int sjd_o_max_threads(void) { return omp_get_max_threads(); }
void sjd_o_set_threads(int n) { if (n > 0) omp_set_num_threads(n); }
#else
int sjd_o_max_threads(void) { return 1; }
void sjd_o_set_threads(int n) { (void)n; }
#endif

/* ---------------------------------------------------------------- K4 restatement */
/* Window of n tokens.  win_tok[n]: window ids (win_tok[0] = last accepted token).
 * tokens[n] (in/out): sampled tokens Y, corrected in place (JL:288,307).
 * p: [n, V] target rows.  q_rows[i]: pointer to the draft row for window position i, or NULL when the
 * draft was a fresh random token whose distribution is one-hot(win_tok[i]) (JL:511-514).
 * rs: [n, V] uniforms (JL:260); only rs[i, win_tok[i]] is read (JL:282).
 * resid_rules[i-1] is the rule of the residual call when rejection happens at i (JL:297-306).
 * noise2: [V] Exp(1) for the residual multinomial (JL:237).
 * Returns first_misaligned in [1, n]; *rejected = 1 when a residual resample happened. */
int sjd_o_verify_accept(int n, int V, const int64_t *win_tok, int64_t *tokens, const float *p,
                        const float *const *q_rows, const float *rs, const sjd_row_rule *resid_rules,
                        const float *noise2, int *rejected)
{
    *rejected = 0;
    for (int i = 1; i < n; ++i) {
        int64_t x = win_tok[i];
        float pa = p[(size_t)(i - 1) * V + x];
        float qd = q_rows[i] ? q_rows[i][x] : 1.0f;
        float ratio = pa / qd;                              /* JL:286; NaN compares false => reject */
        float u = rs[(size_t)i * V + x];
        if (u < (ratio > 1.0f ? 1.0f : ratio)) {
            tokens[i - 1] = x;                              /* JL:288 */
            continue;
        }
        /* first rejection: resample position i-1 from norm(max(p - q, 0)) (JL:203-241) */
        float *d = (float *)malloc((size_t)V * sizeof(float));
        float *scratch = (float *)malloc((size_t)V * sizeof(float));
        const float *prow = p + (size_t)(i - 1) * V;
        for (int c = 0; c < V; ++c) {
            float qv = q_rows[i] ? q_rows[i][c] : (c == x ? 1.0f : 0.0f);
            float dv = prow[c] - qv;
            d[c] = dv > 0.0f ? dv : 0.0f;
        }
        const sjd_row_rule *r = &resid_rules[i - 1];
        if (r->forced >= 0) {
            tokens[i - 1] = r->forced;                      /* softmax of a one-hot logit row */
        } else {
            for (int c = 0; c < V; ++c) if (!col_allowed(r, c)) d[c] = 0.0f;
            if (r->top_k > 0 && r->top_k < V) {             /* top-k on log d  <=>  top-k on d */
                float kth = kth_largest(d, V, r->top_k, scratch);
                if (kth > 0.0f)
                    for (int c = 0; c < V; ++c) if (d[c] < kth) d[c] = 0.0f;
            }
            float S = sjd_canonical_sum(d, V);
            if (r->temperature > 0.0f && r->temperature != 1.0f && S > 0.0f) {
                /* the warper list of the residual call holds the TemperatureLogitsWarper too (JL:222-228): weights
                 * exp(log(d) / T - max) instead of d */
                float dm = 0.0f;
                for (int c = 0; c < V; ++c) if (d[c] > dm) dm = d[c];
                const float lm = sjd_logf(dm) / r->temperature;
                for (int c = 0; c < V; ++c) d[c] = d[c] > 0.0f ? sjd_expf(sjd_logf(d[c]) / r->temperature - lm) : 0.0f;
                S = sjd_canonical_sum(d, V);
            }
            if (r->top_p_thr >= 0.0f) {                     /* top-p on softmax(log d) = d/S, LP:406-419 */
                float dm = 0.0f;
                int imax = 0;
                for (int c = 0; c < V; ++c) if (d[c] > dm) { dm = d[c]; imax = c; }
                uint32_t K = top_p_cut_key(d, d, S, V, r->top_p_thr, scratch);
                for (int c = 0; c < V; ++c) if (c != imax && f2key(d[c]) <= K) d[c] = 0.0f;
                S = sjd_canonical_sum(d, V);
            }
            for (int c = 0; c < V; ++c) scratch[c] = d[c] / S;
            tokens[i - 1] = argmax_ratio(scratch, noise2, V);
        }
        free(d);
        free(scratch);
        *rejected = 1;
        return i;
    }
    return n;
}

/* plain Jacobi decoding: first i with win_tok[i] != tokens[i-1] (JL:317-333) */
int sjd_o_first_mismatch(int n, const int64_t *win_tok, const int64_t *tokens)
{
    for (int i = 1; i < n; ++i)
        if (win_tok[i] != tokens[i - 1]) return i;
    return n;
}

/* exported so tests can pin the canonical primitives themselves */
float sjd_o_expf(float x) { return sjd_expf(x); }
float sjd_o_logf(float x) { return sjd_logf(x); }
float sjd_o_sum(const float *v, int V) { return sjd_canonical_sum(v, V); }
