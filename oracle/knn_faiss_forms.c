/*
 * TEST INFRASTRUCTURE -- not part of the product path.
 *
 * SECOND ORACLES for the kNN site (faiss.IndexFlatL2.search, called at
 * /root/reference/src/adaptive_classifier/memory.py:113-114): fp32 restatements of the two code paths
 * faiss-cpu (>=1.7.4, requirements.txt:4; third party, NOT under /root/reference, not installable here)
 * takes for a flat L2 search, as published in faiss/utils/distances.cpp + distances_simd.cpp:
 *
 *   nq <  20  (distance_compute_blas_threshold): exhaustive_L2sqr_seq -- per (query, row)
 *             fvec_L2sqr = sum_i (x_i - y_i)^2 in fp32.  The summation order is compiler / ISA defined
 *             (the 1.7.4 source is a plain loop under an "imprecise" pragma that the compiler vectorises;
 *             older releases hand-wrote the AVX2 form).  Restated here as a FAMILY: `lanes` partial sums
 *             (1 = scalar sequential, 8 = AVX2, 16 = AVX-512, 32 = 2x unrolled AVX-512), combined by the
 *             AVX2 extract/hadd tree for 8 lanes and by a halving tree otherwise, with or without fma
 *             contraction.
 *   nq >= 20: exhaustive_L2sqr_blas -- dis = |x|^2 + |y|^2 - 2 <x, y>, the inner products from sgemm, norms
 *             from fvec_norm_L2sqr, negative results clamped to 0.  sgemm's accumulation order is
 *             BLAS-implementation defined; restated with the same (lanes, fma) family for the dot product.
 *
 * None of these IS faiss (its exact rounding cannot be reproduced without the binary), which is why kNN
 * parity vs real faiss stays "parity unpinned".  What they give: every member of the family rounds each
 * distance within the classical a-priori bound E of the exact value, so (order statistics) the j-th id any
 * such implementation returns must satisfy |d_exact(id_j) - d_exact_sorted[j]| <= 2E.  tests/test_knn_forms*.py
 * count the id disagreements of every form with the exact oracle and assert each one is such a near-tie --
 * i.e. the exact-definition ids the HIP path returns are what faiss returns except inside fp32 near-ties.
 *
 * Also here: oracle_knn_l2_topk_batch, the exact fp64 oracle of knn_oracle.c restructured row-outer /
 * query-inner (each row is read once for all queries) so that 64-query subsets of the 10M x 768 BASELINE
 * store finish in seconds.  Same arithmetic and tie rule as oracle_knn_l2_topk.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct { double d; int64_t i; } ent_t;

static inline int ent_less(double d1, int64_t i1, double d2, int64_t i2) {
    return d1 < d2 || (d1 == d2 && i1 < i2);
}

static inline void topk_insert(ent_t* L, int* n, int k, double d, int64_t i) {
    int m = *n;
    if (m == k) {
        if (!ent_less(d, i, L[k - 1].d, L[k - 1].i)) return;
        m = k - 1;
    }
    int p = m;
    while (p > 0 && ent_less(d, i, L[p - 1].d, L[p - 1].i)) { L[p] = L[p - 1]; --p; }
    L[p].d = d; L[p].i = i;
    *n = m + 1;
}

static int n_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ---- the fp32 reduction family ------------------------------------------------------------- */
#define MAXL 32

/* combine `lanes` partial sums the way the ISA form does */
static inline float combine(const float* s, int lanes) {
    if (lanes == 1) return s[0];
    if (lanes == 8) {
        /* AVX2: msum2 = hi128 + lo128; hadd; hadd  (faiss distances_simd.cpp, pre-1.7.3 hand-written form) */
        const float a0 = s[4] + s[0], a1 = s[5] + s[1], a2 = s[6] + s[2], a3 = s[7] + s[3];
        return (a0 + a1) + (a2 + a3);
    }
    float t[MAXL];
    for (int i = 0; i < lanes; ++i) t[i] = s[i];
    for (int h = lanes / 2; h >= 1; h /= 2)
        for (int i = 0; i < h; ++i) t[i] = t[i] + t[i + h];
    return t[0];
}

/* sum_i (x_i - y_i)^2 */
static inline float l2sqr_form(const float* x, const float* y, int d, int lanes, int use_fma) {
    float s[MAXL];
    for (int l = 0; l < lanes; ++l) s[l] = 0.f;
    int i = 0;
    for (; i + lanes <= d; i += lanes)
        for (int l = 0; l < lanes; ++l) {
            const float t = x[i + l] - y[i + l];
            s[l] = use_fma ? fmaf(t, t, s[l]) : s[l] + t * t;
        }
    for (int l = 0; i < d; ++i, ++l) {          /* remainder: masked lanes */
        const float t = x[i] - y[i];
        s[l] = use_fma ? fmaf(t, t, s[l]) : s[l] + t * t;
    }
    return combine(s, lanes);
}

/* <x, y> */
static inline float dot_form(const float* x, const float* y, int d, int lanes, int use_fma) {
    float s[MAXL];
    for (int l = 0; l < lanes; ++l) s[l] = 0.f;
    int i = 0;
    for (; i + lanes <= d; i += lanes)
        for (int l = 0; l < lanes; ++l) s[l] = use_fma ? fmaf(x[i + l], y[i + l], s[l]) : s[l] + x[i + l] * y[i + l];
    for (int l = 0; i < d; ++i, ++l) s[l] = use_fma ? fmaf(x[i], y[i], s[l]) : s[l] + x[i] * y[i];
    return combine(s, lanes);
}

/*
 * form: 0 = difference form (exhaustive_L2sqr_seq), 1 = norm/dot form (exhaustive_L2sqr_blas).
 * lanes in {1, 8, 16, 32}; use_fma 0/1.  Selection: k smallest by (fp32 value, id) -- faiss keeps a max-heap
 * and on equal values keeps the earlier (lower) id.  outD fp32 [nq,k], outI int64 [nq,k]; (FLT_MAX, -1) pads.
 */
int oracle_knn_form_topk(const float* P, int64_t N, int64_t ldP, int D, const float* Q, int nq, int64_t ldQ,
                         int k, int form, int lanes, int use_fma, float* outD, int64_t* outI) {
    if (k <= 0 || D <= 0 || nq < 0 || N < 0) return -1;
    if (!(lanes == 1 || lanes == 8 || lanes == 16 || lanes == 32) || form < 0 || form > 1) return -1;
    const int nt = n_threads();
    ent_t* lists = (ent_t*)malloc((size_t)nt * k * sizeof(ent_t));
    int* counts = (int*)malloc((size_t)nt * sizeof(int));
    float* pn = NULL;
    if (!lists || !counts) return -2;
    if (form == 1) {
        pn = (float*)malloc((size_t)(N > 0 ? N : 1) * sizeof(float));
        if (!pn) return -2;
#pragma omp parallel for schedule(static)
        for (int64_t r = 0; r < N; ++r) pn[r] = dot_form(P + (size_t)r * ldP, P + (size_t)r * ldP, D, lanes, use_fma);
    }
    for (int q = 0; q < nq; ++q) {
        const float* qv = Q + (size_t)q * ldQ;
        const float qn = form == 1 ? dot_form(qv, qv, D, lanes, use_fma) : 0.f;
        for (int t = 0; t < nt; ++t) counts[t] = 0;
#pragma omp parallel
        {
#ifdef _OPENMP
            const int t = omp_get_thread_num();
#else
            const int t = 0;
#endif
            ent_t* L = lists + (size_t)t * k;
            int n = 0;
#pragma omp for schedule(static)
            for (int64_t r = 0; r < N; ++r) {
                const float* p = P + (size_t)r * ldP;
                float dis;
                if (form == 0) {
                    dis = l2sqr_form(qv, p, D, lanes, use_fma);
                } else {
                    const float ip = dot_form(qv, p, D, lanes, use_fma);
                    dis = qn + pn[r] - 2.f * ip;            /* distances.cpp: x_norms[i] + y_norms[j] - 2 * ip */
                    if (dis < 0.f) dis = 0.f;
                }
                topk_insert(L, &n, k, (double)dis, r);
            }
            counts[t] = n;
        }
        ent_t* M = (ent_t*)malloc((size_t)k * sizeof(ent_t));
        int m = 0;
        for (int t = 0; t < nt; ++t)
            for (int j = 0; j < counts[t]; ++j)
                topk_insert(M, &m, k, lists[(size_t)t * k + j].d, lists[(size_t)t * k + j].i);
        for (int j = 0; j < k; ++j) {
            if (j < m) { outD[(size_t)q * k + j] = (float)M[j].d; outI[(size_t)q * k + j] = M[j].i; }
            else       { outD[(size_t)q * k + j] = FLT_MAX;       outI[(size_t)q * k + j] = -1; }
        }
        free(M);
    }
    free(lists); free(counts); free(pn);
    return 0;
}

/* exact fp64 squared distances of given (query, row id) pairs: out[q][j] = d(Q[q], P[ids[q][j]]); id < 0 -> +inf */
int oracle_exact_dist_of_ids(const float* P, int64_t N, int64_t ldP, int D, const float* Q, int nq, int64_t ldQ,
                             const int64_t* ids, int k, double* out) {
    for (int q = 0; q < nq; ++q)
        for (int j = 0; j < k; ++j) {
            const int64_t r = ids[(size_t)q * k + j];
            if (r < 0 || r >= N) { out[(size_t)q * k + j] = INFINITY; continue; }
            const float* p = P + (size_t)r * ldP;
            const float* qv = Q + (size_t)q * ldQ;
            double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
            int c = 0;
            for (; c + 3 < D; c += 4) {
                double e0 = (double)p[c] - (double)qv[c], e1 = (double)p[c + 1] - (double)qv[c + 1];
                double e2 = (double)p[c + 2] - (double)qv[c + 2], e3 = (double)p[c + 3] - (double)qv[c + 3];
                s0 += e0 * e0; s1 += e1 * e1; s2 += e2 * e2; s3 += e3 * e3;
            }
            for (; c < D; ++c) { double e = (double)p[c] - (double)qv[c]; s0 += e * e; }
            out[(size_t)q * k + j] = (s0 + s1) + (s2 + s3);
        }
    return 0;
}

/*
 * Exact oracle, batched: identical definition / arithmetic to oracle_knn_l2_topk (knn_oracle.c): fp64 sum of
 * (p - q)^2 over 4 interleaved accumulators, order by (d, id), fp32-rounded distance out.  Row-outer loop:
 * every thread walks its rows once and updates all nq per-query lists.
 */
int oracle_knn_l2_topk_batch(const float* P, int64_t N, int64_t ldP, int D, const float* Q, int nq, int64_t ldQ,
                             int k, int64_t row_offset, float* outD, int64_t* outI) {
    if (k <= 0 || D <= 0 || nq < 0 || N < 0) return -1;
    const int nt = n_threads();
    const int Dp = (D + 3) / 4 * 4;
    double* q64 = (double*)calloc((size_t)(nq > 0 ? nq : 1) * Dp, sizeof(double));
    ent_t* lists = (ent_t*)malloc((size_t)nt * (nq > 0 ? nq : 1) * k * sizeof(ent_t));
    int* counts = (int*)calloc((size_t)nt * (nq > 0 ? nq : 1), sizeof(int));
    if (!q64 || !lists || !counts) return -2;
    for (int q = 0; q < nq; ++q)
        for (int c = 0; c < D; ++c) q64[(size_t)q * Dp + c] = (double)Q[(size_t)q * ldQ + c];
#pragma omp parallel
    {
#ifdef _OPENMP
        const int t = omp_get_thread_num();
#else
        const int t = 0;
#endif
        double* p64 = (double*)calloc((size_t)Dp, sizeof(double));
        ent_t* L = lists + (size_t)t * nq * k;
        int* cnt = counts + (size_t)t * nq;
#pragma omp for schedule(static)
        for (int64_t r = 0; r < N; ++r) {
            const float* p = P + (size_t)r * ldP;
            for (int c = 0; c < D; ++c) p64[c] = (double)p[c];
            for (int q = 0; q < nq; ++q) {
                const double* qq = q64 + (size_t)q * Dp;
                double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
                int c = 0;
                for (; c + 3 < D; c += 4) {
                    double e0 = p64[c] - qq[c], e1 = p64[c + 1] - qq[c + 1];
                    double e2 = p64[c + 2] - qq[c + 2], e3 = p64[c + 3] - qq[c + 3];
                    s0 += e0 * e0; s1 += e1 * e1; s2 += e2 * e2; s3 += e3 * e3;
                }
                for (; c < D; ++c) { double e = p64[c] - qq[c]; s0 += e * e; }
                const double d = (s0 + s1) + (s2 + s3);
                int n = cnt[q];
                if (n < k || ent_less(d, r, L[(size_t)q * k + k - 1].d, L[(size_t)q * k + k - 1].i)) {
                    topk_insert(L + (size_t)q * k, &n, k, d, r);
                    cnt[q] = n;
                }
            }
        }
        free(p64);
    }
    ent_t* M = (ent_t*)malloc((size_t)k * sizeof(ent_t));
    for (int q = 0; q < nq; ++q) {
        int m = 0;
        for (int t = 0; t < nt; ++t) {
            const ent_t* L = lists + ((size_t)t * nq + q) * k;
            const int n = counts[(size_t)t * nq + q];
            for (int j = 0; j < n; ++j) topk_insert(M, &m, k, L[j].d, L[j].i);
        }
        for (int j = 0; j < k; ++j) {
            if (j < m) { outD[(size_t)q * k + j] = (float)M[j].d; outI[(size_t)q * k + j] = M[j].i + row_offset; }
            else       { outD[(size_t)q * k + j] = FLT_MAX;       outI[(size_t)q * k + j] = -1; }
        }
    }
    free(M); free(q64); free(lists); free(counts);
    return 0;
}
