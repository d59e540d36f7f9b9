/*
 * TEST INFRASTRUCTURE -- not part of the product path.
 *
 * Plain-C restatement of the reference's kNN site (faiss.IndexFlatL2.search as called at
 * /root/reference/src/adaptive_classifier/memory.py:113-114): exact squared-L2 brute force,
 * k smallest per query, ascending, ties to the lower row id.  Same definition as
 * oracle/knn_oracle.py (fp64 accumulation of (p-q)^2, result rounded once to fp32), fast
 * enough (OpenMP over row blocks) to check million-row cases and to serve as the
 * `cpu_baseline` "port" leg of bench.py.  faiss itself is absent => parity vs real faiss
 * is "parity unpinned"; see the header of knn_oracle.py.
 *
 * Build: make -C oracle   (gcc -O3 -fopenmp -shared -fPIC)
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

/* sorted insertion into an ascending list of at most k entries */
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

void oracle_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* outD [nq,k] float, outI [nq,k] int64; pads with (FLT_MAX,-1) when k > N. */
int oracle_knn_l2_topk(const float* P, int64_t N, int64_t ldP, int D,
                       const float* Q, int nq, int64_t ldQ, int k, int64_t row_offset,
                       float* outD, int64_t* outI) {
    if (k <= 0 || D <= 0 || nq < 0 || N < 0) return -1;
    int nt = oracle_num_threads();
    ent_t* lists = (ent_t*)malloc((size_t)nt * k * sizeof(ent_t));
    int* counts = (int*)malloc((size_t)nt * sizeof(int));
    double* q64 = (double*)malloc((size_t)D * sizeof(double));
    if (!lists || !counts || !q64) return -2;
    for (int q = 0; q < nq; ++q) {
        for (int c = 0; c < D; ++c) q64[c] = (double)Q[(size_t)q * ldQ + c];
        for (int t = 0; t < nt; ++t) counts[t] = 0;
#pragma omp parallel
        {
#ifdef _OPENMP
            int t = omp_get_thread_num();
#else
            int t = 0;
#endif
            ent_t* L = lists + (size_t)t * k;
            int n = 0;
#pragma omp for schedule(static)
            for (int64_t r = 0; r < N; ++r) {
                const float* p = P + (size_t)r * ldP;
                double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
                int c = 0;
                for (; c + 3 < D; c += 4) {
                    double e0 = (double)p[c] - q64[c], e1 = (double)p[c + 1] - q64[c + 1];
                    double e2 = (double)p[c + 2] - q64[c + 2], e3 = (double)p[c + 3] - q64[c + 3];
                    s0 += e0 * e0; s1 += e1 * e1; s2 += e2 * e2; s3 += e3 * e3;
                }
                for (; c < D; ++c) { double e = (double)p[c] - q64[c]; s0 += e * e; }
                topk_insert(L, &n, k, (s0 + s1) + (s2 + s3), r);
            }
            counts[t] = n;
        }
        /* merge the per-thread lists */
        ent_t* M = (ent_t*)malloc((size_t)k * sizeof(ent_t));
        int m = 0;
        for (int t = 0; t < nt; ++t)
            for (int j = 0; j < counts[t]; ++j)
                topk_insert(M, &m, k, lists[(size_t)t * k + j].d, lists[(size_t)t * k + j].i);
        for (int j = 0; j < k; ++j) {
            if (j < m) { outD[(size_t)q * k + j] = (float)M[j].d; outI[(size_t)q * k + j] = M[j].i + row_offset; }
            else       { outD[(size_t)q * k + j] = FLT_MAX;       outI[(size_t)q * k + j] = -1; }
        }
        free(M);
    }
    free(lists); free(counts); free(q64);
    return 0;
}

/*
 * What the reference's CPU path actually executes per query for nq < 20 (faiss's non-BLAS
 * path): fp32 sum of (x-y)^2, max-heap of size k.  Used only as the timed CPU baseline
 * ("port" of the fp32 path); not used for parity because fp32 ordering of near-ties is
 * implementation-defined.
 */
int oracle_knn_l2_topk_f32(const float* P, int64_t N, int64_t ldP, int D,
                           const float* Q, int nq, int64_t ldQ, int k,
                           float* outD, int64_t* outI) {
    if (k <= 0 || D <= 0) return -1;
    int nt = oracle_num_threads();
    ent_t* lists = (ent_t*)malloc((size_t)nt * k * sizeof(ent_t));
    int* counts = (int*)malloc((size_t)nt * sizeof(int));
    if (!lists || !counts) return -2;
    for (int q = 0; q < nq; ++q) {
        const float* qv = Q + (size_t)q * ldQ;
#pragma omp parallel
        {
#ifdef _OPENMP
            int t = omp_get_thread_num();
#else
            int t = 0;
#endif
            ent_t* L = lists + (size_t)t * k;
            int n = 0;
#pragma omp for schedule(static)
            for (int64_t r = 0; r < N; ++r) {
                const float* p = P + (size_t)r * ldP;
                float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                int c = 0;
                for (; c + 7 < D; c += 8)
                    for (int u = 0; u < 8; ++u) { float e = p[c + u] - qv[c + u]; s[u] += e * e; }
                for (; c < D; ++c) { float e = p[c] - qv[c]; s[0] += e * e; }
                float tot = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
                topk_insert(L, &n, k, (double)tot, r);
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
    free(lists); free(counts);
    return 0;
}
