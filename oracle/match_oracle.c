/*
 * match_oracle.c -- CPU restatement of the descriptor-matching stage.
 *
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see oracle.h).
 *
 * Reference call sites: Source Files/feature_processing.cpp:101-130 (matcher
 * construction, ratioMaches) and :144-176 (findRobustMatches up to and
 * including the cross-check).  The k-nearest search itself is OpenCV's
 * BFMatcher::knnMatchImpl -> cv::batchDistance (modules/core/src/
 * batch_distance.cpp), restated in knn2_insert() below: candidates are visited
 * in ascending train index, a candidate enters the list only when it is
 * strictly smaller than the current last entry, and it is placed behind every
 * entry that is <= it.  Equal distances therefore keep the lower train index
 * first, for both neighbours.
 */
#include "oracle.h"

#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static int resolve_threads(int threads) {
#ifdef _OPENMP
    if (threads <= 0) return omp_get_max_threads();
    return threads;
#else
    (void)threads;
    return 1;
#endif
}

/* batch_distance.cpp, the K-insertion loop of batchDistance for one query. */
static inline void knn2_insert_i(int d, int j, int* dist, int* idx) {
    if (d < dist[1]) {
        int k = 0;
        if (dist[0] > d) {
            dist[1] = dist[0];
            idx[1] = idx[0];
            k = -1;
        }
        dist[k + 1] = d;
        idx[k + 1] = j;
    }
}

static inline void knn2_insert_f(float d, int j, float* dist, int* idx) {
    if (d < dist[1]) {
        int k = 0;
        if (dist[0] > d) {
            dist[1] = dist[0];
            idx[1] = idx[0];
            k = -1;
        }
        dist[k + 1] = d;
        idx[k + 1] = j;
    }
}

static inline int hamming_row(const uint8_t* a, const uint8_t* b, int n) {
    int d = 0, i = 0;
    for (; i + 8 <= n; i += 8) {
        uint64_t x, y;
        memcpy(&x, a + i, 8);
        memcpy(&y, b + i, 8);
        d += __builtin_popcountll(x ^ y);
    }
    for (; i < n; ++i) d += __builtin_popcount((unsigned)(a[i] ^ b[i]));
    return d;
}

void orc_knn2_hamming(const uint8_t* q, int nq, size_t q_stride,
                      const uint8_t* t, int nt, size_t t_stride, int dbytes,
                      int32_t* idx, float* dist, int threads) {
    int nthr = resolve_threads(threads);
    (void)nthr;
#pragma omp parallel for schedule(static) num_threads(nthr)
    for (int i = 0; i < nq; ++i) {
        int bd[2] = {INT_MAX, INT_MAX};
        int bi[2] = {-1, -1};
        const uint8_t* qi = q + (size_t)i * q_stride;
        for (int j = 0; j < nt; ++j) {
            int d = hamming_row(qi, t + (size_t)j * t_stride, dbytes);
            knn2_insert_i(d, j, bd, bi);
        }
        for (int k = 0; k < 2; ++k) {
            idx[2 * i + k] = bi[k];
            /* matchers.cpp converts the int32 distance to float in DMatch */
            dist[2 * i + k] = bi[k] >= 0 ? (float)bd[k] : FLT_MAX;
        }
    }
}

/* stat.cpp normL2Sqr, the scalar (no-SIMD) accumulation order. */
static inline float l2sqr_row(const float* a, const float* b, int n) {
    float d = 0.f;
    for (int j = 0; j < n; ++j) {
        float v = a[j] - b[j];
        d += v * v;
    }
    return d;
}

void orc_knn2_l2(const float* q, int nq, size_t q_stride,
                 const float* t, int nt, size_t t_stride, int dim, int squared,
                 int32_t* idx, float* dist, int threads) {
    int nthr = resolve_threads(threads);
    (void)nthr;
#pragma omp parallel for schedule(static) num_threads(nthr)
    for (int i = 0; i < nq; ++i) {
        float bd[2] = {FLT_MAX, FLT_MAX};
        int bi[2] = {-1, -1};
        const float* qi = q + (size_t)i * q_stride;
        for (int j = 0; j < nt; ++j) {
            float d = l2sqr_row(qi, t + (size_t)j * t_stride, dim);
            if (!squared) d = sqrtf(d); /* batchDistL2_32f */
            knn2_insert_f(d, j, bd, bi);
        }
        for (int k = 0; k < 2; ++k) {
            idx[2 * i + k] = bi[k];
            dist[2 * i + k] = bi[k] >= 0 ? bd[k] : FLT_MAX;
        }
    }
}

/* feature_processing.cpp:123-129.  The reference reads k[1] even when the
 * train set has fewer than two rows (undefined behaviour); here such a query
 * yields no match. */
int orc_ratio_filter(const int32_t* idx, const float* dist, int nq, int nt,
                     float ratio, orc_dmatch* out) {
    int n = 0;
    if (nt < 2) return 0;
    for (int i = 0; i < nq; ++i) {
        float d0 = dist[2 * i], d1 = dist[2 * i + 1];
        if (d0 < ratio * d1) {
            out[n].queryIdx = i;
            out[n].trainIdx = idx[2 * i];
            out[n].imgIdx = 0;
            out[n].distance = d0;
            ++n;
        }
    }
    return n;
}

/* feature_processing.cpp:159-176: output follows the order of b. */
int orc_cross_check(const orc_dmatch* f, int nf, const orc_dmatch* b, int nb,
                    orc_dmatch* out) {
    int n = 0;
    for (int ib = 0; ib < nb; ++ib) {
        for (int jf = 0; jf < nf; ++jf) {
            if (b[ib].queryIdx == f[jf].trainIdx &&
                b[ib].trainIdx == f[jf].queryIdx) {
                out[n++] = f[jf];
                break;
            }
        }
    }
    return n;
}

int orc_robust_matches(int norm, const void* prev, int n_prev,
                       const void* curr, int n_curr, int d, float ratio,
                       orc_dmatch* out, int threads) {
    int nmax = n_prev > n_curr ? n_prev : n_curr;
    if (nmax <= 0) return 0;
    int32_t* idx = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)nmax);
    float* dist = (float*)malloc(sizeof(float) * 2 * (size_t)nmax);
    orc_dmatch* f = (orc_dmatch*)malloc(sizeof(orc_dmatch) * (size_t)nmax);
    orc_dmatch* b = (orc_dmatch*)malloc(sizeof(orc_dmatch) * (size_t)nmax);
    int nf, nb, n;

    if (norm == ORC_NORM_HAMMING)
        orc_knn2_hamming((const uint8_t*)prev, n_prev, (size_t)d,
                         (const uint8_t*)curr, n_curr, (size_t)d, d, idx, dist,
                         threads);
    else
        orc_knn2_l2((const float*)prev, n_prev, (size_t)d, (const float*)curr,
                    n_curr, (size_t)d, d, norm == ORC_NORM_L2SQR, idx, dist,
                    threads);
    nf = orc_ratio_filter(idx, dist, n_prev, n_curr, ratio, f);

    if (norm == ORC_NORM_HAMMING)
        orc_knn2_hamming((const uint8_t*)curr, n_curr, (size_t)d,
                         (const uint8_t*)prev, n_prev, (size_t)d, d, idx, dist,
                         threads);
    else
        orc_knn2_l2((const float*)curr, n_curr, (size_t)d, (const float*)prev,
                    n_prev, (size_t)d, d, norm == ORC_NORM_L2SQR, idx, dist,
                    threads);
    nb = orc_ratio_filter(idx, dist, n_curr, n_prev, ratio, b);

    /* The reference loop is O(nf*nb); a query index appears at most once in f,
     * so a lookup table gives the same result in O(nf+nb). */
    {
        int32_t* f_of_prev = (int32_t*)malloc(sizeof(int32_t) * (size_t)nmax);
        for (int i = 0; i < nmax; ++i) f_of_prev[i] = -1;
        for (int j = 0; j < nf; ++j) f_of_prev[f[j].queryIdx] = j;
        n = 0;
        for (int ib = 0; ib < nb; ++ib) {
            int p = b[ib].trainIdx;
            int j = (p >= 0 && p < nmax) ? f_of_prev[p] : -1;
            if (j >= 0 && f[j].trainIdx == b[ib].queryIdx) out[n++] = f[j];
        }
        free(f_of_prev);
    }
    free(idx);
    free(dist);
    free(f);
    free(b);
    return n;
}
