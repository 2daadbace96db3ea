/*
 * fundamental_oracle.c -- CPU restatement of cv::findFundamentalMat as the reference calls
 * it in the epipolar filter of DescriptorMatcher::findRobustMatches
 * (Source Files/feature_processing.cpp:187-188):
 *
 *     std::vector<uint8_t> inliersMask(matches.size());
 *     cv::findFundamentalMat(prevAligPts, currAligPts, inliersMask);
 *
 * i.e. method FM_RANSAC, ransacReprojThreshold 3, confidence 0.99, maxIters 1000.
 *
 * TEST INFRASTRUCTURE ONLY -- PARITY UNPINNED (see oracle.h).  Restated from the published
 * OpenCV 4.x sources: modules/calib3d/src/fundam.cpp (findFundamentalMat, run7Point,
 * FMEstimatorCallback::computeError / checkSubset), modules/calib3d/src/ptsetreg.cpp
 * (RANSACPointSetRegistrator::run, LMeDSPointSetRegistrator::run, getSubset,
 * RANSACUpdateNumIters), modules/core/src/mathfuncs.cpp (solveCubic) and cv::RNG
 * (modules/core/include/opencv2/core/operations.hpp).  One deliberate difference: OpenCV takes
 * the two-dimensional null space of the 7x9 system from the last two rows of a full SVD,
 * whose choice of basis inside that plane depends on its Jacobi sweeps and on a random
 * completion; here the basis comes from Gauss-Jordan elimination with complete pivoting
 * followed by Gram-Schmidt.  The fundamental matrices (the roots of the cubic along the
 * pencil) do not depend on the basis beyond rounding.
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

/* ---- cv::RNG ------------------------------------------------------------ */

typedef struct {
    uint64_t state;
} orc_rng;

static unsigned rng_next(orc_rng* r) {
    r->state = (uint64_t)(unsigned)r->state * 4164903690U + (unsigned)(r->state >> 32);
    return (unsigned)r->state;
}

/* RNG::uniform(int a, int b) */
static int rng_uniform(orc_rng* r, int a, int b) { return a == b ? a : (int)(rng_next(r) % (unsigned)(b - a) + a); }

/* ---- solveCubic --------------------------------------------------------- */

static int solve_cubic(const double c[4], double roots[3]) {
    double a0 = c[0], a1 = c[1], a2 = c[2], a3 = c[3];
    double x0 = 0., x1 = 0., x2 = 0.;
    int n = 0;
    if (a0 == 0) {
        if (a1 == 0) {
            if (a2 == 0)
                n = a3 == 0 ? -1 : 0;
            else {
                x0 = -a3 / a2;
                n = 1;
            }
        } else {
            double d = a2 * a2 - 4 * a1 * a3;
            if (d >= 0) {
                d = sqrt(d);
                double q1 = (-a2 + d) * 0.5;
                double q2 = (a2 + d) * -0.5;
                if (fabs(q1) > fabs(q2)) {
                    x0 = q1 / a1;
                    x1 = a3 / q1;
                } else {
                    x0 = q2 / a1;
                    x1 = a3 / q2;
                }
                n = d > 0 ? 2 : 1;
            }
        }
    } else {
        a0 = 1. / a0;
        a1 *= a0;
        a2 *= a0;
        a3 *= a0;
        double Q = (a1 * a1 - 3 * a2) * (1. / 9);
        double R = (2 * a1 * a1 * a1 - 9 * a1 * a2 + 27 * a3) * (1. / 54);
        double Qcubed = Q * Q * Q;
        double d = Qcubed - R * R;
        if (d > 0) {
            double theta = acos(R / sqrt(Qcubed));
            double sqrtQ = sqrt(Q);
            double t0 = -2 * sqrtQ;
            double t1 = theta * (1. / 3);
            double t2 = a1 * (1. / 3);
            x0 = t0 * cos(t1) - t2;
            x1 = t0 * cos(t1 + (2. * 3.1415926535897932384626433832795 / 3)) - t2;
            x2 = t0 * cos(t1 + (4. * 3.1415926535897932384626433832795 / 3)) - t2;
            n = 3;
        } else if (d == 0) {
            if (R >= 0) {
                x0 = -2 * pow(R, 1. / 3) - a1 / 3;
                x1 = pow(R, 1. / 3) - a1 / 3;
            } else {
                x0 = 2 * pow(-R, 1. / 3) - a1 / 3;
                x1 = -pow(-R, 1. / 3) - a1 / 3;
            }
            x2 = 0;
            n = x0 == x1 ? 1 : 2;
            x1 = x0 == x1 ? 0 : x1;
        } else {
            double e;
            d = sqrt(-d);
            e = pow(d + fabs(R), 1. / 3);
            if (R > 0) e = -e;
            x0 = (e + Q / e) - a1 * (1. / 3);
            n = 1;
        }
    }
    roots[0] = x0;
    roots[1] = x1;
    roots[2] = x2;
    return n;
}

/* ---- null space of the 7x9 system --------------------------------------- */

/* Orthonormal basis (f1, f2) of the null space of the 7x9 matrix a (row-major, destroyed):
 * Gauss-Jordan with complete pivoting, the two columns left without a pivot are the free
 * variables; the lower-numbered free column gives f1, the other f2; then Gram-Schmidt with
 * f2 first.  Returns 0 when the system has rank below 7. */
static int null_space_7x9(double a[63], double f1[9], double f2[9]) {
    int col_of[7];
    int used_col[9] = {0};
    for (int k = 0; k < 7; ++k) {
        int pr = -1, pc = -1;
        double best = 0;
        for (int r = k; r < 7; ++r)
            for (int c = 0; c < 9; ++c) {
                if (used_col[c]) continue;
                const double v = fabs(a[r * 9 + c]);
                if (v > best) {
                    best = v;
                    pr = r;
                    pc = c;
                }
            }
        if (pr < 0) return 0;
        if (pr != k)
            for (int c = 0; c < 9; ++c) {
                const double t = a[k * 9 + c];
                a[k * 9 + c] = a[pr * 9 + c];
                a[pr * 9 + c] = t;
            }
        used_col[pc] = 1;
        col_of[k] = pc;
        const double inv = 1.0 / a[k * 9 + pc];
        for (int c = 0; c < 9; ++c) a[k * 9 + c] *= inv;
        for (int r = 0; r < 7; ++r) {
            if (r == k) continue;
            const double m = a[r * 9 + pc];
            if (m == 0) continue;
            for (int c = 0; c < 9; ++c) a[r * 9 + c] -= m * a[k * 9 + c];
        }
    }
    int free_cols[2], nf = 0;
    for (int c = 0; c < 9; ++c)
        if (!used_col[c]) free_cols[nf++] = c;
    double* out[2] = {f1, f2};
    for (int v = 0; v < 2; ++v) {
        double* f = out[v];
        for (int c = 0; c < 9; ++c) f[c] = 0;
        f[free_cols[v]] = 1.0;
        for (int k = 0; k < 7; ++k) f[col_of[k]] = -a[k * 9 + free_cols[v]];
    }
    double n2 = 0;
    for (int c = 0; c < 9; ++c) n2 += f2[c] * f2[c];
    double inv = 1.0 / sqrt(n2);
    for (int c = 0; c < 9; ++c) f2[c] *= inv;
    double dot = 0;
    for (int c = 0; c < 9; ++c) dot += f1[c] * f2[c];
    for (int c = 0; c < 9; ++c) f1[c] -= dot * f2[c];
    n2 = 0;
    for (int c = 0; c < 9; ++c) n2 += f1[c] * f1[c];
    inv = 1.0 / sqrt(n2);
    for (int c = 0; c < 9; ++c) f1[c] *= inv;
    return 1;
}

/* ---- run7Point ---------------------------------------------------------- */

int orc_fm_7point(const float* m1, const float* m2, double* fmatrix /* up to 3 x 9 */) {
    double a[7 * 9], c[4], r[3] = {0, 0, 0};
    double f1[9], f2[9];
    for (int i = 0; i < 7; i++) {
        const double x0 = m1[2 * i], y0 = m1[2 * i + 1];
        const double x1 = m2[2 * i], y1 = m2[2 * i + 1];
        a[i * 9 + 0] = x1 * x0;
        a[i * 9 + 1] = x1 * y0;
        a[i * 9 + 2] = x1;
        a[i * 9 + 3] = y1 * x0;
        a[i * 9 + 4] = y1 * y0;
        a[i * 9 + 5] = y1;
        a[i * 9 + 6] = x0;
        a[i * 9 + 7] = y0;
        a[i * 9 + 8] = 1;
    }
    if (!null_space_7x9(a, f1, f2)) return 0;
    double t0, t1, t2;
    for (int i = 0; i < 9; i++) f1[i] -= f2[i];

    t0 = f2[4] * f2[8] - f2[5] * f2[7];
    t1 = f2[3] * f2[8] - f2[5] * f2[6];
    t2 = f2[3] * f2[7] - f2[4] * f2[6];
    c[3] = f2[0] * t0 - f2[1] * t1 + f2[2] * t2;
    c[2] = f1[0] * t0 - f1[1] * t1 + f1[2] * t2 - f1[3] * (f2[1] * f2[8] - f2[2] * f2[7]) +
           f1[4] * (f2[0] * f2[8] - f2[2] * f2[6]) - f1[5] * (f2[0] * f2[7] - f2[1] * f2[6]) +
           f1[6] * (f2[1] * f2[5] - f2[2] * f2[4]) - f1[7] * (f2[0] * f2[5] - f2[2] * f2[3]) +
           f1[8] * (f2[0] * f2[4] - f2[1] * f2[3]);
    t0 = f1[4] * f1[8] - f1[5] * f1[7];
    t1 = f1[3] * f1[8] - f1[5] * f1[6];
    t2 = f1[3] * f1[7] - f1[4] * f1[6];
    c[1] = f2[0] * t0 - f2[1] * t1 + f2[2] * t2 - f2[3] * (f1[1] * f1[8] - f1[2] * f1[7]) +
           f2[4] * (f1[0] * f1[8] - f1[2] * f1[6]) - f2[5] * (f1[0] * f1[7] - f1[1] * f1[6]) +
           f2[6] * (f1[1] * f1[5] - f1[2] * f1[4]) - f2[7] * (f1[0] * f1[5] - f1[2] * f1[3]) +
           f2[8] * (f1[0] * f1[4] - f1[1] * f1[3]);
    c[0] = f1[0] * t0 - f1[1] * t1 + f1[2] * t2;

    const int n = solve_cubic(c, r);
    if (n < 1 || n > 3) return n;
    for (int k = 0; k < n; k++, fmatrix += 9) {
        double lambda = r[k], mu = 1.;
        const double s = f1[8] * r[k] + f2[8];
        if (fabs(s) > DBL_EPSILON) {
            mu = 1. / s;
            lambda *= mu;
            fmatrix[8] = 1.;
        } else
            fmatrix[8] = 0.;
        for (int i = 0; i < 8; i++) fmatrix[i] = f1[i] * lambda + f2[i] * mu;
    }
    return n;
}

/* ---- FMEstimatorCallback ------------------------------------------------ */

static void compute_error(const float* m1, const float* m2, int count, const double* F, float* err) {
    for (int i = 0; i < count; i++) {
        double a, b, c, d1, d2, s1, s2;
        const double x1 = m1[2 * i], y1 = m1[2 * i + 1], x2 = m2[2 * i], y2 = m2[2 * i + 1];
        a = F[0] * x1 + F[1] * y1 + F[2];
        b = F[3] * x1 + F[4] * y1 + F[5];
        c = F[6] * x1 + F[7] * y1 + F[8];
        s2 = 1. / (a * a + b * b);
        d2 = x2 * a + y2 * b + c;
        a = F[0] * x2 + F[3] * y2 + F[6];
        b = F[1] * x2 + F[4] * y2 + F[7];
        c = F[2] * x2 + F[5] * y2 + F[8];
        s1 = 1. / (a * a + b * b);
        d1 = x1 * a + y1 * b + c;
        const double e1 = d1 * d1 * s1, e2 = d2 * d2 * s2;
        err[i] = (float)(e1 > e2 ? e1 : e2);
    }
}

/* haveCollinearPoints: is the last of `count` points on a line through two earlier ones */
static int have_collinear(const float* m, int count) {
    const int i = count - 1;
    for (int j = 0; j < i; j++) {
        const double dx1 = m[2 * j] - m[2 * i];
        const double dy1 = m[2 * j + 1] - m[2 * i + 1];
        for (int k = 0; k < j; k++) {
            const double dx2 = m[2 * k] - m[2 * i];
            const double dy2 = m[2 * k + 1] - m[2 * i + 1];
            if (fabs(dx2 * dy1 - dy2 * dx1) <= FLT_EPSILON * (fabs(dx1) + fabs(dy1) + fabs(dx2) + fabs(dy2)))
                return 1;
        }
    }
    return 0;
}

static int find_inliers(const float* m1, const float* m2, int count, const double* F, float* err,
                        uint8_t* mask, double thresh) {
    compute_error(m1, m2, count, F, err);
    const float t = (float)(thresh * thresh);
    int good = 0;
    for (int i = 0; i < count; i++) {
        const int f = err[i] <= t;
        mask[i] = (uint8_t)f;
        good += f;
    }
    return good;
}

/* PointSetRegistrator getSubset for modelPoints = 7 */
static int get_subset(const float* m1, const float* m2, int count, orc_rng* rng, int max_attempts,
                      int idx[7], float ms1[14], float ms2[14]) {
    int iters = 0, i = 0;
    for (; iters < max_attempts; iters++) {
        for (i = 0; i < 7; ++i) {
            int idx_i;
            for (;;) {
                idx_i = rng_uniform(rng, 0, count);
                int dup = 0;
                for (int k = 0; k < i; ++k) dup |= idx[k] == idx_i;
                if (!dup) break;
            }
            idx[i] = idx_i;
            ms1[2 * i] = m1[2 * idx_i];
            ms1[2 * i + 1] = m1[2 * idx_i + 1];
            ms2[2 * i] = m2[2 * idx_i];
            ms2[2 * i + 1] = m2[2 * idx_i + 1];
        }
        if (i == 7 && (have_collinear(ms1, 7) || have_collinear(ms2, 7))) continue;
        break;
    }
    return i == 7 && iters < max_attempts;
}

int orc_ransac_update_num_iters(double p, double ep, int model_points, int max_iters) {
    p = p > 0 ? p : 0;
    p = p < 1 ? p : 1;
    ep = ep > 0 ? ep : 0;
    ep = ep < 1 ? ep : 1;
    double num = 1 - p > DBL_MIN ? 1 - p : DBL_MIN;
    double denom = 1 - pow(1 - ep, model_points);
    if (denom < DBL_MIN) return 0;
    num = log(num);
    denom = log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)lrint(num / denom);
}

/* The subsets RANSAC / LMedS would draw, one after the other, for `count` points: idx is
 * [max_iters][7]; returns how many were drawn before getSubset gave up (max_iters normally). */
int orc_fm_subsets(const float* m1, const float* m2, int count, int max_iters, int max_attempts,
                   int32_t* idx) {
    orc_rng rng = {(uint64_t)-1};
    float ms1[14], ms2[14];
    int it = 0;
    for (; it < max_iters; ++it) {
        int sub[7];
        if (!get_subset(m1, m2, count, &rng, max_attempts, sub, ms1, ms2)) break;
        for (int k = 0; k < 7; ++k) idx[it * 7 + k] = sub[k];
    }
    return it;
}

static int cmp_float(const void* a, const void* b) {
    const float x = *(const float*)a, y = *(const float*)b;
    return (x > y) - (x < y);
}

/* cv::findFundamentalMat(points1, points2, mask, FM_RANSAC, thresh, confidence) with
 * maxIters iterations.  mask has `count` entries and is written as OpenCV writes it (left
 * untouched when there are fewer than 7 points).  Returns 1 when a matrix was found. */
int orc_find_fundamental(const float* m1, const float* m2, int count, double thresh, double confidence,
                         int max_iters, double* F_out, uint8_t* mask_out, int* iters_run) {
    if (iters_run) *iters_run = 0;
    if (count < 7) return 0;
    if (count == 7) {
        double F[27];
        const int n = orc_fm_7point(m1, m2, F);
        memset(mask_out, 1, 7);
        if (n <= 0) return 0;
        if (F_out) memcpy(F_out, F, sizeof(double) * 9);
        return 1;
    }
    if (thresh <= 0) thresh = 3;
    if (confidence < DBL_EPSILON || confidence > 1 - DBL_EPSILON) confidence = 0.99;

    float* err = (float*)malloc(sizeof(float) * (size_t)count);
    uint8_t* mask = (uint8_t*)malloc((size_t)count);
    uint8_t* best_mask = (uint8_t*)malloc((size_t)count);
    double best[9] = {0};
    orc_rng rng = {(uint64_t)-1};
    float ms1[14], ms2[14];
    int sub[7];
    int result = 0;

    if (count >= 15) { /* RANSACPointSetRegistrator::run, modelPoints 7 */
        int niters = max_iters > 1 ? max_iters : 1, max_good = 0, iter;
        for (iter = 0; iter < niters; iter++) {
            if (!get_subset(m1, m2, count, &rng, 10000, sub, ms1, ms2)) {
                if (iter == 0) goto done;
                break;
            }
            double models[27];
            const int nmodels = orc_fm_7point(ms1, ms2, models);
            if (nmodels <= 0) continue;
            for (int i = 0; i < nmodels; i++) {
                const int good = find_inliers(m1, m2, count, models + 9 * i, err, mask, thresh);
                if (good > (max_good > 6 ? max_good : 6)) {
                    uint8_t* t = mask;
                    mask = best_mask;
                    best_mask = t;
                    memcpy(best, models + 9 * i, sizeof(best));
                    max_good = good;
                    niters = orc_ransac_update_num_iters(confidence, (double)(count - good) / count, 7, niters);
                }
            }
        }
        if (iters_run) *iters_run = iter;
        if (max_good > 0) {
            memcpy(mask_out, best_mask, (size_t)count);
            if (F_out) memcpy(F_out, best, sizeof(best));
            result = 1;
        }
    } else { /* LMeDSPointSetRegistrator::run */
        double min_median = DBL_MAX;
        int niters = orc_ransac_update_num_iters(confidence, 0.45, 7, max_iters), iter;
        niters = niters > 3 ? niters : 3;
        for (iter = 0; iter < niters; iter++) {
            if (!get_subset(m1, m2, count, &rng, 1000, sub, ms1, ms2)) {
                if (iter == 0) goto done;
                break;
            }
            double models[27];
            const int nmodels = orc_fm_7point(ms1, ms2, models);
            if (nmodels <= 0) continue;
            for (int i = 0; i < nmodels; i++) {
                compute_error(m1, m2, count, models + 9 * i, err);
                qsort(err, (size_t)count, sizeof(float), cmp_float); /* nth_element(count/2) */
                const double median = err[count / 2];
                if (median < min_median) {
                    min_median = median;
                    memcpy(best, models + 9 * i, sizeof(best));
                }
            }
        }
        if (iters_run) *iters_run = iter;
        if (min_median < DBL_MAX) {
            double sigma = 2.5 * 1.4826 * (1 + 5. / (count - 7)) * sqrt(min_median);
            sigma = sigma > 0.001 ? sigma : 0.001;
            const int good = find_inliers(m1, m2, count, best, err, mask, sigma);
            memcpy(mask_out, mask, (size_t)count);
            if (F_out) memcpy(F_out, best, sizeof(best));
            result = good >= 7;
        }
    }
done:
    free(err);
    free(mask);
    free(best_mask);
    return result;
}
