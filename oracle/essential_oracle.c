/*
 * essential_oracle.c -- CPU restatement of Tracking::findCameraPose,
 * Source Files/tracking.cpp:91-113: cv::findEssentialMat(prevPts, currPts, K, method, prob,
 * threshold, mask), the residual check of the first correspondence, and
 * cv::recoverPose(E, prevPts, currPts, K, R, t, mask).
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  PARITY UNPINNED: restates the published OpenCV 4.x
 * algorithm (modules/calib3d/src/five-point.cpp, ptsetreg.cpp, triangulate.cpp;
 * modules/core/src/mathfuncs.cpp for solvePoly) --
 *   findEssentialMat: points to double, normalised by K, threshold / ((fx + fy) / 2);
 *   RANSAC (or LMedS) point set registrator with 5-point minimal samples (cv::RNG seeded ~0,
 *   repeated picks redrawn, no degeneracy test), Sampson error, RANSACUpdateNumIters;
 *   five-point solver: null space of the 5x9 epipolar system, the ten cubic constraints
 *   det E = 0 and E E^T E - tr(E E^T) E / 2 = 0 in (x, y, z), elimination in Nister's
 *   monomial order, the 3x3 polynomial matrix B(z), its determinant (degree 10),
 *   solvePoly (Durand-Kerner from the start values (1+i)^k), real roots (|im| <= 1e-10),
 *   (x, y) from the null vector of B(z), E normalised to unit Frobenius norm;
 *   recoverPose: decomposeEssentialMat (SVD, det corrections, W), four candidates, DLT
 *   triangulation of every correspondence, depth tests with distanceThresh 50, mask.
 * Where OpenCV's result depends on an SVD's choice among equivalent answers, the restatement
 * takes another member of the same family: the 4-dimensional null space of the 5x9 system is
 * spanned by Gauss-Jordan elimination with complete pivoting + Gram-Schmidt (the set of
 * essential matrices found does not depend on the basis), the null vector of the 3x3 B(z) is
 * the largest cross product of two of its rows, the expanded determinant polynomial is
 * accumulated by polynomial products rather than OpenCV's pre-expanded sums, and the SVD of E
 * is a one-sided Jacobi (sign conventions of U and V differ between SVD implementations;
 * they permute the four candidate poses, of which the cheirality count picks one).
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

/* ---- small polynomial algebra in (x, y, z), total degree <= 3 ------------------------------ */

/* 20 monomials in Nister's elimination order:
 * x^3 y^3 x^2y xy^2 x^2z x^2 y^2z y^2 xyz xy | xz^2 xz x yz^2 yz y z^3 z^2 z 1 */
static const int kMono[20][3] = {{3, 0, 0}, {0, 3, 0}, {2, 1, 0}, {1, 2, 0}, {2, 0, 1}, {2, 0, 0}, {0, 2, 1},
                                 {0, 2, 0}, {1, 1, 1}, {1, 1, 0}, {1, 0, 2}, {1, 0, 1}, {1, 0, 0}, {0, 1, 2},
                                 {0, 1, 1}, {0, 1, 0}, {0, 0, 3}, {0, 0, 2}, {0, 0, 1}, {0, 0, 0}};

static int mono_index(int i, int j, int k) {
    for (int m = 0; m < 20; ++m)
        if (kMono[m][0] == i && kMono[m][1] == j && kMono[m][2] == k) return m;
    return -1;
}

typedef struct {
    double c[20];
} poly3; /* coefficient per monomial of kMono */

static void poly_mul_acc(const poly3* a, const poly3* b, double scale, poly3* out) {
    for (int p = 0; p < 20; ++p) {
        if (a->c[p] == 0) continue;
        for (int q = 0; q < 20; ++q) {
            if (b->c[q] == 0) continue;
            const int i = kMono[p][0] + kMono[q][0], j = kMono[p][1] + kMono[q][1], k = kMono[p][2] + kMono[q][2];
            if (i + j + k > 3) continue; /* cannot happen for the products formed here */
            out->c[mono_index(i, j, k)] += scale * a->c[p] * b->c[q];
        }
    }
}

/* ---- solvePoly: Durand-Kerner ---------------------------------------------------------------- */

typedef struct {
    double re, im;
} cplx;
static cplx c_mul(cplx a, cplx b) { return (cplx){a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
static cplx c_sub(cplx a, cplx b) { return (cplx){a.re - b.re, a.im - b.im}; }
static cplx c_add(cplx a, cplx b) { return (cplx){a.re + b.re, a.im + b.im}; }
static cplx c_div(cplx a, cplx b) {
    const double t = 1. / (b.re * b.re + b.im * b.im);
    return (cplx){(a.re * b.re + a.im * b.im) * t, (-a.re * b.im + a.im * b.re) * t};
}

/* coeffs[0..n0], lowest degree first; roots[0..n0-1]. */
static void solve_poly(const double* coeffs, int n0, cplx* roots, int max_iters) {
    int n = n0;
    for (; n > 1; n--)
        if (fabs(coeffs[n]) > DBL_EPSILON) break;
    cplx p = {1, 0}, r = {1, 1};
    for (int i = 0; i < n; i++) {
        roots[i] = p;
        p = c_mul(p, r);
    }
    for (int iter = 0; iter < max_iters; iter++) {
        double max_diff = 0;
        for (int i = 0; i < n; i++) {
            p = roots[i];
            cplx num = {coeffs[n], 0}, denom = {coeffs[n], 0};
            for (int j = 0; j < n; j++) {
                num = c_add(c_mul(num, p), (cplx){coeffs[n - j - 1], 0});
                if (j != i) {
                    const cplx d = c_sub(p, roots[j]);
                    if (d.re != 0 || d.im != 0) denom = c_mul(denom, d);
                }
            }
            num = c_div(num, denom);
            roots[i] = c_sub(p, num);
            const double mag = sqrt(num.re * num.re + num.im * num.im);
            if (mag > max_diff) max_diff = mag;
        }
        if (max_diff <= 0) break;
    }
    for (int i = 0; i < n; i++)
        if (fabs(roots[i].im) < 1e-100) roots[i].im = 0;
    for (; n < n0; n++) roots[n] = roots[n - 1];
}

/* ---- the five-point solver ------------------------------------------------------------------- */

/* Orthonormal basis (4 x 9, rows) of the null space of the 5 x 9 matrix q (destroyed). */
static int null_space_5x9(double* q, double basis[4][9]) {
    int col_of[5], used[9] = {0};
    for (int k = 0; k < 5; ++k) {
        int pr = -1, pc = -1;
        double best = 0;
        for (int r = k; r < 5; ++r)
            for (int c = 0; c < 9; ++c) {
                if (used[c]) continue;
                const double v = fabs(q[r * 9 + c]);
                if (v > best) best = v, pr = r, pc = c;
            }
        if (pr < 0) return 0;
        if (pr != k)
            for (int c = 0; c < 9; ++c) {
                const double t = q[k * 9 + c];
                q[k * 9 + c] = q[pr * 9 + c];
                q[pr * 9 + c] = t;
            }
        used[pc] = 1;
        col_of[k] = pc;
        const double inv = 1.0 / q[k * 9 + pc];
        for (int c = 0; c < 9; ++c) q[k * 9 + c] *= inv;
        for (int r = 0; r < 5; ++r) {
            if (r == k) continue;
            const double m = q[r * 9 + pc];
            if (m == 0) continue;
            for (int c = 0; c < 9; ++c) q[r * 9 + c] -= m * q[k * 9 + c];
        }
    }
    int nf = 0;
    for (int c = 0; c < 9 && nf < 4; ++c) {
        if (used[c]) continue;
        double* f = basis[nf++];
        for (int j = 0; j < 9; ++j) f[j] = 0;
        f[c] = 1.0;
        for (int k = 0; k < 5; ++k) f[col_of[k]] = -q[k * 9 + c];
    }
    /* modified Gram-Schmidt */
    for (int v = 0; v < 4; ++v) {
        for (int u = 0; u < v; ++u) {
            double dot = 0;
            for (int j = 0; j < 9; ++j) dot += basis[v][j] * basis[u][j];
            for (int j = 0; j < 9; ++j) basis[v][j] -= dot * basis[u][j];
        }
        double n2 = 0;
        for (int j = 0; j < 9; ++j) n2 += basis[v][j] * basis[v][j];
        if (!(n2 > 0)) return 0;
        const double inv = 1.0 / sqrt(n2);
        for (int j = 0; j < 9; ++j) basis[v][j] *= inv;
    }
    return 1;
}

/* EMEstimatorCallback::runKernel: m1, m2 are five normalised points (x, y) each; ematrix
 * receives up to 10 matrices of 9 doubles.  Returns their number. */
int orc_em_5point(const double* m1, const double* m2, double* ematrix) {
    double q[45];
    for (int i = 0; i < 5; ++i) {
        const double x1 = m1[2 * i], y1 = m1[2 * i + 1], x2 = m2[2 * i], y2 = m2[2 * i + 1];
        double* r = q + 9 * i;
        r[0] = x2 * x1; /* x2^T E x1 = 0 with E row-major */
        r[1] = x2 * y1;
        r[2] = x2;
        r[3] = y2 * x1;
        r[4] = y2 * y1;
        r[5] = y2;
        r[6] = x1;
        r[7] = y1;
        r[8] = 1.0;
    }
    double ee[4][9]; /* E = x ee[0] + y ee[1] + z ee[2] + ee[3] */
    if (!null_space_5x9(q, ee)) return 0;

    /* entries of E as linear polynomials */
    const int ix = mono_index(1, 0, 0), iy = mono_index(0, 1, 0), iz = mono_index(0, 0, 1), i1 = mono_index(0, 0, 0);
    poly3 e[3][3];
    memset(e, 0, sizeof(e));
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            e[r][c].c[ix] = ee[0][3 * r + c];
            e[r][c].c[iy] = ee[1][3 * r + c];
            e[r][c].c[iz] = ee[2][3 * r + c];
            e[r][c].c[i1] = ee[3][3 * r + c];
        }
    /* E E^T and Lambda = E E^T - tr(E E^T) / 2 I (quadratic) */
    poly3 eet[3][3], trace;
    memset(eet, 0, sizeof(eet));
    memset(&trace, 0, sizeof(trace));
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            for (int k = 0; k < 3; ++k) poly_mul_acc(&e[r][k], &e[c][k], 1.0, &eet[r][c]);
    for (int m = 0; m < 20; ++m) trace.c[m] = eet[0][0].c[m] + eet[1][1].c[m] + eet[2][2].c[m];
    for (int r = 0; r < 3; ++r)
        for (int m = 0; m < 20; ++m) eet[r][r].c[m] -= 0.5 * trace.c[m];
    /* ten cubic equations: det E, then Lambda E */
    double a[10][20];
    memset(a, 0, sizeof(a));
    {
        poly3 det, t;
        memset(&det, 0, sizeof(det));
        const int perm[6][4] = {{0, 1, 2, 1}, {1, 2, 0, 1}, {2, 0, 1, 1}, {0, 2, 1, -1}, {1, 0, 2, -1}, {2, 1, 0, -1}};
        for (int p = 0; p < 6; ++p) {
            memset(&t, 0, sizeof(t));
            poly_mul_acc(&e[0][perm[p][0]], &e[1][perm[p][1]], 1.0, &t);
            poly_mul_acc(&t, &e[2][perm[p][2]], (double)perm[p][3], &det);
        }
        memcpy(a[0], det.c, sizeof(det.c));
    }
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            poly3 s;
            memset(&s, 0, sizeof(s));
            for (int k = 0; k < 3; ++k) poly_mul_acc(&eet[r][k], &e[k][c], 1.0, &s);
            memcpy(a[1 + 3 * r + c], s.c, sizeof(s.c));
        }
    /* A = A[:, :10]^-1 A[:, 10:] by Gauss-Jordan with partial pivoting */
    for (int k = 0; k < 10; ++k) {
        int pr = k;
        for (int r = k + 1; r < 10; ++r)
            if (fabs(a[r][k]) > fabs(a[pr][k])) pr = r;
        if (a[pr][k] == 0) return 0;
        if (pr != k)
            for (int c = 0; c < 20; ++c) {
                const double t = a[k][c];
                a[k][c] = a[pr][c];
                a[pr][c] = t;
            }
        const double inv = 1.0 / a[k][k];
        for (int c = 0; c < 20; ++c) a[k][c] *= inv;
        for (int r = 0; r < 10; ++r) {
            if (r == k) continue;
            const double m = a[r][k];
            if (m == 0) continue;
            for (int c = 0; c < 20; ++c) a[r][c] -= m * a[k][c];
        }
    }
    /* B(z): rows from (x^2z, x^2), (y^2z, y^2), (xyz, xy); columns
     * x z^3, x z^2, x z, x | y z^3, y z^2, y z, y | z^4, z^3, z^2, z, 1 */
    double b[3][13];
    for (int i = 0; i < 3; ++i) {
        const double* r1 = a[4 + 2 * i] + 10; /* leading monomial carries the extra z */
        const double* r2 = a[5 + 2 * i] + 10;
        double row1[13] = {0}, row2[13] = {0};
        for (int k = 0; k < 3; ++k) {
            row1[1 + k] = r1[k];
            row1[5 + k] = r1[3 + k];
            row2[k] = r2[k];
            row2[4 + k] = r2[3 + k];
        }
        for (int k = 0; k < 4; ++k) {
            row1[9 + k] = r1[6 + k];
            row2[8 + k] = r2[6 + k];
        }
        for (int k = 0; k < 13; ++k) b[i][k] = row1[k] - row2[k];
    }
    /* det B(z), coefficients lowest degree first.  Entry polynomials, lowest first: */
    double pz[3][3][5];
    for (int i = 0; i < 3; ++i) {
        for (int k = 0; k < 5; ++k) pz[i][0][k] = pz[i][1][k] = pz[i][2][k] = 0;
        for (int k = 0; k < 4; ++k) {
            pz[i][0][k] = b[i][3 - k];
            pz[i][1][k] = b[i][7 - k];
        }
        for (int k = 0; k < 5; ++k) pz[i][2][k] = b[i][12 - k];
    }
    double c[11] = {0};
    {
        const int perm[6][4] = {{0, 1, 2, 1}, {1, 2, 0, 1}, {2, 0, 1, 1}, {0, 2, 1, -1}, {1, 0, 2, -1}, {2, 1, 0, -1}};
        for (int p = 0; p < 6; ++p) {
            double t[9] = {0};
            for (int u = 0; u < 5; ++u)
                for (int v = 0; v < 5; ++v) t[u + v] += pz[0][perm[p][0]][u] * pz[1][perm[p][1]][v];
            for (int u = 0; u < 9; ++u)
                for (int v = 0; v < 5; ++v)
                    if (u + v <= 10) c[u + v] += perm[p][3] * t[u] * pz[2][perm[p][2]][v];
        }
    }
    cplx roots[10];
    solve_poly(c, 10, roots, 300);

    int count = 0;
    for (int i = 0; i < 10; ++i) {
        if (fabs(roots[i].im) > 1e-10) continue;
        const double z1 = roots[i].re, z2 = z1 * z1, z3 = z2 * z1, z4 = z3 * z1;
        double bz[3][3];
        for (int j = 0; j < 3; ++j) {
            const double* br = b[j];
            bz[j][0] = br[0] * z3 + br[1] * z2 + br[2] * z1 + br[3];
            bz[j][1] = br[4] * z3 + br[5] * z2 + br[6] * z1 + br[7];
            bz[j][2] = br[8] * z4 + br[9] * z3 + br[10] * z2 + br[11] * z1 + br[12];
        }
        /* null vector of bz: the largest cross product of two rows */
        double best[3] = {0, 0, 0}, best_n2 = -1;
        for (int u = 0; u < 3; ++u) {
            const double* r0 = bz[u];
            const double* r1 = bz[(u + 1) % 3];
            const double v[3] = {r0[1] * r1[2] - r0[2] * r1[1], r0[2] * r1[0] - r0[0] * r1[2], r0[0] * r1[1] - r0[1] * r1[0]};
            const double n2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
            if (n2 > best_n2) best_n2 = n2, best[0] = v[0], best[1] = v[1], best[2] = v[2];
        }
        if (!(best_n2 > 0)) continue;
        const double inv_n = 1.0 / sqrt(best_n2);
        if (fabs(best[2] * inv_n) < 1e-10) continue;
        const double x = best[0] / best[2], y = best[1] / best[2];
        double* out = ematrix + 9 * count;
        double n2 = 0;
        for (int k = 0; k < 9; ++k) {
            out[k] = ee[0][k] * x + ee[1][k] * y + ee[2][k] * z1 + ee[3][k];
            n2 += out[k] * out[k];
        }
        const double inv = 1.0 / sqrt(n2);
        for (int k = 0; k < 9; ++k) out[k] *= inv;
        count++;
    }
    return count;
}

/* ---- the registrators ------------------------------------------------------------------------ */

typedef struct {
    uint64_t state;
} em_rng;
static unsigned em_next(em_rng* r) {
    r->state = (uint64_t)(unsigned)r->state * 4164903690U + (unsigned)(r->state >> 32);
    return (unsigned)r->state;
}

/* EMEstimatorCallback::computeError: Sampson distance */
static void sampson_error(const double* m1, const double* m2, int count, const double* E, float* err) {
    for (int i = 0; i < count; i++) {
        const double x1 = m1[2 * i], y1 = m1[2 * i + 1], x2 = m2[2 * i], y2 = m2[2 * i + 1];
        const double ex0 = E[0] * x1 + E[1] * y1 + E[2] * 1., ex1 = E[3] * x1 + E[4] * y1 + E[5] * 1.,
                     ex2 = E[6] * x1 + E[7] * y1 + E[8] * 1.;
        const double et0 = E[0] * x2 + E[3] * y2 + E[6] * 1., et1 = E[1] * x2 + E[4] * y2 + E[7] * 1.;
        const double x2tex1 = x2 * ex0 + y2 * ex1 + 1. * ex2;
        const double a = ex0 * ex0, b = ex1 * ex1, c = et0 * et0, d = et1 * et1;
        err[i] = (float)(x2tex1 * x2tex1 / (a + b + c + d));
    }
}

static int em_find_inliers(const double* m1, const double* m2, int count, const double* E, float* err, uint8_t* mask,
                           double thresh) {
    sampson_error(m1, m2, count, E, err);
    const float t = (float)(thresh * thresh);
    int good = 0;
    for (int i = 0; i < count; i++) {
        const int f = err[i] <= t;
        mask[i] = (uint8_t)f;
        good += f;
    }
    return good;
}

static void em_subset(const double* m1, const double* m2, int count, em_rng* rng, int idx[5], double s1[10],
                      double s2[10]) {
    for (int i = 0; i < 5; ++i) {
        int pick;
        for (;;) {
            pick = (int)(em_next(rng) % (unsigned)count);
            int dup = 0;
            for (int k = 0; k < i; ++k) dup |= idx[k] == pick;
            if (!dup) break;
        }
        idx[i] = pick;
        s1[2 * i] = m1[2 * pick];
        s1[2 * i + 1] = m1[2 * pick + 1];
        s2[2 * i] = m2[2 * pick];
        s2[2 * i + 1] = m2[2 * pick + 1];
    }
}

static int cmp_float(const void* a, const void* b) {
    const float x = *(const float*)a, y = *(const float*)b;
    return (x > y) - (x < y);
}

/* The subsets the loop draws, for tests: out is [max_iters][5]. */
int orc_em_subsets(int count, int max_iters, int32_t* out) {
    if (count < 6) return 0;
    em_rng rng = {(uint64_t)-1};
    double dummy[2] = {0, 0};
    (void)dummy;
    for (int it = 0; it < max_iters; ++it) {
        int idx[5];
        for (int i = 0; i < 5; ++i) {
            int pick;
            for (;;) {
                pick = (int)(em_next(&rng) % (unsigned)count);
                int dup = 0;
                for (int k = 0; k < i; ++k) dup |= idx[k] == pick;
                if (!dup) break;
            }
            idx[i] = pick;
        }
        for (int i = 0; i < 5; ++i) out[5 * it + i] = idx[i];
    }
    return max_iters;
}

/* cv::findEssentialMat(points1, points2, K, method, prob, threshold, mask) with 1000
 * iterations at most.  pts: float pixel coordinates; K: 3x3 row-major; method 0 = RANSAC,
 * 1 = LMEDS.  E_out receives rows x 9 doubles (one 3x3 matrix, or every solution of the
 * direct solve when there are exactly five points: OpenCV stacks them into a 3k x 3 matrix);
 * returns the number of matrices written (0 = none found; the mask is then left as given). */
int orc_find_essential(const float* pts1, const float* pts2, int count, const double* K, int method, double prob,
                       double threshold, int max_iters, double* E_out, uint8_t* mask_out, int* iters_run) {
    if (iters_run) *iters_run = 0;
    if (count < 5) return 0;
    const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    double* m1 = (double*)malloc(sizeof(double) * 2 * (size_t)count);
    double* m2 = (double*)malloc(sizeof(double) * 2 * (size_t)count);
    for (int i = 0; i < count; ++i) {
        m1[2 * i] = ((double)pts1[2 * i] - cx) / fx;
        m1[2 * i + 1] = ((double)pts1[2 * i + 1] - cy) / fy;
        m2[2 * i] = ((double)pts2[2 * i] - cx) / fx;
        m2[2 * i + 1] = ((double)pts2[2 * i + 1] - cy) / fy;
    }
    threshold /= (fx + fy) / 2;
    int result = 0;
    if (count == 5) {
        double models[90];
        const int n = orc_em_5point(m1, m2, models);
        memset(mask_out, 1, 5);
        if (n > 0) memcpy(E_out, models, sizeof(double) * 9 * (size_t)n);
        free(m1);
        free(m2);
        return n;
    }
    float* err = (float*)malloc(sizeof(float) * (size_t)count);
    uint8_t* mask = (uint8_t*)malloc((size_t)count);
    uint8_t* best_mask = (uint8_t*)malloc((size_t)count);
    double best[9] = {0};
    em_rng rng = {(uint64_t)-1};
    int idx[5];
    double s1[10], s2[10], models[90];
    if (method == 0) { /* RANSACPointSetRegistrator::run, modelPoints 5 */
        int niters = max_iters > 1 ? max_iters : 1, max_good = 0, iter;
        for (iter = 0; iter < niters; iter++) {
            em_subset(m1, m2, count, &rng, idx, s1, s2);
            const int nmodels = orc_em_5point(s1, s2, models);
            if (nmodels <= 0) continue;
            for (int i = 0; i < nmodels; i++) {
                const int good = em_find_inliers(m1, m2, count, models + 9 * i, err, mask, threshold);
                if (good > (max_good > 4 ? max_good : 4)) {
                    uint8_t* t = mask;
                    mask = best_mask;
                    best_mask = t;
                    memcpy(best, models + 9 * i, sizeof(best));
                    max_good = good;
                    niters = orc_ransac_update_num_iters(prob, (double)(count - good) / count, 5, niters);
                }
            }
        }
        if (iters_run) *iters_run = iter;
        if (max_good > 0) {
            memcpy(mask_out, best_mask, (size_t)count);
            memcpy(E_out, best, sizeof(best));
            result = 1;
        }
    } else { /* LMeDSPointSetRegistrator::run */
        double min_median = DBL_MAX;
        int niters = orc_ransac_update_num_iters(prob, 0.45, 5, max_iters), iter;
        niters = niters > 3 ? niters : 3;
        for (iter = 0; iter < niters; iter++) {
            em_subset(m1, m2, count, &rng, idx, s1, s2);
            const int nmodels = orc_em_5point(s1, s2, models);
            if (nmodels <= 0) continue;
            for (int i = 0; i < nmodels; i++) {
                sampson_error(m1, m2, count, models + 9 * i, err);
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
            double sigma = 2.5 * 1.4826 * (1 + 5. / (count - 5)) * sqrt(min_median);
            sigma = sigma > 0.001 ? sigma : 0.001;
            const int good = em_find_inliers(m1, m2, count, best, err, mask, sigma);
            memcpy(mask_out, mask, (size_t)count);
            memcpy(E_out, best, sizeof(best));
            result = good >= 5;
        }
    }
    free(err);
    free(mask);
    free(best_mask);
    free(m1);
    free(m2);
    return result;
}

/* ---- recoverPose ------------------------------------------------------------------------------ */

/* One-sided Jacobi SVD of a 3x3 matrix (row-major a): a = u diag(w) v^T, w descending. */
static void svd3(const double* a, double* u, double* w, double* v) {
    double at[3][3], vt[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}; /* at[i] = column i of a (rotated in place) */
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) at[i][j] = a[3 * j + i];
    for (int sweep = 0; sweep < 30; ++sweep) {
        int changed = 0;
        for (int i = 0; i < 2; ++i)
            for (int j = i + 1; j < 3; ++j) {
                double p = 0, qa = 0, qb = 0;
                for (int k = 0; k < 3; ++k) p += at[i][k] * at[j][k], qa += at[i][k] * at[i][k], qb += at[j][k] * at[j][k];
                if (fabs(p) <= DBL_EPSILON * sqrt(qa * qb)) continue;
                p *= 2;
                const double beta = qa - qb, gamma = hypot(p, beta);
                double c, s;
                if (beta < 0) {
                    const double delta = (gamma - beta) * 0.5;
                    s = sqrt(delta / gamma);
                    c = p / (gamma * s * 2);
                } else {
                    c = sqrt((gamma + beta) / (gamma * 2));
                    s = p / (gamma * c * 2);
                }
                for (int k = 0; k < 3; ++k) {
                    const double t0 = c * at[i][k] + s * at[j][k], t1 = -s * at[i][k] + c * at[j][k];
                    at[i][k] = t0;
                    at[j][k] = t1;
                    const double v0 = c * vt[i][k] + s * vt[j][k], v1 = -s * vt[i][k] + c * vt[j][k];
                    vt[i][k] = v0;
                    vt[j][k] = v1;
                }
                changed = 1;
            }
        if (!changed) break;
    }
    double sv[3];
    for (int i = 0; i < 3; ++i) sv[i] = sqrt(at[i][0] * at[i][0] + at[i][1] * at[i][1] + at[i][2] * at[i][2]);
    int order[3] = {0, 1, 2};
    for (int i = 0; i < 2; ++i)
        for (int j = i + 1; j < 3; ++j)
            if (sv[order[j]] > sv[order[i]]) {
                const int t = order[i];
                order[i] = order[j];
                order[j] = t;
            }
    for (int c = 0; c < 3; ++c) {
        const int i = order[c];
        w[c] = sv[i];
        for (int k = 0; k < 3; ++k) {
            v[3 * k + c] = vt[i][k];
            u[3 * k + c] = sv[i] > 0 ? at[i][k] / sv[i] : 0;
        }
    }
    /* a rank-2 matrix leaves the last column of u undetermined: complete it */
    if (!(w[2] > DBL_EPSILON * w[0])) {
        u[2] = u[3] * u[7] - u[6] * u[4];
        u[5] = u[6] * u[1] - u[0] * u[7];
        u[8] = u[0] * u[4] - u[3] * u[1];
    }
}

static double det3(const double* m) {
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}

/* cv::decomposeEssentialMat */
void orc_decompose_essential(const double* E, double* R1, double* R2, double* t) {
    double u[9], w[3], v[9];
    svd3(E, u, w, v);
    if (det3(u) < 0)
        for (int i = 0; i < 9; ++i) u[i] = -u[i];
    if (det3(v) < 0) /* det(Vt) == det(V) */
        for (int i = 0; i < 9; ++i) v[i] = -v[i];
    /* W = [0 1 0; -1 0 0; 0 0 1]; R1 = U W Vt, R2 = U W^T Vt */
    double uw[9], uwt[9];
    for (int r = 0; r < 3; ++r) {
        uw[3 * r + 0] = -u[3 * r + 1];
        uw[3 * r + 1] = u[3 * r + 0];
        uw[3 * r + 2] = u[3 * r + 2];
        uwt[3 * r + 0] = u[3 * r + 1];
        uwt[3 * r + 1] = -u[3 * r + 0];
        uwt[3 * r + 2] = u[3 * r + 2];
    }
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            double s1 = 0, s2 = 0;
            for (int k = 0; k < 3; ++k) s1 += uw[3 * r + k] * v[3 * c + k], s2 += uwt[3 * r + k] * v[3 * c + k];
            R1[3 * r + c] = s1;
            R2[3 * r + c] = s2;
        }
    for (int r = 0; r < 3; ++r) t[r] = u[3 * r + 2];
}


/* The cheirality mask of one candidate pose (R, t) as cv::recoverPose computes it. */
static int cheirality(const double* m1, const double* m2, int count, const double* R, const double* t, double dist,
                      uint8_t* mask) {
    const double P0[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    double P1[12];
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) P1[4 * r + c] = R[3 * r + c];
        P1[4 * r + 3] = t[r];
    }
    int good = 0;
    for (int i = 0; i < count; ++i) {
        const double x1 = m1[2 * i], y1 = m1[2 * i + 1], x2 = m2[2 * i], y2 = m2[2 * i + 1];
        double a[16], q[4];
        for (int c = 0; c < 4; ++c) {
            a[c] = x1 * P0[8 + c] - P0[c];
            a[4 + c] = y1 * P0[8 + c] - P0[4 + c];
            a[8 + c] = x2 * P1[8 + c] - P1[c];
            a[12 + c] = y2 * P1[8 + c] - P1[4 + c];
        }
        double w4[4];
        orc_svd4_null(a, w4, q);
        int ok = q[2] * q[3] > 0;
        const double X = q[0] / q[3], Y = q[1] / q[3], Z = q[2] / q[3];
        ok = ok && Z < dist;
        const double z2 = P1[8] * X + P1[9] * Y + P1[10] * Z + P1[11] * 1.0;
        ok = ok && z2 > 0 && z2 < dist;
        mask[i] = ok ? 255 : 0;
        good += ok;
    }
    return good;
}

/* cv::recoverPose(E, points1, points2, K, R, t, mask) (distanceThresh 50).  mask_io: on
 * input the inlier mask of findEssentialMat (NULL = none), on output the mask of the chosen
 * pose ANDed with it, as OpenCV's bitwise_and leaves it.  Returns the number of points that
 * pass. */
int orc_recover_pose(const double* E, const float* pts1, const float* pts2, int count, const double* K, double* R,
                     double* t, uint8_t* mask_io) {
    const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    double* m1 = (double*)malloc(sizeof(double) * 2 * (size_t)(count > 0 ? count : 1));
    double* m2 = (double*)malloc(sizeof(double) * 2 * (size_t)(count > 0 ? count : 1));
    for (int i = 0; i < count; ++i) {
        m1[2 * i] = ((double)pts1[2 * i] - cx) / fx;
        m1[2 * i + 1] = ((double)pts1[2 * i + 1] - cy) / fy;
        m2[2 * i] = ((double)pts2[2 * i] - cx) / fx;
        m2[2 * i + 1] = ((double)pts2[2 * i + 1] - cy) / fy;
    }
    double R1[9], R2[9], tt[3], tn[3];
    orc_decompose_essential(E, R1, R2, tt);
    for (int i = 0; i < 3; ++i) tn[i] = -tt[i];
    uint8_t* masks[4];
    int good[4];
    const double* Rs[4] = {R1, R2, R1, R2};
    const double* ts[4] = {tt, tt, tn, tn};
    for (int k = 0; k < 4; ++k) {
        masks[k] = (uint8_t*)malloc((size_t)(count > 0 ? count : 1));
        cheirality(m1, m2, count, Rs[k], ts[k], 50.0, masks[k]);
        good[k] = 0;
        for (int i = 0; i < count; ++i) {
            if (mask_io) masks[k][i] &= mask_io[i];
            good[k] += masks[k][i] != 0;
        }
    }
    int pick;
    if (good[0] >= good[1] && good[0] >= good[2] && good[0] >= good[3]) pick = 0;
    else if (good[1] >= good[0] && good[1] >= good[2] && good[1] >= good[3]) pick = 1;
    else if (good[2] >= good[0] && good[2] >= good[1] && good[2] >= good[3]) pick = 2;
    else pick = 3;
    memcpy(R, Rs[pick], sizeof(double) * 9);
    memcpy(t, ts[pick], sizeof(double) * 3);
    if (mask_io) memcpy(mask_io, masks[pick], (size_t)count);
    const int n = good[pick];
    for (int k = 0; k < 4; ++k) free(masks[k]);
    free(m1);
    free(m2);
    return n;
}

/* The residual test of Tracking::findCameraPose, Source Files/tracking.cpp:99-107:
 * | p1^T K^-T E K^-1 p0 | of the FIRST correspondence. */
double orc_essential_residual(const double* E, const float* p0, const float* p1, const double* K) {
    const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    /* K^-1 p for an upper triangular K without skew */
    const double a[3] = {((double)p0[0] - cx) / fx, ((double)p0[1] - cy) / fy, 1.0};
    const double b[3] = {((double)p1[0] - cx) / fx, ((double)p1[1] - cy) / fy, 1.0};
    double s = 0;
    for (int r = 0; r < 3; ++r) s += b[r] * (E[3 * r] * a[0] + E[3 * r + 1] * a[1] + E[3 * r + 2] * a[2]);
    return fabs(s);
}
