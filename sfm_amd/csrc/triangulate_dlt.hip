// Batched two-view DLT triangulation with reprojection / depth filtering.
//
// Replaces Reconstruction::triangulateCloud and pointsToRGBCloud
// (Source Files/reconstruction.cpp:41-73 and :6-39): per correspondence the 4x4
// homogeneous system of cv::triangulatePoints, its right singular vector of the
// smallest singular value (one-sided Jacobi, the algorithm of cv::SVD::compute),
// division by w, reprojection with the separately supplied R, t, and the mask
//   err < maxProjErr && depth > minDist && depth < maxDist.
//
// One lane per correspondence, all arithmetic in fp64 as in the reference (points
// are widened to double by Header Files/common.h:47-57 before
// cv::triangulatePoints sees them).  Two solvers for the right singular vector
// of the smallest singular value of the 4x4 system A:
//   - fast path: the eigenvector of the smallest eigenvalue of M = A^T A by shifted
//     inverse iteration.  Every step factors M - s I = L D L^T with a shift s kept
//     below the smallest eigenvalue (Rayleigh quotient minus the residual norm), so
//     the factorisation stays positive definite and needs no pivoting; the
//     contamination by the other eigenvectors is squared per step.  A point whose
//     iteration has not settled after kMaxInvIt steps is appended to a list;
//   - one-sided Jacobi, the algorithm of cv::SVD::compute itself (36 doubles in
//     VGPRs, every matrix loop unrolled, wave-uniform sweep loop).  It handles the
//     listed points, and every point when SFM_TRIANGULATE_JACOBI=1 is set.
// Both give the same vector up to rounding (observed ~1e-12 relative on
// well-conditioned points, against a stated tolerance of 1e-4).
#include <cfloat>
#include <cstdlib>
#include <mutex>

#include "sfm_internal.h"

namespace sfm {

namespace {

struct TriArgs {
    double P1[12];
    double P2[12];
    double R[9];
    double t[3];
    double Pc2[4];  // third row of K*[R|t]
    double fx, fy, cx, cy;
    const float2* prev_pts;     // point i of each image, or (matches != nullptr) ...
    const float2* curr_pts;
    const sfm_dmatch* matches;  // ... point i = key point matches[i].queryIdx / .trainIdx
    const uint8_t* kp_prev;     // key point records, `pt` (two floats) first, strides apart
    const uint8_t* kp_curr;
    size_t kp_prev_stride, kp_curr_stride;
    const int32_t* count;       // device-side number of correspondences (nullptr: n)
    int n;
    const uint8_t* bgr;
    int img_w, img_h;
    size_t img_stride;
    double min_dist, max_dist, max_err;
    double* X;
    uint8_t* rgb;
    uint8_t* mask;
    double* err;
};

// Number of correspondences: the host's n, clipped by the device-side count when given.
__device__ __forceinline__ int live_count(const TriArgs& a) {
    if (!a.count) return a.n;
    const int c = *a.count;
    return c < a.n ? (c < 0 ? 0 : c) : a.n;
}

// Image points of correspondence i: straight from the point lists, or through the match
// list from the two key point arrays (the gather the reference does on the host at
// Source Files/feature_processing.cpp:164-165 and :196-197).
__device__ __forceinline__ void load_points(const TriArgs& a, int i, float2& p1, float2& p2) {
    if (a.matches) {
        const sfm_dmatch m = a.matches[i];
        p1 = *reinterpret_cast<const float2*>(a.kp_prev + (size_t)m.queryIdx * a.kp_prev_stride);
        p2 = *reinterpret_cast<const float2*>(a.kp_curr + (size_t)m.trainIdx * a.kp_curr_stride);
    } else {
        p1 = a.prev_pts[i];
        p2 = a.curr_pts[i];
    }
}

// One Jacobi rotation of rows i, j of At (columns of A) and of Vt.
template <int I, int J>
__device__ __forceinline__ bool rotate(double (&At)[4][4], double (&Vt)[4][4], double (&W)[4]) {
    const double eps = DBL_EPSILON * 10;
    double a = W[I], b = W[J], p = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) p += At[I][k] * At[J][k];
    if (fabs(p) <= eps * sqrt(a * b)) return false;
    p *= 2;
    const double beta = a - b, gamma = hypot(p, beta);
    double c, s;
    if (beta < 0) {
        const double delta = (gamma - beta) * 0.5;
        s = sqrt(delta / gamma);
        c = p / (gamma * s * 2);
    } else {
        c = sqrt((gamma + beta) / (gamma * 2));
        s = p / (gamma * c * 2);
    }
    a = b = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double t0 = c * At[I][k] + s * At[J][k];
        const double t1 = -s * At[I][k] + c * At[J][k];
        At[I][k] = t0;
        At[J][k] = t1;
        a += t0 * t0;
        b += t1 * t1;
    }
    W[I] = a;
    W[J] = b;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double t0 = c * Vt[I][k] + s * Vt[J][k];
        const double t1 = -s * Vt[I][k] + c * Vt[J][k];
        Vt[I][k] = t0;
        Vt[J][k] = t1;
    }
    return true;
}

// The 4x4 system of cv::triangulatePoints.  kFused = false keeps OpenCV's operation order
// (multiply, then subtract: what the Jacobi path, the faithful one, is compared bit-wise
// against); the fast path fuses them.
template <bool kFused>
__device__ __forceinline__ void build_system(const TriArgs& a, double x1, double y1, double x2,
                                             double y2, double (&At)[4][4]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (kFused) {
            At[k][0] = fma(x1, a.P1[8 + k], -a.P1[k]);
            At[k][1] = fma(y1, a.P1[8 + k], -a.P1[4 + k]);
            At[k][2] = fma(x2, a.P2[8 + k], -a.P2[k]);
            At[k][3] = fma(y2, a.P2[8 + k], -a.P2[4 + k]);
        } else {
            At[k][0] = x1 * a.P1[8 + k] - a.P1[k];
            At[k][1] = y1 * a.P1[8 + k] - a.P1[4 + k];
            At[k][2] = x2 * a.P2[8 + k] - a.P2[k];
            At[k][3] = y2 * a.P2[8 + k] - a.P2[4 + k];
        }
    }
}

// cv::SVD::compute's one-sided Jacobi on At (rows = columns of A); v = last row of Vt
// after the descending sort of the singular values.
__device__ __forceinline__ void null_vector_jacobi(double (&At)[4][4], double (&v)[4]) {
    double Vt[4][4], W[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        double sd = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            sd += At[r][k] * At[r][k];
            Vt[r][k] = r == k ? 1.0 : 0.0;
        }
        W[r] = sd;
    }

    for (int iter = 0; iter < 30; ++iter) {
        bool changed = false;
        changed |= rotate<0, 1>(At, Vt, W);
        changed |= rotate<0, 2>(At, Vt, W);
        changed |= rotate<0, 3>(At, Vt, W);
        changed |= rotate<1, 2>(At, Vt, W);
        changed |= rotate<1, 3>(At, Vt, W);
        changed |= rotate<2, 3>(At, Vt, W);
        if (!__any(changed)) break;
    }

#pragma unroll
    for (int r = 0; r < 4; ++r) {
        double sd = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) sd += At[r][k] * At[r][k];
        W[r] = sqrt(sd);
    }
    // The reference sorts the singular values in descending order with a
    // selection sort and takes the last row of Vt; follow the permutation.
    int perm[4] = {0, 1, 2, 3};
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        double wmax = W[r];
        int j = r;
#pragma unroll
        for (int k = r + 1; k < 4; ++k)
            if (wmax < W[k]) {
                wmax = W[k];
                j = k;
            }
        // swap entries r and j of W and perm (static indices via selects)
        const double wr = W[r];
        const int pr = perm[r];
#pragma unroll
        for (int k = r + 1; k < 4; ++k)
            if (k == j) {
                W[k] = wr;
                perm[r] = perm[k];
                perm[k] = pr;
            }
        W[r] = wmax;
    }
    const int last = perm[3];
#pragma unroll
    for (int k = 0; k < 4; ++k)
        v[k] = last == 0 ? Vt[0][k] : last == 1 ? Vt[1][k] : last == 2 ? Vt[2][k] : Vt[3][k];
}

constexpr int kMaxInvIt = 10;

// Solves (M - shift I) x = b by L D L^T; M symmetric, given by its upper triangle
// m[] = {00,01,02,03,11,12,13,22,23,33}.  Returns false when the shift was not below the
// spectrum: one of the first three pivots not positive, or the last one negative by more
// than rounding (1e17 tiny = 1e-13 trace; a matrix with one negative eigenvalue can have its
// one negative pivot in the last place).  A last pivot within rounding of zero is replaced
// by a tiny positive one (exactly singular system).
__device__ __forceinline__ bool ldl_solve(const double (&m)[10], double shift, double tiny,
                                          const double (&b)[4], double (&x)[4]) {
    const double a00 = m[0] - shift, a11 = m[4] - shift, a22 = m[7] - shift, a33 = m[9] - shift;
    const double a01 = m[1], a02 = m[2], a03 = m[3], a12 = m[5], a13 = m[6], a23 = m[8];
    const double d0 = a00;
    const double i0 = 1.0 / d0;
    const double l10 = a01 * i0, l20 = a02 * i0, l30 = a03 * i0;
    const double d1 = fma(-l10, a01, a11);
    const double i1 = 1.0 / d1;
    const double t21 = fma(-l20, a01, a12), t31 = fma(-l30, a01, a13);
    const double l21 = t21 * i1, l31 = t31 * i1;
    const double d2 = fma(-l21, t21, fma(-l20, a02, a22));
    const double i2 = 1.0 / d2;
    const double t32 = fma(-l31, t21, fma(-l30, a02, a23));
    const double l32 = t32 * i2;
    double d3 = fma(-l32, t32, fma(-l31, t31, fma(-l30, a03, a33)));
    const bool ok = d0 > 0 && d1 > 0 && d2 > 0 && d3 > -1e17 * tiny;
    d3 = d3 > tiny ? d3 : tiny;
    const double i3 = 1.0 / d3;
    const double y0 = b[0];
    const double y1 = fma(-l10, y0, b[1]);
    const double y2 = fma(-l21, y1, fma(-l20, y0, b[2]));
    const double y3 = fma(-l32, y2, fma(-l31, y1, fma(-l30, y0, b[3])));
    x[3] = y3 * i3;
    x[2] = fma(-l32, x[3], y2 * i2);
    x[1] = fma(-l31, x[3], fma(-l21, x[2], y1 * i1));
    x[0] = fma(-l30, x[3], fma(-l20, x[2], fma(-l10, x[1], y0 * i0)));
    return ok;
}

// Smallest eigenvector of M = A^T A by safely shifted inverse iteration.
// Returns false when the iteration did not settle (caller falls back to Jacobi).
__device__ __forceinline__ bool null_vector_invit(const double (&At)[4][4], double (&v)[4]) {
    double m[10];
    {
        int t = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = i; j < 4; ++j) {
                double s = 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) s = fma(At[i][r], At[j][r], s);
                m[t++] = s;
            }
    }
    const double trace = m[0] + m[4] + m[7] + m[9];
    const double tiny = trace * 1e-30 + 1e-300;
    double x[4] = {0.1, 0.2, 0.9, 0.3};
    double shift = -trace * 1e-15;  // strictly below the spectrum of a positive semidefinite M
    bool done = false, bad = false;
    double prev_delta = 1.0;
    for (int it = 0; it < kMaxInvIt; ++it) {
        double y[4];
        const bool ok = ldl_solve(m, shift, tiny, x, y);
        // normalise
        const double n2 = fma(y[0], y[0], fma(y[1], y[1], fma(y[2], y[2], y[3] * y[3])));
        const double inv = rsqrt(n2);
        double z[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) z[k] = y[k] * inv;
        // change against the previous iterate, up to sign
        const double dot = fma(z[0], x[0], fma(z[1], x[1], fma(z[2], x[2], z[3] * x[3])));
        const double sgn = dot < 0 ? -1.0 : 1.0;
        double delta = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            z[k] *= sgn;
            delta = fmax(delta, fabs(z[k] - x[k]));
        }
        if (!done) {
#pragma unroll
            for (int k = 0; k < 4; ++k) x[k] = z[k];
            bad |= !ok || !(n2 > 0) || !(n2 < 1e300);
        }
        // the moves of two consecutive steps give the contraction q = delta / prev_delta and
        // the distance left as about delta q / (1 - q): accepted below 1e-10 (or when the
        // iterate has stopped moving at rounding level); a small move alone says nothing
        // about the distance left when the moves shrink slowly
        done |= (it > 0 && delta < 1e-7 &&
                 (delta <= 1e-13 || delta * delta <= 1e-10 * (prev_delta - delta))) || bad;
        if (!done) prev_delta = delta;
        if (__all(done)) break;
        // Rayleigh quotient and residual: an eigenvalue lies within |r| of rho; it is the
        // smallest one once the iterate is dominated by that eigenvector
        const double w0 = fma(m[0], x[0], fma(m[1], x[1], fma(m[2], x[2], m[3] * x[3])));
        const double w1 = fma(m[1], x[0], fma(m[4], x[1], fma(m[5], x[2], m[6] * x[3])));
        const double w2 = fma(m[2], x[0], fma(m[5], x[1], fma(m[7], x[2], m[8] * x[3])));
        const double w3 = fma(m[3], x[0], fma(m[6], x[1], fma(m[8], x[2], m[9] * x[3])));
        const double rho = fma(w0, x[0], fma(w1, x[1], fma(w2, x[2], w3 * x[3])));
        const double r0 = fma(-rho, x[0], w0), r1 = fma(-rho, x[1], w1), r2 = fma(-rho, x[2], w2),
                     r3 = fma(-rho, x[3], w3);
        const double rn = sqrt(fma(r0, r0, fma(r1, r1, fma(r2, r2, r3 * r3))));
        const double s = rho - 1.0625 * rn - trace * 1e-15;
        if (!done) shift = s > shift ? s : shift;  // never move the shift back
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = x[k];
    return done && !bad;
}

// Reciprocal of a positive, normal double: hardware estimate + two Newton steps
// (about 1 ulp; the factorisation below does not need correctly rounded quotients).
__device__ __forceinline__ double rcp_pos(double d) {
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    return r;
}

// 1/sqrt(d) of a positive, normal double: hardware estimate + two Newton steps.
__device__ __forceinline__ double rsqrt_pos(double d) {
    double r = __builtin_amdgcn_rsq(d);
    const double h = 0.5 * d;
    r = fma(fma(-h * r, r, 0.5), r, r);
    r = fma(fma(-h * r, r, 0.5), r, r);
    return r;
}

// Fixed-cost fast path: ONE factorisation M + e I = L D L^T (e = 1e-15 trace, so the
// matrix is positive definite), three inverse-iteration solves with it, and an
// a-posteriori acceptance test.  With a fixed shift the contamination of the iterate by
// the other eigenvectors shrinks by the same factor q per solve, so the last two moves
// d2 = |x2 - x1| (about the error of x1) and d3 = |x3 - x2| (about the error of x2) give
// q = d3 / d2 and the error of x3 as d3 * q.  Points with d3^2 >= 1e-12 d2 (a small
// eigenvalue gap: noisy or mismatched correspondences) are reported as not settled.
__device__ __forceinline__ bool null_vector_invit3(const double (&At)[4][4], double (&v)[4]) {
    double m[10];
    {
        int t = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = i; j < 4; ++j) {
                double s = 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) s = fma(At[i][r], At[j][r], s);
                m[t++] = s;
            }
    }
    const double trace = m[0] + m[4] + m[7] + m[9];
    const double shift = trace * 1e-15;
    const double tiny = trace * 1e-30 + 1e-300;
    const double a00 = m[0] + shift, a11 = m[4] + shift, a22 = m[7] + shift, a33 = m[9] + shift;
    const double a01 = m[1], a02 = m[2], a03 = m[3], a12 = m[5], a13 = m[6], a23 = m[8];
    const double i0 = rcp_pos(a00 > tiny ? a00 : tiny);
    const double l10 = a01 * i0, l20 = a02 * i0, l30 = a03 * i0;
    const double d1 = fma(-l10, a01, a11);
    const double i1 = rcp_pos(d1 > tiny ? d1 : tiny);
    const double t21 = fma(-l20, a01, a12), t31 = fma(-l30, a01, a13);
    const double l21 = t21 * i1, l31 = t31 * i1;
    const double d2 = fma(-l21, t21, fma(-l20, a02, a22));
    const double i2 = rcp_pos(d2 > tiny ? d2 : tiny);
    const double t32 = fma(-l31, t21, fma(-l30, a02, a23));
    const double l32 = t32 * i2;
    const double d3 = fma(-l32, t32, fma(-l31, t31, fma(-l30, a03, a33)));
    const double i3 = rcp_pos(d3 > tiny ? d3 : tiny);
    const bool pivots_ok = a00 > tiny && d1 > tiny && d2 > tiny;

    auto solve = [&](const double (&b)[4], double (&x)[4]) {
        const double y0 = b[0];
        const double y1 = fma(-l10, y0, b[1]);
        const double y2 = fma(-l21, y1, fma(-l20, y0, b[2]));
        const double y3 = fma(-l32, y2, fma(-l31, y1, fma(-l30, y0, b[3])));
        x[3] = y3 * i3;
        x[2] = fma(-l32, x[3], y2 * i2);
        x[1] = fma(-l31, x[3], fma(-l21, x[2], y1 * i1));
        x[0] = fma(-l30, x[3], fma(-l20, x[2], fma(-l10, x[1], y0 * i0)));
    };
    // unit-norm iterate with the sign that makes it closest to `ref`; returns the move
    auto settle = [&](double (&y)[4], const double (&ref)[4], bool& finite) {
        const double n2 = fma(y[0], y[0], fma(y[1], y[1], fma(y[2], y[2], y[3] * y[3])));
        finite = n2 > 0 && n2 < 1e300;
        const double dot = fma(y[0], ref[0], fma(y[1], ref[1], fma(y[2], ref[2], y[3] * ref[3])));
        double inv = rsqrt_pos(finite ? n2 : 1.0);
        inv = dot < 0 ? -inv : inv;
        double delta = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            y[k] *= inv;
            delta = fmax(delta, fabs(y[k] - ref[k]));
        }
        return delta;
    };

    const double x0[4] = {0.1, 0.2, 0.9, 0.3};
    double x1[4], x2[4], x3[4];
    bool f1, f2, f3;
    solve(x0, x1);
    (void)settle(x1, x0, f1);
    solve(x1, x2);
    const double m2 = settle(x2, x1, f2);
    solve(x2, x3);
    const double m3 = settle(x3, x2, f3);
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = x3[k];
    return pivots_ok && f1 && f2 && f3 && (m3 <= 1e-13 || m3 * m3 < 1e-12 * m2);
}

// ---- the default solver: inverse iteration on a triangular factor of A ------------------------
// Householder reflections bring A to R (A = Q R, Q never formed); (R^T R) y = b is two
// triangular solves.  The rounding of the result then grows with the first power of the
// system's condition number, not with its square as for a factorisation of A^T A: for
// baselines far below the depth (consecutive video frames) the null vector agrees with the
// one-sided Jacobi SVD to ~1e-11 where the normal-equations form gives 1e-7 to 1e-6.
struct TriFactor {
    double r01, r02, r03, r12, r13, r23;   // R above the diagonal
    double i0, i1, i2, i3;                 // reciprocals of the diagonal
    bool pivots_ok;
};

__device__ __forceinline__ void qr_factor(const double (&At)[4][4], TriFactor& f) {
    double c0[4], c1[4], c2[4], c3[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        c0[r] = At[0][r];
        c1[r] = At[1][r];
        c2[r] = At[2][r];
        c3[r] = At[3][r];
    }
    constexpr double kFloor = 1e-290;   // keeps the estimates' inputs normal; such systems fail the finite test
    // reflection 0: column 0, rows 0..3
    const double n0 = fma(c0[0], c0[0], fma(c0[1], c0[1], fma(c0[2], c0[2], c0[3] * c0[3])));
    const double rs0 = rsqrt_pos(fmax(n0, kFloor));
    const double norm0 = n0 * rs0;
    const double r00 = c0[0] >= 0 ? -norm0 : norm0;
    const double h0 = c0[0] - r00;                                      // first entry of the reflection vector
    const double beta0 = rcp_pos(fmax(fma(-r00, c0[0], n0), kFloor));   // 2 / |h|^2
    auto reflect0 = [&](double (&c)[4]) {
        const double s = beta0 * fma(h0, c[0], fma(c0[1], c[1], fma(c0[2], c[2], c0[3] * c[3])));
        c[0] = fma(-s, h0, c[0]);
        c[1] = fma(-s, c0[1], c[1]);
        c[2] = fma(-s, c0[2], c[2]);
        c[3] = fma(-s, c0[3], c[3]);
    };
    reflect0(c1);
    reflect0(c2);
    reflect0(c3);
    f.r01 = c1[0];
    f.r02 = c2[0];
    f.r03 = c3[0];
    // reflection 1: column 1, rows 1..3
    const double n1 = fma(c1[1], c1[1], fma(c1[2], c1[2], c1[3] * c1[3]));
    const double rs1 = rsqrt_pos(fmax(n1, kFloor));
    const double norm1 = n1 * rs1;
    const double r11 = c1[1] >= 0 ? -norm1 : norm1;
    const double h1 = c1[1] - r11;
    const double beta1 = rcp_pos(fmax(fma(-r11, c1[1], n1), kFloor));
    auto reflect1 = [&](double (&c)[4]) {
        const double s = beta1 * fma(h1, c[1], fma(c1[2], c[2], c1[3] * c[3]));
        c[1] = fma(-s, h1, c[1]);
        c[2] = fma(-s, c1[2], c[2]);
        c[3] = fma(-s, c1[3], c[3]);
    };
    reflect1(c2);
    reflect1(c3);
    f.r12 = c2[1];
    f.r13 = c3[1];
    // reflection 2: column 2, rows 2..3
    const double n2 = fma(c2[2], c2[2], c2[3] * c2[3]);
    const double rs2 = rsqrt_pos(fmax(n2, kFloor));
    const double norm2 = n2 * rs2;
    const double r22 = c2[2] >= 0 ? -norm2 : norm2;
    const double h2 = c2[2] - r22;
    const double beta2 = rcp_pos(fmax(fma(-r22, c2[2], n2), kFloor));
    const double s2 = beta2 * fma(h2, c3[2], c2[3] * c3[3]);
    f.r23 = fma(-s2, h2, c3[2]);
    const double r33 = fma(-s2, c2[3], c3[3]);
    // The first three pivots are -+ the column norms, whose reciprocal square roots exist
    // already.  An exactly singular system (noise-free correspondences) has r33 = 0: a last
    // pivot below 1e-15 of the first one is replaced by that, which leaves the direction the
    // solves amplify alone.
    const double floor_r = norm0 * 1e-15 + 1e-300;
    f.i0 = c0[0] >= 0 ? -rs0 : rs0;
    f.i1 = c1[1] >= 0 ? -rs1 : rs1;
    f.i2 = c2[2] >= 0 ? -rs2 : rs2;
    const double m3 = fabs(r33);
    const double ri3 = rcp_pos(m3 > floor_r ? m3 : floor_r);
    f.i3 = r33 < 0 ? -ri3 : ri3;
    f.pivots_ok = norm0 > floor_r && norm1 > floor_r && norm2 > floor_r;
}

// (R^T R) x = b
__device__ __forceinline__ void tri_solve(const TriFactor& f, const double (&b)[4], double (&x)[4]) {
    const double z0 = b[0] * f.i0;
    const double z1 = fma(-f.r01, z0, b[1]) * f.i1;
    const double z2 = fma(-f.r12, z1, fma(-f.r02, z0, b[2])) * f.i2;
    const double z3 = fma(-f.r23, z2, fma(-f.r13, z1, fma(-f.r03, z0, b[3]))) * f.i3;
    x[3] = z3 * f.i3;
    x[2] = fma(-f.r23, x[3], z2) * f.i2;
    x[1] = fma(-f.r13, x[3], fma(-f.r12, x[2], z1)) * f.i1;
    x[0] = fma(-f.r03, x[3], fma(-f.r02, x[2], fma(-f.r01, x[1], z0))) * f.i0;
}

// y scaled to unit norm with the sign that makes it closest to `ref`; returns the move
__device__ __forceinline__ double unit_towards(double (&y)[4], const double (&ref)[4], bool& finite) {
    const double n2 = fma(y[0], y[0], fma(y[1], y[1], fma(y[2], y[2], y[3] * y[3])));
    finite = n2 > 0 && n2 < 1e300;
    const double dot = fma(y[0], ref[0], fma(y[1], ref[1], fma(y[2], ref[2], y[3] * ref[3])));
    double inv = rsqrt_pos(finite ? n2 : 1.0);
    inv = dot < 0 ? -inv : inv;
    double delta = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        y[k] *= inv;
        delta = fmax(delta, fabs(y[k] - ref[k]));
    }
    return delta;
}

// Fixed cost: the factor, three solves, and the acceptance test of null_vector_invit3 (with a
// fixed factor the contamination shrinks by the same q per solve: the last two moves give q and
// the error of the result as m3 q; accepted when that is below 1e-12).
__device__ __forceinline__ bool null_vector_qr3(const double (&At)[4][4], double (&v)[4]) {
    TriFactor f;
    qr_factor(At, f);
    const double x0[4] = {0.1, 0.2, 0.9, 0.3};
    double x1[4], x2[4], x3[4];
    bool f1, f2, f3;
    tri_solve(f, x0, x1);
    (void)unit_towards(x1, x0, f1);
    tri_solve(f, x1, x2);
    const double m2 = unit_towards(x2, x1, f2);
    tri_solve(f, x2, x3);
    const double m3 = unit_towards(x3, x2, f3);
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = x3[k];
    return f.pivots_ok && f1 && f2 && f3 && (m3 <= 1e-13 || m3 * m3 < 1e-12 * m2);
}

// The deferred points: the same iteration continued for as long as it takes, for systems whose
// two smallest singular values are a factor 2 or more apart (contraction q <= 1/4 per solve, at
// most kMaxTriIt solves).  Returns false for a system without such a gap (q above 0.3 after
// the third solve, the mismatched correspondences) or one that has not settled: the caller
// then takes the shifted iteration on A^T A and, after that, the Jacobi SVD.
constexpr int kMaxTriIt = 24;
__device__ __forceinline__ bool null_vector_qr_iter(const double (&At)[4][4], double (&v)[4], bool live) {
    TriFactor f;
    qr_factor(At, f);
    double x[4] = {0.1, 0.2, 0.9, 0.3};
    double prev_move = 1.0;
    bool done = !live, good = false;
    for (int it = 0; it < kMaxTriIt; ++it) {
        double y[4];
        bool finite;
        tri_solve(f, x, y);
        const double move = unit_towards(y, x, finite);
        if (!done) {
#pragma unroll
            for (int k = 0; k < 4; ++k) x[k] = y[k];
            const bool settled = it >= 2 && (move <= 1e-13 || move * move < 1e-12 * prev_move);
            const bool hopeless = !finite || !f.pivots_ok || (it >= 2 && move > 0.3 * prev_move && move > 1e-13);
            good = settled && finite;
            done = settled || hopeless;
            prev_move = move;
        }
        if (__all(done)) break;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = x[k];
    return good;
}

// Division by w, reprojection, mask and colour for correspondence i (reconstruction.cpp:6-39).
__device__ __forceinline__ void finish_point(const TriArgs& a, int i, const double (&v)[4], double x2,
                                             double y2) {
    const double scale = v[3] != 0 ? 1. / v[3] : 1.;
    const double Xw = v[0] * scale, Yw = v[1] * scale, Zw = v[2] * scale;

    double x = fma(a.R[0], Xw, fma(a.R[1], Yw, fma(a.R[2], Zw, a.t[0])));
    double y = fma(a.R[3], Xw, fma(a.R[4], Yw, fma(a.R[5], Zw, a.t[1])));
    double z = fma(a.R[6], Xw, fma(a.R[7], Yw, fma(a.R[8], Zw, a.t[2])));
    z = z != 0 ? 1. / z : 1.;
    x *= z;
    y *= z;
    const double u = fma(x, a.fx, a.cx), w = fma(y, a.fy, a.cy);
    const double du = u - x2, dv = w - y2;
    const double e = sqrt(fma(du, du, dv * dv));
    const double depth = fma(a.Pc2[0], Xw, fma(a.Pc2[1], Yw, fma(a.Pc2[2], Zw, a.Pc2[3])));

    a.X[3 * (size_t)i] = Xw;
    a.X[3 * (size_t)i + 1] = Yw;
    a.X[3 * (size_t)i + 2] = Zw;
    a.mask[i] = (uint8_t)(e < a.max_err && depth > a.min_dist && depth < a.max_dist);
    if (a.err) a.err[i] = e;
    if (a.bgr) {
        long px = __double2ll_rn(x2), py = __double2ll_rn(y2);
        px = px < 0 ? 0 : (px > a.img_w - 1 ? a.img_w - 1 : px);
        py = py < 0 ? 0 : (py > a.img_h - 1 ? a.img_h - 1 : py);
        const uint8_t* c = a.bgr + (size_t)py * a.img_stride + (size_t)px * 3;
        a.rgb[3 * (size_t)i] = c[0];
        a.rgb[3 * (size_t)i + 1] = c[1];
        a.rgb[3 * (size_t)i + 2] = c[2];
    }
}

// Every point by one-sided Jacobi.
__global__ __launch_bounds__(256) void triangulate_dlt_kernel(TriArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < live_count(a);
    float2 p1 = make_float2(0.f, 0.f), p2 = make_float2(1.f, 1.f);
    if (live) load_points(a, i, p1, p2);
    double At[4][4], v[4];
    build_system<false>(a, p1.x, p1.y, p2.x, p2.y, At);
    null_vector_jacobi(At, v);
    if (live) finish_point(a, i, v, p2.x, p2.y);
}

// Deferred points are appended to one of kListBins lists (bin = workgroup index modulo
// kListBins, bin b at list + b * bin_cap): one atomic per wave, spread over many addresses
// (a single counter would serialise tens of thousands of atomics).
constexpr int kListBins = 1024;
static_assert(kListBins == kStatsTriBinCount, "the context keeps one counter per bin");

// Fast path; unsettled points go to the lists.  kNormalEq: the factorisation of A^T A
// (SFM_TRIANGULATE_NORMAL_EQ=1, a measurement aid) instead of the triangular factor of A.
template <bool kNormalEq>
__global__ __launch_bounds__(256) void triangulate_invit_kernel(TriArgs a, int* bin_count, int* list,
                                                                int bin_cap) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < live_count(a);
    float2 p1 = make_float2(0.f, 0.f), p2 = make_float2(1.f, 1.f);
    if (live) load_points(a, i, p1, p2);
    double At[4][4], v[4];
    build_system<true>(a, p1.x, p1.y, p2.x, p2.y, At);
    const bool ok = kNormalEq ? null_vector_invit3(At, v) : null_vector_qr3(At, v);
    if (live && ok) finish_point(a, i, v, p2.x, p2.y);
    const bool defer = live && !ok;
    const unsigned long long votes = __ballot(defer);
    if (votes) {
        const int bin = blockIdx.x % kListBins;
        const int lane = __lane_id();
        const int leader = __ffsll((long long)votes) - 1;
        int base = 0;
        if (lane == leader) base = atomicAdd(bin_count + bin, __popcll(votes));
        base = __shfl(base, leader);
        if (defer) list[(size_t)bin * bin_cap + base + __popcll(votes & ((1ull << lane) - 1))] = i;
    }
}

// The listed points, workgroup b taking bin b (exits at once when its bin is empty): the
// iteration on the triangular factor continued (not with kNormalEq), then adaptively shifted
// inverse iteration on A^T A for the systems without a clear smallest singular value, and
// one-sided Jacobi, the reference's own algorithm, for the points that does not settle either.
// reset: the workgroup leaves its bin's counter at zero for the next call.
template <bool kNormalEq>
__global__ __launch_bounds__(256) void triangulate_list_kernel(TriArgs a, int* bin_count, const int* list,
                                                               int bin_cap, bool reset) {
    const int count = bin_count[blockIdx.x];
    if (reset) {
        __syncthreads();   // every wave has read the count
        if (threadIdx.x == 0) bin_count[blockIdx.x] = 0;
    }
    const int* mine = list + (size_t)blockIdx.x * bin_cap;
    for (int k0 = 0; k0 < count; k0 += blockDim.x) {  // workgroup-uniform trip count: the solvers vote
        const int k = k0 + threadIdx.x;
        const bool live = k < count;
        const int i = live ? mine[k] : mine[0];
        float2 p1, p2;
        load_points(a, i, p1, p2);
        double At[4][4], v[4];
        build_system<false>(a, p1.x, p1.y, p2.x, p2.y, At);
        bool ok = !kNormalEq && null_vector_qr_iter(At, v, live);
        if (__any(live && !ok)) {  // wave-uniform: the solver below votes on its loop
            double vi[4];
            const bool ok2 = null_vector_invit(At, vi);
            if (!ok) {
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = vi[c];
                ok = ok2;
            }
        }
        if (__any(live && !ok)) {  // wave-uniform: the Jacobi sweep loop votes too
            double vj[4];
            null_vector_jacobi(At, vj);
            if (!ok) {
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = vj[c];
            }
        }
        if (live) finish_point(a, i, v, p2.x, p2.y);
    }
}

void mul_3x3_3x4(const double* K, const double* pose, double* P) {
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 4; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += K[i * 3 + k] * pose[k * 4 + j];
            P[i * 4 + j] = s;
        }
}

}  // namespace

}  // namespace sfm

using namespace sfm;

extern "C" {

namespace {

// Fills the camera part of the kernel arguments (reconstruction.cpp:68 and :13-14).
void set_cameras(TriArgs& a, const double* K, const double* prev_pose, const double* curr_pose,
                 const double* R, const double* t) {
    mul_3x3_3x4(K, prev_pose, a.P1);
    mul_3x3_3x4(K, curr_pose, a.P2);
    double Rt[12], Pc[12];
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) Rt[i * 4 + j] = R[i * 3 + j];
        Rt[i * 4 + 3] = t[i];
    }
    mul_3x3_3x4(K, Rt, Pc);
    for (int i = 0; i < 9; i++) a.R[i] = R[i];
    for (int i = 0; i < 3; i++) a.t[i] = t[i];
    for (int i = 0; i < 4; i++) a.Pc2[i] = Pc[8 + i];
    a.fx = K[0];
    a.fy = K[4];
    a.cx = K[2];
    a.cy = K[5];
}

struct TriLists;
int run_triangulation(sfm_ctx* ctx, TriArgs& a, const TriLists* fixed);
int point_list_args(sfm_ctx* ctx, const double* K, const double* prev_pose, const double* curr_pose,
                    const double* R, const double* t, const float* d_prev_pts, const float* d_curr_pts, int n,
                    const uint8_t* d_bgr, int img_w, int img_h, size_t img_stride, float min_dist, float max_dist,
                    float max_proj_err, double* d_X, uint8_t* d_rgb, uint8_t* d_mask, double* d_err, TriArgs* out,
                    bool* nothing_to_do);

// ---- cv::recoverPose -------------------------------------------------------------------------
// (Tracking::findCameraPose, Source Files/tracking.cpp:110): decomposeEssentialMat gives two
// rotations and a translation direction; each of the four (R, +-t) is tested by triangulating
// every correspondence (the 4x4 system and Jacobi SVD of cv::triangulatePoints, on points
// normalised by K, cameras [I|0] and [R|t]) and counting the points in front of both cameras
// and nearer than distanceThresh; the candidate with the most such points wins.

struct PoseArgs {
    const double* E;        // [problem][E_stride] doubles, the first nine used
    size_t E_stride;
    const int32_t* found;   // per problem: exactly 1 to proceed (nullptr: all proceed)
    const float2* pix1;
    const float2* pix2;
    size_t stride;
    const int32_t* counts;
    int count_uniform;
    int n_problems;
    double fx, fy, cx, cy, dist;
    double max_first_residual;  // > 0: the reference's check of the first correspondence
    double* cand;           // [problem][4][12]: [R | t] row-major
    int32_t* ok;            // [problem]
    int32_t* good;          // [problem][4]
    uint8_t* bits;          // [problem][stride]: bit k set = passes candidate k
    double* R;              // [problem][9]
    double* t;              // [problem][3]
    uint8_t* mask;          // [problem][stride] in/out
    int32_t* n_good;        // [problem]
};

__device__ __forceinline__ int pose_count(const PoseArgs& a, int p) {
    const int c = a.counts ? a.counts[p] : a.count_uniform;
    return c < 0 ? 0 : (c > (int)a.stride ? (int)a.stride : c);
}

__device__ double det3(const double* m) {
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}

// One-sided Jacobi SVD of a 3x3 matrix (row-major a): a = u diag(w) v^T, w descending.
__device__ void svd3(const double* a, double* u, double* w, double* v) {
    double at[3][3], vt[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) at[i][j] = a[3 * j + i];
    for (int sweep = 0; sweep < 30; ++sweep) {
        bool changed = false;
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
                changed = true;
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
    // a rank-2 matrix leaves the last column of u undetermined: complete it
    if (!(w[2] > DBL_EPSILON * w[0])) {
        u[2] = u[3] * u[7] - u[6] * u[4];
        u[5] = u[6] * u[1] - u[0] * u[7];
        u[8] = u[0] * u[4] - u[3] * u[1];
    }
}

// cv::decomposeEssentialMat and the four candidates; the reference's residual test.
__global__ void pose_decompose_kernel(PoseArgs a) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= a.n_problems) return;
    for (int k = 0; k < 4; ++k) a.good[4 * p + k] = 0;
    const int count = pose_count(a, p);
    // with the reference's checks on (max_first_residual > 0) also its early return for five
    // points or fewer (Source Files/tracking.cpp:92)
    int ok = (!a.found || a.found[p] == 1) && count > (a.max_first_residual > 0 ? 5 : 0);
    const double* E = a.E + (size_t)p * a.E_stride;
    if (ok && a.max_first_residual > 0) {
        // | p1^T K^-T E K^-1 p0 | of the first correspondence (Source Files/tracking.cpp:99-107)
        const float2 u0 = a.pix1[(size_t)p * a.stride], u1 = a.pix2[(size_t)p * a.stride];
        const double x0[3] = {((double)u0.x - a.cx) / a.fx, ((double)u0.y - a.cy) / a.fy, 1.0};
        const double x1[3] = {((double)u1.x - a.cx) / a.fx, ((double)u1.y - a.cy) / a.fy, 1.0};
        double s = 0;
        for (int r = 0; r < 3; ++r) s += x1[r] * (E[3 * r] * x0[0] + E[3 * r + 1] * x0[1] + E[3 * r + 2] * x0[2]);
        ok = !(fabs(s) > a.max_first_residual);
    }
    a.ok[p] = ok;
    if (!ok) return;
    double u[9], w[3], v[9];
    svd3(E, u, w, v);
    if (det3(u) < 0)
        for (int i = 0; i < 9; ++i) u[i] = -u[i];
    if (det3(v) < 0)
        for (int i = 0; i < 9; ++i) v[i] = -v[i];
    double uw[9], uwt[9];  // U W and U W^T, W = [0 1 0; -1 0 0; 0 0 1]
    for (int r = 0; r < 3; ++r) {
        uw[3 * r + 0] = -u[3 * r + 1];
        uw[3 * r + 1] = u[3 * r + 0];
        uw[3 * r + 2] = u[3 * r + 2];
        uwt[3 * r + 0] = u[3 * r + 1];
        uwt[3 * r + 1] = -u[3 * r + 0];
        uwt[3 * r + 2] = u[3 * r + 2];
    }
    double* cand = a.cand + (size_t)p * 48;
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) {
            double s1 = 0, s2 = 0;
            for (int k = 0; k < 3; ++k) s1 += uw[3 * r + k] * v[3 * c + k], s2 += uwt[3 * r + k] * v[3 * c + k];
            cand[0 * 12 + 4 * r + c] = cand[2 * 12 + 4 * r + c] = s1;  // R1 with t, R1 with -t
            cand[1 * 12 + 4 * r + c] = cand[3 * 12 + 4 * r + c] = s2;  // R2 with t, R2 with -t
        }
        const double tr = u[3 * r + 2];
        cand[0 * 12 + 4 * r + 3] = cand[1 * 12 + 4 * r + 3] = tr;
        cand[2 * 12 + 4 * r + 3] = cand[3 * 12 + 4 * r + 3] = -tr;
    }
}

// One lane per correspondence: the four cheirality tests.
__global__ __launch_bounds__(256) void pose_cheirality_kernel(PoseArgs a) {
    __shared__ int s_good[4];
    const int p = blockIdx.y, tid = threadIdx.x;
    if (!a.ok[p]) return;
    if (tid < 4) s_good[tid] = 0;
    __syncthreads();
    const int count = pose_count(a, p);
    const double* cand = a.cand + (size_t)p * 48;
    int mine[4] = {0, 0, 0, 0};
    for (int i = blockIdx.x * blockDim.x + tid; i < count; i += gridDim.x * blockDim.x) {
        const size_t k = (size_t)p * a.stride + i;
        const float2 u1 = a.pix1[k], u2 = a.pix2[k];
        const double x1 = ((double)u1.x - a.cx) / a.fx, y1 = ((double)u1.y - a.cy) / a.fy;
        const double x2 = ((double)u2.x - a.cx) / a.fx, y2 = ((double)u2.y - a.cy) / a.fy;
        const bool counted = a.mask[k] != 0;
        unsigned bits = 0;
        for (int c = 0; c < 4; ++c) {
            const double* P = cand + 12 * c;
            double At[4][4], q[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const double p0_row2 = j == 2 ? 1.0 : 0.0;  // [I | 0]
                At[j][0] = x1 * p0_row2 - (j == 0 ? 1.0 : 0.0);
                At[j][1] = y1 * p0_row2 - (j == 1 ? 1.0 : 0.0);
                At[j][2] = x2 * P[8 + j] - P[j];
                At[j][3] = y2 * P[8 + j] - P[4 + j];
            }
            // the null vector as in the triangulation kernels: the fixed-cost inverse iteration,
            // for the lanes it does not settle the adaptive one, then one-sided Jacobi (what
            // cv::triangulatePoints runs); the slower solvers only run when a lane of the wave
            // needs them
            bool solved = null_vector_invit3(At, q);
            if (__any(!solved)) {
                double q2[4];
                const bool ok2 = null_vector_invit(At, q2);
                if (!solved) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) q[j] = q2[j];
                    solved = ok2;
                }
                if (__any(!solved)) {
                    null_vector_jacobi(At, q2);
                    if (!solved) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) q[j] = q2[j];
                    }
                }
            }
            bool pass = q[2] * q[3] > 0;
            const double X = q[0] / q[3], Y = q[1] / q[3], Z = q[2] / q[3];
            pass = pass && Z < a.dist;
            const double z2 = P[8] * X + P[9] * Y + P[10] * Z + P[11] * 1.0;
            pass = pass && z2 > 0 && z2 < a.dist;
            bits |= pass ? 1u << c : 0u;
            mine[c] += pass && counted;
        }
        a.bits[k] = (uint8_t)bits;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        int v = mine[c];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if ((tid & 63) == 0 && v) atomicAdd(&s_good[c], v);
    }
    __syncthreads();
    if (tid < 4 && s_good[tid]) atomicAdd(a.good + 4 * p + tid, s_good[tid]);
}

// The winner (first of the best, in OpenCV's order), its R and t, the mask.
__global__ __launch_bounds__(256) void pose_pick_kernel(PoseArgs a) {
    const int p = blockIdx.x, tid = threadIdx.x;
    const int count = pose_count(a, p);
    uint8_t* mask = a.mask + (size_t)p * a.stride;
    if (!a.ok[p]) {
        if (tid == 0) a.n_good[p] = 0;
        return;
    }
    const int g0 = a.good[4 * p], g1 = a.good[4 * p + 1], g2 = a.good[4 * p + 2], g3 = a.good[4 * p + 3];
    int pick;
    if (g0 >= g1 && g0 >= g2 && g0 >= g3) pick = 0;
    else if (g1 >= g0 && g1 >= g2 && g1 >= g3) pick = 1;
    else if (g2 >= g0 && g2 >= g1 && g2 >= g3) pick = 2;
    else pick = 3;
    const double* P = a.cand + (size_t)p * 48 + 12 * pick;
    if (tid < 9) a.R[(size_t)p * 9 + tid] = P[4 * (tid / 3) + tid % 3];
    if (tid < 3) a.t[(size_t)p * 3 + tid] = P[4 * tid + 3];
    if (tid == 0) a.n_good[p] = a.good[4 * p + pick];
    const uint8_t* bits = a.bits + (size_t)p * a.stride;
    for (int i = tid; i < count; i += blockDim.x) mask[i] = ((bits[i] >> pick) & 1) ? mask[i] : 0;
}

}  // namespace

extern "C" size_t sfm_recover_pose_workspace(int n_problems, size_t problem_stride) {
    if (n_problems <= 0 || problem_stride == 0) return 0;
    Carver c(nullptr);
    c.take<double>((size_t)n_problems * 48);
    c.take<int32_t>((size_t)n_problems);
    c.take<int32_t>((size_t)n_problems * 4);
    c.take<uint8_t>((size_t)n_problems * problem_stride);
    return c.total();
}

extern "C" int sfm_recover_pose_dev(sfm_ctx* ctx, const double* d_E, size_t E_stride, const int32_t* d_found,
                                    const float* d_pts1, const float* d_pts2, size_t problem_stride,
                                    const int32_t* d_counts, int count_uniform, int n_problems, const double* K,
                                    double distance_thresh, double max_first_residual, double* d_R, double* d_t,
                                    uint8_t* d_mask_io, int32_t* d_good) {
    if (!ctx) return set_error(nullptr, SFM_E_INVALID, "ctx is NULL");
    if (n_problems < 0) return set_error(ctx, SFM_E_INVALID, "negative problem count");
    if (n_problems == 0) return SFM_OK;
    if (!d_E || !d_pts1 || !d_pts2 || !K || !d_R || !d_t || !d_mask_io || !d_good)
        return set_error(ctx, SFM_E_INVALID, "NULL pointer argument");
    if (E_stride < 9 || problem_stride == 0 || problem_stride > 0x7fffffff ||
        (!d_counts && (count_uniform < 0 || (size_t)count_uniform > problem_stride)))
        return set_error(ctx, SFM_E_INVALID, "bad stride or point count");
    if (!(K[0] != 0) || !(K[4] != 0)) return set_error(ctx, SFM_E_INVALID, "camera matrix without focal lengths");
    SFM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    // the essential-matrix stage may have left its result in the workspace's front: work behind it
    void* ws;
    SFM_RETURN_IF(workspace(ctx, sfm_recover_pose_workspace(n_problems, problem_stride), &ws));
    Carver c(ws);
    PoseArgs a{};
    a.E = d_E;
    a.E_stride = E_stride;
    a.found = d_found;
    a.pix1 = reinterpret_cast<const float2*>(d_pts1);
    a.pix2 = reinterpret_cast<const float2*>(d_pts2);
    a.stride = problem_stride;
    a.counts = d_counts;
    a.count_uniform = count_uniform;
    a.n_problems = n_problems;
    a.fx = K[0];
    a.fy = K[4];
    a.cx = K[2];
    a.cy = K[5];
    a.dist = distance_thresh;
    a.max_first_residual = max_first_residual;
    a.cand = c.take<double>((size_t)n_problems * 48);
    a.ok = c.take<int32_t>((size_t)n_problems);
    a.good = c.take<int32_t>((size_t)n_problems * 4);
    a.bits = c.take<uint8_t>((size_t)n_problems * problem_stride);
    a.R = d_R;
    a.t = d_t;
    a.mask = d_mask_io;
    a.n_good = d_good;
    KernelTimer timer(ctx, KID_POSE);
    hipLaunchKernelGGL(pose_decompose_kernel, dim3((n_problems + 63) / 64), dim3(64), 0, ctx->stream, a);
    SFM_HIP_CHECK(ctx, hipGetLastError());
    hipLaunchKernelGGL(pose_cheirality_kernel, dim3((unsigned)std::min<size_t>((problem_stride + 255) / 256, 256), n_problems),
                       dim3(256), 0, ctx->stream, a);
    SFM_HIP_CHECK(ctx, hipGetLastError());
    hipLaunchKernelGGL(pose_pick_kernel, dim3(n_problems), dim3(256), 0, ctx->stream, a);
    SFM_HIP_CHECK(ctx, hipGetLastError());
    return SFM_OK;
}

extern "C" int sfm_recover_pose(sfm_ctx* ctx, const double* E, const float* pts1, const float* pts2, int n,
                                const double* K, double* R, double* t, uint8_t* mask_io, int* good) {
    if (!ctx) return set_error(nullptr, SFM_E_INVALID, "ctx is NULL");
    if (good) *good = 0;
    if (n < 0) return set_error(ctx, SFM_E_INVALID, "negative point count");
    if (!E || !K || !R || !t || (n > 0 && (!pts1 || !pts2 || !mask_io)))
        return set_error(ctx, SFM_E_INVALID, "NULL pointer argument");
    if (n == 0) return SFM_OK;
    SFM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    Carver sz(nullptr);
    sz.take<float>(2 * (size_t)n);
    sz.take<float>(2 * (size_t)n);
    sz.take<double>(9 + 9 + 3);
    sz.take<uint8_t>((size_t)n);
    sz.take<int32_t>(4);
    void* st;
    SFM_RETURN_IF(staging(ctx, sz.total(), &st));
    Carver c(st);
    float* d1 = c.take<float>(2 * (size_t)n);
    float* d2 = c.take<float>(2 * (size_t)n);
    double* dE = c.take<double>(9 + 9 + 3);
    uint8_t* dm = c.take<uint8_t>((size_t)n);
    int32_t* dg = c.take<int32_t>(4);
    hipStream_t s = ctx->stream;
    SFM_HIP_CHECK(ctx, hipMemcpyAsync(d1, pts1, sizeof(float) * 2 * n, hipMemcpyHostToDevice, s));
    SFM_HIP_CHECK(ctx, hipMemcpyAsync(d2, pts2, sizeof(float) * 2 * n, hipMemcpyHostToDevice, s));
    SFM_HIP_CHECK(ctx, hipMemcpyAsync(dE, E, sizeof(double) * 9, hipMemcpyHostToDevice, s));
    SFM_HIP_CHECK(ctx, hipMemcpyAsync(dm, mask_io, (size_t)n, hipMemcpyHostToDevice, s));
    SFM_RETURN_IF(sfm_recover_pose_dev(ctx, dE, 9, nullptr, d1, d2, (size_t)n, nullptr, n, 1, K, 50.0, 0.0, dE + 9,
                                       dE + 18, dm, dg));
    double h[12];
    int32_t h_good = 0;
    SFM_HIP_CHECK(ctx, hipMemcpyAsync(h, dE + 9, sizeof(h), hipMemcpyDeviceToHost, s));
    SFM_HIP_CHECK(ctx, hipMemcpyAsync(mask_io, dm, (size_t)n, hipMemcpyDeviceToHost, s));
    SFM_HIP_CHECK(ctx, hipMemcpyAsync(&h_good, dg, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    SFM_HIP_CHECK(ctx, hipStreamSynchronize(s));
    for (int i = 0; i < 9; ++i) R[i] = h[i];
    for (int i = 0; i < 3; ++i) t[i] = h[9 + i];
    if (good) *good = h_good;
    return SFM_OK;
}

int sfm_triangulate_dev(sfm_ctx* ctx, const double* K, const double* prev_pose,
                        const double* curr_pose, const double* R, const double* t,
                        const float* d_prev_pts, const float* d_curr_pts, int n,
                        const uint8_t* d_bgr, int img_w, int img_h, size_t img_stride,
                        float min_dist, float max_dist, float max_proj_err, double* d_X,
                        uint8_t* d_rgb, uint8_t* d_mask, double* d_err) {
    TriArgs a{};
    bool nothing_to_do = false;
    SFM_RETURN_IF(point_list_args(ctx, K, prev_pose, curr_pose, R, t, d_prev_pts, d_curr_pts, n, d_bgr, img_w,
                                  img_h, img_stride, min_dist, max_dist, max_proj_err, d_X, d_rgb, d_mask, d_err,
                                  &a, &nothing_to_do));
    if (nothing_to_do) return SFM_OK;  // reconstruction.cpp:42
    return run_triangulation(ctx, a, nullptr);
}

int sfm_triangulate_matches_dev(sfm_ctx* ctx, const double* K, const double* prev_pose,
                                const double* curr_pose, const double* R, const double* t,
                                const sfm_dmatch* d_matches, const int32_t* d_count, int cap,
                                const void* d_prev_keypts, size_t prev_keypt_stride,
                                const void* d_curr_keypts, size_t curr_keypt_stride,
                                const uint8_t* d_bgr, int img_w, int img_h, size_t img_stride,
                                float min_dist, float max_dist, float max_proj_err, double* d_X,
                                uint8_t* d_rgb, uint8_t* d_mask, double* d_err) {
    if (!ctx) return set_error(nullptr, SFM_E_INVALID, "ctx is NULL");
    if (cap < 0) return set_error(ctx, SFM_E_INVALID, "negative match capacity");
    if (cap == 0) return SFM_OK;
    if (!K || !prev_pose || !curr_pose || !R || !t || !d_matches || !d_prev_keypts || !d_curr_keypts ||
        !d_X || !d_mask)
        return set_error(ctx, SFM_E_INVALID, "NULL pointer argument");
    if (prev_keypt_stride < 8 || curr_keypt_stride < 8 || (prev_keypt_stride & 3) || (curr_keypt_stride & 3))
        return set_error(ctx, SFM_E_INVALID, "key point stride must be a multiple of 4, at least 8");
    if (d_bgr && (!d_rgb || img_w <= 0 || img_h <= 0 || img_stride < (size_t)img_w * 3))
        return set_error(ctx, SFM_E_INVALID, "bad colour image description");
    SFM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    TriArgs a{};
    set_cameras(a, K, prev_pose, curr_pose, R, t);
    a.matches = d_matches;
    a.kp_prev = static_cast<const uint8_t*>(d_prev_keypts);
    a.kp_curr = static_cast<const uint8_t*>(d_curr_keypts);
    a.kp_prev_stride = prev_keypt_stride;
    a.kp_curr_stride = curr_keypt_stride;
    a.count = d_count;
    a.n = cap;
    a.bgr = d_bgr;
    a.img_w = img_w;
    a.img_h = img_h;
    a.img_stride = img_stride;
    a.min_dist = (double)min_dist;
    a.max_dist = (double)max_dist;
    a.max_err = (double)max_proj_err;
    a.X = d_X;
    a.rgb = d_rgb;
    a.mask = d_mask;
    a.err = d_err;
    return run_triangulation(ctx, a, nullptr);
}

}  // extern "C"

namespace {

// Scratch of the entry point that allocates nothing (sfm_triangulate_noalloc_dev): the bin
// counters followed by the deferred-point lists, sized for kNoAllocMaxPoints correspondences.
// Every GPU the library runs on holds its own copy of this array at its own address until the
// process ends; host code reaches it only through the per-device pointer below, never by name.
constexpr int kNoAllocMaxPoints = 1 << 20;
constexpr int kNoAllocMaxBlocks = kNoAllocMaxPoints / 256;
constexpr int kNoAllocMaxBinCap = (kNoAllocMaxBlocks + kListBins - 1) / kListBins * 256;
constexpr size_t kNoAllocInts = (size_t)kListBins + (size_t)kListBins * kNoAllocMaxBinCap;
static __device__ __align__(256) int scratch[kNoAllocInts];

constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex;
int* g_scratch_ptr[kMaxDevices];        // device address of `scratch` on device d, once known
bool g_scratch_asked[kMaxDevices];      // the lookup ran on device d (it runs at context creation)
hipError_t g_scratch_rc[kMaxDevices];   // what the lookup returned when it did not yield an address

// Lists and counters of one launch pair.  persistent: the counters are zero on entry and the
// second launch leaves them so (the context's own); otherwise they are cleared first.
struct TriLists {
    int* bin_count;
    int* list;
    int bins, bin_cap;
    bool persistent;
};

template <bool kNormalEq>
int launch_triangulation_t(sfm_ctx* ctx, TriArgs& a, const TriLists& l, bool timed) {
    const int blocks = (a.n + 255) / 256;
    if (!l.persistent || ctx->tri_bins_dirty)
        SFM_HIP_CHECK(ctx, hipMemsetAsync(l.bin_count, 0, sizeof(int) * kListBins, ctx->stream));
    if (l.persistent) ctx->tri_bins_dirty = true;   // until both launches are under way
    if (timed) {
        KernelTimer timer(ctx, KID_TRIANGULATE_DLT);
        hipLaunchKernelGGL(triangulate_invit_kernel<kNormalEq>, dim3(blocks), dim3(256), 0, ctx->stream, a,
                           l.bin_count, l.list, l.bin_cap);
    } else {
        hipLaunchKernelGGL(triangulate_invit_kernel<kNormalEq>, dim3(blocks), dim3(256), 0, ctx->stream, a,
                           l.bin_count, l.list, l.bin_cap);
    }
    SFM_HIP_CHECK(ctx, hipGetLastError());
    if (timed) {
        KernelTimer timer(ctx, KID_TRIANGULATE_LIST);
        hipLaunchKernelGGL(triangulate_list_kernel<kNormalEq>, dim3(l.bins), dim3(256), 0, ctx->stream, a,
                           l.bin_count, l.list, l.bin_cap, l.persistent);
    } else {
        hipLaunchKernelGGL(triangulate_list_kernel<kNormalEq>, dim3(l.bins), dim3(256), 0, ctx->stream, a,
                           l.bin_count, l.list, l.bin_cap, l.persistent);
    }
    SFM_HIP_CHECK(ctx, hipGetLastError());
    if (l.persistent) ctx->tri_bins_dirty = false;
    return SFM_OK;
}

int launch_triangulation(sfm_ctx* ctx, TriArgs& a, const TriLists& l, bool timed) {
    static const bool normal_eq = [] {   // measurement aid: the factorisation of A^T A
        const char* e = getenv("SFM_TRIANGULATE_NORMAL_EQ");
        return e && e[0] == '1';
    }();
    return normal_eq ? launch_triangulation_t<true>(ctx, a, l, timed) : launch_triangulation_t<false>(ctx, a, l, timed);
}

// Bins in use and lanes that can land in one bin, for n correspondences.
void list_geometry(int n, TriLists* l) {
    const int blocks = (n + 255) / 256;
    l->bins = blocks < kListBins ? blocks : kListBins;
    l->bin_cap = ((blocks + kListBins - 1) / kListBins) * 256;
}

// fixed == nullptr: the lists are carved from the context's workspace and the launches are
// covered by sfm_ctx_set_profiling; otherwise `fixed` holds the list memory to use and nothing
// is allocated and no event is created.  SFM_TRIANGULATE_JACOBI=1 (a measurement aid) sends
// every point through the one-sided Jacobi kernel, which needs no lists, on either route.
int run_triangulation(sfm_ctx* ctx, TriArgs& a, const TriLists* fixed) {
    const int n = a.n;
    static const bool force_jacobi = [] {
        const char* e = getenv("SFM_TRIANGULATE_JACOBI");
        return e && e[0] == '1';
    }();
    if (force_jacobi) {
        if (fixed) {
            hipLaunchKernelGGL(triangulate_dlt_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, a);
        } else {
            KernelTimer timer(ctx, KID_TRIANGULATE_DLT);
            hipLaunchKernelGGL(triangulate_dlt_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, a);
        }
        SFM_HIP_CHECK(ctx, hipGetLastError());
        return SFM_OK;
    }
    if (fixed) return launch_triangulation(ctx, a, *fixed, false);
    TriLists l;
    list_geometry(n, &l);
    Carver sz(nullptr);
    sz.take<int>((size_t)l.bins * l.bin_cap);
    void* ws;
    SFM_RETURN_IF(workspace(ctx, sz.total(), &ws));
    Carver c(ws);
    // the counters are the context's own: no launch spent on clearing them (4 us of a 0.2 ms call)
    l.bin_count = ctx->stats + kStatsTriBins;
    l.persistent = true;
    l.list = c.take<int>((size_t)l.bins * l.bin_cap);
    return launch_triangulation(ctx, a, l, true);
}

// Argument checks and kernel arguments shared by sfm_triangulate_dev and
// sfm_triangulate_noalloc_dev (two point lists on the device).
int point_list_args(sfm_ctx* ctx, const double* K, const double* prev_pose, const double* curr_pose,
                    const double* R, const double* t, const float* d_prev_pts, const float* d_curr_pts, int n,
                    const uint8_t* d_bgr, int img_w, int img_h, size_t img_stride, float min_dist, float max_dist,
                    float max_proj_err, double* d_X, uint8_t* d_rgb, uint8_t* d_mask, double* d_err, TriArgs* out,
                    bool* nothing_to_do) {
    *nothing_to_do = false;
    if (!ctx) return set_error(nullptr, SFM_E_INVALID, "ctx is NULL");
    if (n < 0) return set_error(ctx, SFM_E_INVALID, "negative point count");
    if (n == 0) {
        *nothing_to_do = true;
        return SFM_OK;
    }
    if (!K || !prev_pose || !curr_pose || !R || !t || !d_prev_pts || !d_curr_pts || !d_X || !d_mask)
        return set_error(ctx, SFM_E_INVALID, "NULL pointer argument");
    if (d_bgr && (!d_rgb || img_w <= 0 || img_h <= 0 || img_stride < (size_t)img_w * 3))
        return set_error(ctx, SFM_E_INVALID, "bad colour image description");
    SFM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    TriArgs& a = *out;
    set_cameras(a, K, prev_pose, curr_pose, R, t);
    a.prev_pts = reinterpret_cast<const float2*>(d_prev_pts);
    a.curr_pts = reinterpret_cast<const float2*>(d_curr_pts);
    a.n = n;
    a.bgr = d_bgr;
    a.img_w = img_w;
    a.img_h = img_h;
    a.img_stride = img_stride;
    a.min_dist = (double)min_dist;
    a.max_dist = (double)max_dist;
    a.max_err = (double)max_proj_err;
    a.X = d_X;
    a.rgb = d_rgb;
    a.mask = d_mask;
    a.err = d_err;
    return SFM_OK;
}

}  // namespace

namespace sfm {

// Called by sfm_ctx_create* with `device` current: looks up this device's copy of the scratch
// array once.  A failure is kept and reported by the entry point that needs the array.
void triangulate_scratch_setup(int device) {
    if (device < 0 || device >= kMaxDevices) return;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (g_scratch_ptr[device]) return;
    int* ptr = nullptr;
    const hipError_t e = hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch));
    g_scratch_asked[device] = true;
    if (e == hipSuccess && ptr) {
        g_scratch_ptr[device] = ptr;
        g_scratch_rc[device] = hipSuccess;
    } else {
        g_scratch_rc[device] = e == hipSuccess ? hipErrorInvalidSymbol : e;
    }
}

}  // namespace sfm

extern "C" {

int sfm_triangulate_noalloc_max_points(void) { return kNoAllocMaxPoints; }

int sfm_triangulate_noalloc_dev(sfm_ctx* ctx, const double* K, const double* prev_pose,
                                const double* curr_pose, const double* R, const double* t,
                                const float* d_prev_pts, const float* d_curr_pts, int n,
                                const uint8_t* d_bgr, int img_w, int img_h, size_t img_stride,
                                float min_dist, float max_dist, float max_proj_err, double* d_X,
                                uint8_t* d_rgb, uint8_t* d_mask, double* d_err) {
    if (ctx && n > kNoAllocMaxPoints)
        return set_error(ctx, SFM_E_INVALID,
                         "sfm_triangulate_noalloc_dev: %d correspondences exceed the fixed scratch "
                         "(at most %d per call; use sfm_triangulate_dev)", n, kNoAllocMaxPoints);
    TriArgs a{};
    bool nothing_to_do = false;
    SFM_RETURN_IF(point_list_args(ctx, K, prev_pose, curr_pose, R, t, d_prev_pts, d_curr_pts, n, d_bgr, img_w,
                                  img_h, img_stride, min_dist, max_dist, max_proj_err, d_X, d_rgb, d_mask, d_err,
                                  &a, &nothing_to_do));
    if (nothing_to_do) return SFM_OK;
    if (ctx->device < 0 || ctx->device >= kMaxDevices)
        return set_error(ctx, SFM_E_UNSUPPORTED, "device %d has no scratch slot (devices 0..%d have one)",
                         ctx->device, kMaxDevices - 1);
    int* ptr;
    bool asked;
    hipError_t why;
    {
        std::lock_guard<std::mutex> lock(g_scratch_mutex);
        ptr = g_scratch_ptr[ctx->device];
        asked = g_scratch_asked[ctx->device];
        why = g_scratch_rc[ctx->device];
    }
    if (!ptr && !asked)
        return set_error(ctx, SFM_E_HIP, "the scratch array was never looked up on device %d "
                         "(the look-up runs in sfm_ctx_create*)", ctx->device);
    if (!ptr)
        return set_error(ctx, SFM_E_HIP, "hipGetSymbolAddress(scratch) failed on device %d: %s",
                         ctx->device, hipGetErrorString(why));
    TriLists l;
    list_geometry(n, &l);
    l.bin_count = ptr;              // cleared by launch_triangulation: the array keeps the
    l.list = ptr + kListBins;       // last call's contents
    l.persistent = false;
    return run_triangulation(ctx, a, &l);
}

}  // extern "C"

extern "C" {

int sfm_triangulate(sfm_ctx* ctx, const double* K, const double* prev_pose,
                    const double* curr_pose, const double* R, const double* t,
                    const float* prev_pts, const float* curr_pts, int n, const uint8_t* bgr,
                    int img_w, int img_h, size_t img_stride, float min_dist, float max_dist,
                    float max_proj_err, double* X, uint8_t* rgb, uint8_t* mask, double* err) {
    if (!ctx) return set_error(nullptr, SFM_E_INVALID, "ctx is NULL");
    if (n < 0) return set_error(ctx, SFM_E_INVALID, "negative point count");
    if (n == 0) return SFM_OK;
    if (!prev_pts || !curr_pts || !X || !mask)
        return set_error(ctx, SFM_E_INVALID, "NULL pointer argument");
    if (bgr && !rgb) return set_error(ctx, SFM_E_INVALID, "rgb is NULL but bgr is given");
    SFM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const size_t img_bytes = bgr ? (size_t)img_h * img_stride : 0;
    Carver sz(nullptr);
    sz.take<float>(2 * (size_t)n);
    sz.take<float>(2 * (size_t)n);
    sz.take<double>(3 * (size_t)n);
    sz.take<double>((size_t)n);
    sz.take<uint8_t>(3 * (size_t)n);
    sz.take<uint8_t>((size_t)n);
    sz.take<uint8_t>(img_bytes);
    void* st;
    SFM_RETURN_IF(staging(ctx, sz.total(), &st));
    Carver c(st);
    float* dp1 = c.take<float>(2 * (size_t)n);
    float* dp2 = c.take<float>(2 * (size_t)n);
    double* dX = c.take<double>(3 * (size_t)n);
    double* de = c.take<double>((size_t)n);
    uint8_t* drgb = c.take<uint8_t>(3 * (size_t)n);
    uint8_t* dmask = c.take<uint8_t>((size_t)n);
    uint8_t* dimg = c.take<uint8_t>(img_bytes);
    SFM_HIP_CHECK(ctx, hipMemcpyAsync(dp1, prev_pts, sizeof(float) * 2 * n, hipMemcpyHostToDevice,
                                      ctx->stream));
    SFM_HIP_CHECK(ctx, hipMemcpyAsync(dp2, curr_pts, sizeof(float) * 2 * n, hipMemcpyHostToDevice,
                                      ctx->stream));
    if (bgr)
        SFM_HIP_CHECK(ctx, hipMemcpyAsync(dimg, bgr, img_bytes, hipMemcpyHostToDevice, ctx->stream));
    SFM_RETURN_IF(sfm_triangulate_dev(ctx, K, prev_pose, curr_pose, R, t, dp1, dp2, n,
                                      bgr ? dimg : nullptr, img_w, img_h, img_stride, min_dist,
                                      max_dist, max_proj_err, dX, drgb, dmask, err ? de : nullptr));
    SFM_HIP_CHECK(ctx, hipMemcpyAsync(X, dX, sizeof(double) * 3 * n, hipMemcpyDeviceToHost,
                                      ctx->stream));
    SFM_HIP_CHECK(ctx, hipMemcpyAsync(mask, dmask, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    if (bgr)
        SFM_HIP_CHECK(ctx, hipMemcpyAsync(rgb, drgb, 3 * (size_t)n, hipMemcpyDeviceToHost,
                                          ctx->stream));
    if (err)
        SFM_HIP_CHECK(ctx, hipMemcpyAsync(err, de, sizeof(double) * n, hipMemcpyDeviceToHost,
                                          ctx->stream));
    SFM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return SFM_OK;
}

}  // extern "C"
