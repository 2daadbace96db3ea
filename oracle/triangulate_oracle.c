/*
 * triangulate_oracle.c -- CPU restatement of two-view triangulation and the
 * reprojection / depth filter.
 *
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see oracle.h).
 *
 * Reference: Reconstruction::triangulateCloud, Source Files/
 * reconstruction.cpp:41-73 (live branch :68-69, the "DLT" branch at :59-66 is
 * dead), and Reconstruction::pointsToRGBCloud, :6-39.  Points are widened to
 * double as Header Files/common.h:47-57 does; K(2,2) == 1 is the convention of
 * Header Files/camera.h:26-27.  cv::undistort with empty distortion
 * (reconstruction.cpp:44-45) is an identity remap of the point list and is
 * treated as identity.  The arithmetic restates OpenCV's
 * cv::triangulatePoints (modules/calib3d/src/triangulate.cpp: one 4x4 system
 * per point, right singular vector of the smallest singular value from
 * cv::SVD::compute = one-sided Jacobi, modules/core/src/lapack.cpp),
 * cv::convertPointsFromHomogeneous and the distortion-free path of
 * cv::projectPoints (modules/calib3d/src/calibration.cpp).
 */
#include "oracle.h"

#include <float.h>
#include <math.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* lapack.cpp JacobiSVDImpl_<double>, m = n = 4.  At holds A transposed (row i
 * of At is column i of A) and is rotated in place; Vt rows end up as the right
 * singular vectors, sorted by descending singular value. */
static void jacobi_svd4(double At[4][4], double W[4], double Vt[4][4]) {
    const double eps = DBL_EPSILON * 10;
    const int n = 4, m = 4, max_iter = 30;
    int i, j, k, iter;

    for (i = 0; i < n; i++) {
        double sd = 0;
        for (k = 0; k < m; k++) sd += At[i][k] * At[i][k];
        W[i] = sd;
        for (k = 0; k < n; k++) Vt[i][k] = 0;
        Vt[i][i] = 1;
    }

    for (iter = 0; iter < max_iter; iter++) {
        int changed = 0;
        for (i = 0; i < n - 1; i++)
            for (j = i + 1; j < n; j++) {
                double* Ai = At[i];
                double* Aj = At[j];
                double a = W[i], p = 0, b = W[j], c, s;
                for (k = 0; k < m; k++) p += Ai[k] * Aj[k];
                if (fabs(p) <= eps * sqrt(a * b)) continue;

                p *= 2;
                double beta = a - b, gamma = hypot(p, beta);
                if (beta < 0) {
                    double delta = (gamma - beta) * 0.5;
                    s = sqrt(delta / gamma);
                    c = p / (gamma * s * 2);
                } else {
                    c = sqrt((gamma + beta) / (gamma * 2));
                    s = p / (gamma * c * 2);
                }

                a = b = 0;
                for (k = 0; k < m; k++) {
                    double t0 = c * Ai[k] + s * Aj[k];
                    double t1 = -s * Ai[k] + c * Aj[k];
                    Ai[k] = t0;
                    Aj[k] = t1;
                    a += t0 * t0;
                    b += t1 * t1;
                }
                W[i] = a;
                W[j] = b;
                changed = 1;

                for (k = 0; k < n; k++) {
                    double t0 = c * Vt[i][k] + s * Vt[j][k];
                    double t1 = -s * Vt[i][k] + c * Vt[j][k];
                    Vt[i][k] = t0;
                    Vt[j][k] = t1;
                }
            }
        if (!changed) break;
    }

    for (i = 0; i < n; i++) {
        double sd = 0;
        for (k = 0; k < m; k++) sd += At[i][k] * At[i][k];
        W[i] = sqrt(sd);
    }
    for (i = 0; i < n - 1; i++) {
        j = i;
        for (k = i + 1; k < n; k++)
            if (W[j] < W[k]) j = k;
        if (i != j) {
            double tw = W[i];
            W[i] = W[j];
            W[j] = tw;
            for (k = 0; k < n; k++) {
                double ta = At[i][k];
                At[i][k] = At[j][k];
                At[j][k] = ta;
                double tv = Vt[i][k];
                Vt[i][k] = Vt[j][k];
                Vt[j][k] = tv;
            }
        }
    }
}

void orc_svd4_null(const double* A, double* w, double* v) {
    double At[4][4], W[4], Vt[4][4];
    for (int i = 0; i < 4; i++)
        for (int k = 0; k < 4; k++) At[i][k] = A[k * 4 + i];
    jacobi_svd4(At, W, Vt);
    for (int i = 0; i < 4; i++) {
        w[i] = W[i];
        v[i] = Vt[3][i];
    }
}

/* cv::Matx33d * cv::Matx34d */
static void mul_3x3_3x4(const double* K, const double* pose, double* P) {
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 4; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += K[i * 3 + k] * pose[k * 4 + j];
            P[i * 4 + j] = s;
        }
}

void orc_triangulate(const double* K, const double* prev_pose, const double* curr_pose,
                     const double* R, const double* t, const float* prev_pts,
                     const float* curr_pts, int n, const uint8_t* bgr, int img_w, int img_h,
                     size_t img_stride, float min_dist, float max_dist, float max_proj_err,
                     double* X, uint8_t* rgb, uint8_t* mask, double* err, int threads) {
    if (n <= 0) return; /* reconstruction.cpp:42 */
    double P1[12], P2[12], Rt[12], Pc[12];
    mul_3x3_3x4(K, prev_pose, P1); /* reconstruction.cpp:68 */
    mul_3x3_3x4(K, curr_pose, P2);
    /* reconstruction.cpp:13-14: K * [R|t] built from the separate R, t */
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) Rt[i * 4 + j] = R[i * 3 + j];
        Rt[i * 4 + 3] = t[i];
    }
    mul_3x3_3x4(K, Rt, Pc);
    const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];

    int nthr = 1;
#ifdef _OPENMP
    nthr = threads <= 0 ? omp_get_max_threads() : threads;
#else
    (void)threads;
#endif
#pragma omp parallel for schedule(static) num_threads(nthr)
    for (int i = 0; i < n; i++) {
        const double x1 = prev_pts[2 * i], y1 = prev_pts[2 * i + 1];
        const double x2 = curr_pts[2 * i], y2 = curr_pts[2 * i + 1];
        double At[4][4], W[4], Vt[4][4];
        /* triangulate.cpp: row 2j = x*P[2,:] - P[0,:], row 2j+1 = y*P[2,:] - P[1,:] */
        for (int k = 0; k < 4; k++) {
            At[k][0] = x1 * P1[8 + k] - P1[k];
            At[k][1] = y1 * P1[8 + k] - P1[4 + k];
            At[k][2] = x2 * P2[8 + k] - P2[k];
            At[k][3] = y2 * P2[8 + k] - P2[4 + k];
        }
        jacobi_svd4(At, W, Vt);
        /* convertPointsFromHomogeneous */
        const double hw = Vt[3][3];
        const double scale = hw != 0 ? 1. / hw : 1.;
        const double Xw = Vt[3][0] * scale, Yw = Vt[3][1] * scale, Zw = Vt[3][2] * scale;
        X[3 * i] = Xw;
        X[3 * i + 1] = Yw;
        X[3 * i + 2] = Zw;

        /* projectPoints, no distortion */
        double x = R[0] * Xw + R[1] * Yw + R[2] * Zw + t[0];
        double y = R[3] * Xw + R[4] * Yw + R[5] * Zw + t[1];
        double z = R[6] * Xw + R[7] * Yw + R[8] * Zw + t[2];
        z = z != 0 ? 1. / z : 1.;
        x *= z;
        y *= z;
        const double u = x * fx + cx, v = y * fy + cy;
        const double du = u - x2, dv = v - y2;
        const double e = sqrt(du * du + dv * dv); /* reconstruction.cpp:22 */
        /* reconstruction.cpp:28-33: third row of K*[R|t] times (X,Y,Z,1) */
        const double depth = Pc[8] * Xw + Pc[9] * Yw + Pc[10] * Zw + Pc[11] * 1.0;
        mask[i] = (uint8_t)(e < (double)max_proj_err && depth > (double)min_dist &&
                            depth < (double)max_dist); /* :36 */
        if (err) err[i] = e;
        if (bgr) {
            /* reconstruction.cpp:20; the reference does not bounds-check, the
             * lookup is clamped to the image here */
            long px = lrint(x2), py = lrint(y2);
            if (px < 0) px = 0;
            if (py < 0) py = 0;
            if (px > img_w - 1) px = img_w - 1;
            if (py > img_h - 1) py = img_h - 1;
            const uint8_t* c = bgr + (size_t)py * img_stride + (size_t)px * 3;
            rgb[3 * i] = c[0];
            rgb[3 * i + 1] = c[1];
            rgb[3 * i + 2] = c[2];
        }
    }
}
