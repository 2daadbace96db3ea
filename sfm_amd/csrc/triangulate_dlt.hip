// Batched two-view DLT triangulation with reprojection / depth filtering.
//
// Replaces Reconstruction::triangulateCloud and pointsToRGBCloud
// (Source Files/reconstruction.cpp:41-73 and :6-39): per correspondence the 4x4
// homogeneous system of cv::triangulatePoints, its right singular vector of the
// smallest singular value (one-sided Jacobi, the algorithm of cv::SVD::compute),
// division by w, reprojection with the separately supplied R, t, and the mask
//   err < maxProjErr && depth > minDist && depth < maxDist.
//
// One lane per correspondence; the 4x4 system, the rotation accumulator V and
// the column norms stay in VGPRs (36 doubles).  Every loop over matrix indices
// is fully unrolled so that no array is indexed at run time.  The sweep loop is
// wave-uniform: a wave leaves it when none of its lanes rotated in a sweep.
// All arithmetic is fp64, as in the reference (points are widened to double by
// Header Files/common.h:47-57 before cv::triangulatePoints sees them).
#include <cfloat>

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
    const float2* prev_pts;
    const float2* curr_pts;
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

__global__ __launch_bounds__(256) void triangulate_dlt_kernel(TriArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < a.n;
    const float2 p1 = live ? a.prev_pts[i] : make_float2(0.f, 0.f);
    const float2 p2 = live ? a.curr_pts[i] : make_float2(1.f, 1.f);
    const double x1 = p1.x, y1 = p1.y, x2 = p2.x, y2 = p2.y;

    double At[4][4], Vt[4][4], W[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        At[k][0] = x1 * a.P1[8 + k] - a.P1[k];
        At[k][1] = y1 * a.P1[8 + k] - a.P1[4 + k];
        At[k][2] = x2 * a.P2[8 + k] - a.P2[k];
        At[k][3] = y2 * a.P2[8 + k] - a.P2[4 + k];
    }
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
    double v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
        v[k] = last == 0 ? Vt[0][k] : last == 1 ? Vt[1][k] : last == 2 ? Vt[2][k] : Vt[3][k];

    const double scale = v[3] != 0 ? 1. / v[3] : 1.;
    const double Xw = v[0] * scale, Yw = v[1] * scale, Zw = v[2] * scale;

    double x = a.R[0] * Xw + a.R[1] * Yw + a.R[2] * Zw + a.t[0];
    double y = a.R[3] * Xw + a.R[4] * Yw + a.R[5] * Zw + a.t[1];
    double z = a.R[6] * Xw + a.R[7] * Yw + a.R[8] * Zw + a.t[2];
    z = z != 0 ? 1. / z : 1.;
    x *= z;
    y *= z;
    const double u = x * a.fx + a.cx, w = y * a.fy + a.cy;
    const double du = u - x2, dv = w - y2;
    const double e = sqrt(du * du + dv * dv);
    const double depth = a.Pc2[0] * Xw + a.Pc2[1] * Yw + a.Pc2[2] * Zw + a.Pc2[3] * 1.0;

    if (!live) return;
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

int sfm_triangulate_dev(sfm_ctx* ctx, const double* K, const double* prev_pose,
                        const double* curr_pose, const double* R, const double* t,
                        const float* d_prev_pts, const float* d_curr_pts, int n,
                        const uint8_t* d_bgr, int img_w, int img_h, size_t img_stride,
                        float min_dist, float max_dist, float max_proj_err, double* d_X,
                        uint8_t* d_rgb, uint8_t* d_mask, double* d_err) {
    if (!ctx) return set_error(nullptr, SFM_E_INVALID, "ctx is NULL");
    if (n < 0) return set_error(ctx, SFM_E_INVALID, "negative point count");
    if (n == 0) return SFM_OK;  // reconstruction.cpp:42
    if (!K || !prev_pose || !curr_pose || !R || !t || !d_prev_pts || !d_curr_pts || !d_X || !d_mask)
        return set_error(ctx, SFM_E_INVALID, "NULL pointer argument");
    if (d_bgr && (!d_rgb || img_w <= 0 || img_h <= 0 || img_stride < (size_t)img_w * 3))
        return set_error(ctx, SFM_E_INVALID, "bad colour image description");
    SFM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    TriArgs a{};
    mul_3x3_3x4(K, prev_pose, a.P1);  // reconstruction.cpp:68
    mul_3x3_3x4(K, curr_pose, a.P2);
    double Rt[12], Pc[12];
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) Rt[i * 4 + j] = R[i * 3 + j];
        Rt[i * 4 + 3] = t[i];
    }
    mul_3x3_3x4(K, Rt, Pc);  // reconstruction.cpp:13-14
    for (int i = 0; i < 9; i++) a.R[i] = R[i];
    for (int i = 0; i < 3; i++) a.t[i] = t[i];
    for (int i = 0; i < 4; i++) a.Pc2[i] = Pc[8 + i];
    a.fx = K[0];
    a.fy = K[4];
    a.cx = K[2];
    a.cy = K[5];
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
    KernelTimer timer(ctx, KID_TRIANGULATE_DLT);
    hipLaunchKernelGGL(triangulate_dlt_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, a);
    SFM_HIP_CHECK(ctx, hipGetLastError());
    return SFM_OK;
}

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
