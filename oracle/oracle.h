/*
 * oracle.h -- CPU restatement of the Lowcash/SfM per-frame hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the timed CPU baseline.
 *
 * PARITY UNPINNED.  The arithmetic of this path lives in OpenCV (knnMatch,
 * SparsePyrLKOpticalFlow::calc, triangulatePoints, projectPoints), which is
 * neither vendored in the reference nor installed here, and the reference
 * ships no tests or golden vectors.  These functions restate the published
 * OpenCV 4.x algorithms (modules/core/src/batch_distance.cpp,
 * modules/features2d/src/matchers.cpp, modules/video/src/lkpyramid.cpp,
 * modules/imgproc/src/pyramids.cpp, modules/calib3d/src/triangulate.cpp,
 * modules/calib3d/src/calibration.cpp, modules/calib3d/src/fundam.cpp,
 * modules/calib3d/src/ptsetreg.cpp, modules/imgproc/src/featureselect.cpp,
 * modules/imgproc/src/corner.cpp, modules/calib3d/src/five-point.cpp) and are anchored on the reference's
 * own call sites, cited per function as "Source Files/<file>:<line>".
 * They are checked against independently written NumPy vectors under
 * tests/golden/, not against OpenCV itself.
 */
#ifndef SFM_ORACLE_H
#define SFM_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Same layout as cv::DMatch {queryIdx, trainIdx, imgIdx, distance}. */
typedef struct {
    int32_t queryIdx;
    int32_t trainIdx;
    int32_t imgIdx;
    float distance;
} orc_dmatch;

enum { ORC_NORM_HAMMING = 0, ORC_NORM_L2 = 1, ORC_NORM_L2SQR = 2 };

/* ---- descriptor matching ------------------------------------------------ */

/* cv::BFMatcher::knnMatch(q, t, out, 2) with NORM_HAMMING, called at
 * Source Files/feature_processing.cpp:121.  idx/dist are [nq][2]; a slot that
 * has no neighbour (nt < 2) holds idx -1 and dist FLT_MAX.  threads<=0 means
 * all cores (OpenMP), 1 means the scalar single-thread path. */
void orc_knn2_hamming(const uint8_t* q, int nq, size_t q_stride,
                      const uint8_t* t, int nt, size_t t_stride, int dbytes,
                      int32_t* idx, float* dist, int threads);

/* Same call site with NORM_L2 (squared == 0) or NORM_L2SQR (squared != 0):
 * Source Files/feature_processing.cpp:111,115,121.  Strides are in floats. */
void orc_knn2_l2(const float* q, int nq, size_t q_stride,
                 const float* t, int nt, size_t t_stride, int dim, int squared,
                 int32_t* idx, float* dist, int threads);

/* The ratio filter of DescriptorMatcher::ratioMaches,
 * Source Files/feature_processing.cpp:123-129.  Returns the match count;
 * out has room for nq entries. */
int orc_ratio_filter(const int32_t* idx, const float* dist, int nq, int nt,
                     float ratio, orc_dmatch* out);

/* The cross-check loop of DescriptorMatcher::findRobustMatches,
 * Source Files/feature_processing.cpp:159-176.  Returns the match count;
 * out has room for nb entries. */
int orc_cross_check(const orc_dmatch* f, int nf, const orc_dmatch* b, int nb,
                    orc_dmatch* out);

/* ratioMaches both ways + cross-check, Source Files/feature_processing.cpp:
 * 148-176 (the fundamental-matrix filter at :187-204 is not part of it).
 * norm is one of ORC_NORM_*; d is bytes (Hamming) or floats (L2). Rows are
 * contiguous.  Returns the match count; out has room for n_curr entries. */
int orc_robust_matches(int norm, const void* prev, int n_prev,
                       const void* curr, int n_curr, int d, float ratio,
                       orc_dmatch* out, int threads);

/* ---- pyramidal Lucas-Kanade -------------------------------------------- */

/* cv::pyrDown, 8-bit single channel, BORDER_REFLECT_101, destination size
 * ((w+1)/2, (h+1)/2); used by cv::buildOpticalFlowPyramid inside
 * SparsePyrLKOpticalFlow::calc (Source Files/feature_processing.cpp:241). */
void orc_pyr_down(const uint8_t* src, int w, int h, size_t src_stride,
                  uint8_t* dst, size_t dst_stride);

/* The Scharr derivative image of lkpyramid.cpp (calcSharrDeriv): interleaved
 * int16 (Ix, Iy) per pixel, dst_stride in int16 elements. */
void orc_scharr_deriv(const uint8_t* src, int w, int h, size_t src_stride,
                      int16_t* dst, size_t dst_stride);

/* Number of the top pyramid level cv::buildOpticalFlowPyramid keeps. */
int orc_lk_num_levels(int w, int h, int win_w, int win_h, int max_level);

/* OptFlow::computeFlow, Source Files/feature_processing.cpp:238-249:
 * SparsePyrLKOpticalFlow::calc (flags 0) followed by status[i] = 0 where
 * err[i] > max_error.  crit_type uses cv::TermCriteria bits (COUNT=1, EPS=2).
 * acc_exact != 0 accumulates the normal-equation sums in int64 (exact)
 * instead of OpenCV's float accumulators; everything else is identical.
 * iters_out (nullable) receives the total number of LK iterations run. */
/* Study aid: per (level, point) iteration counts of the following orc_pyrlk calls go to
 * buf[level * n + i]; NULL switches it off. */
void orc_lk_set_iter_buffer(int32_t* buf);
void orc_pyrlk(const uint8_t* prev, const uint8_t* curr, int w, int h,
               size_t stride, const float* prev_pts, int n, int win_w,
               int win_h, int max_level, int crit_type, int max_count,
               double epsilon, double min_eig_threshold, float max_error,
               float* curr_pts, uint8_t* status, float* err, int acc_exact,
               int threads, int64_t* iters_out);

/* ---- triangulation ----------------------------------------------------- */

/* Reconstruction::triangulateCloud + pointsToRGBCloud, Source Files/
 * reconstruction.cpp:6-73.  K, R row-major 3x3; prev_pose/curr_pose row-major
 * 3x4; pts are interleaved float (x, y).  bgr may be NULL (rgb then untouched).
 * X is [n][3] double, rgb [n][3], mask [n], err (nullable) [n] double. */
void orc_triangulate(const double* K, const double* prev_pose,
                     const double* curr_pose, const double* R, const double* t,
                     const float* prev_pts, const float* curr_pts, int n,
                     const uint8_t* bgr, int img_w, int img_h, size_t img_stride,
                     float min_dist, float max_dist, float max_proj_err,
                     double* X, uint8_t* rgb, uint8_t* mask, double* err,
                     int threads);

/* Singular values (descending) and the right singular vector of the smallest
 * one for a row-major 4x4 matrix: the one-sided Jacobi of cv::SVD::compute. */
void orc_svd4_null(const double* A, double* w, double* v);

/* ---- epipolar filter ------------------------------------------------------ */

/* cv::findFundamentalMat(points1, points2, mask) as called at
 * Source Files/feature_processing.cpp:188 (FM_RANSAC, threshold 3, confidence 0.99, 1000
 * iterations; OpenCV switches to LMedS below 15 points and to a direct 7-point solve at 7).
 * Points are interleaved (x, y) floats.  mask gets `count` entries (untouched below 7 points,
 * as in OpenCV).  F (nullable) receives the winning matrix, iters_run (nullable) the number
 * of sampling iterations executed.  Returns 1 when a matrix was found.  The null space of
 * the 7-point system is taken by elimination, not by OpenCV's SVD (see the source file). */
int orc_find_fundamental(const float* m1, const float* m2, int count, double thresh,
                         double confidence, int max_iters, double* F, uint8_t* mask,
                         int* iters_run);
/* run7Point: up to three 3x3 matrices (row-major, 9 doubles each) for seven correspondences;
 * returns their number. */
int orc_fm_7point(const float* m1, const float* m2, double* fmatrix);
/* The index subsets the sampling loop draws, [max_iters][7]; returns how many were drawn. */
int orc_fm_subsets(const float* m1, const float* m2, int count, int max_iters, int max_attempts,
                   int32_t* idx);
/* RANSACUpdateNumIters */
int orc_ransac_update_num_iters(double p, double ep, int model_points, int max_iters);

/* ---- relative pose --------------------------------------------------------- */

/* EMEstimatorCallback::runKernel of cv::findEssentialMat: five normalised correspondences
 * (x, y doubles) -> up to ten essential matrices (9 doubles each, unit Frobenius norm). */
int orc_em_5point(const double* m1, const double* m2, double* ematrix);

/* The index subsets the registrator draws for `count` points: out is [max_iters][5]. */
int orc_em_subsets(int count, int max_iters, int32_t* out);

/* cv::findEssentialMat(prevPts, currPts, cameraK, method, prob, threshold, mask),
 * Source Files/tracking.cpp:94.  method 0 = RANSAC, 1 = LMEDS.  Returns the number of 3x3
 * matrices written to E_out (room for 10): 1 for a found model, several when count == 5
 * (OpenCV stacks every solution and the reference then rejects the 3k x 3 result), 0 for
 * none.  mask_out (count bytes) is written as OpenCV writes it. */
int orc_find_essential(const float* pts1, const float* pts2, int count, const double* K, int method, double prob,
                       double threshold, int max_iters, double* E_out, uint8_t* mask_out, int* iters_run);

/* cv::decomposeEssentialMat */
void orc_decompose_essential(const double* E, double* R1, double* R2, double* t);

/* cv::recoverPose(E, prevPts, currPts, cameraK, R, t, mask), Source Files/tracking.cpp:110.
 * mask_io: inlier mask in (may be NULL), mask of the chosen pose out.  Returns the count. */
int orc_recover_pose(const double* E, const float* pts1, const float* pts2, int count, const double* K, double* R,
                     double* t, uint8_t* mask_io);

/* | p1^T K^-T E K^-1 p0 | of the first correspondence, Source Files/tracking.cpp:99-107. */
double orc_essential_residual(const double* E, const float* p0, const float* p1, const double* K);

/* ---- corner detection ----------------------------------------------------- */

/* cv::cornerMinEigenVal(image, eig, 3, 3) as goodFeaturesToTrack calls it
 * (Source Files/feature_processing.cpp:93): eig is [h][w] float. */
void orc_min_eigen_val(const uint8_t* img, int w, int h, size_t stride, float* eig);

/* cv::goodFeaturesToTrack(imGray, corners, maxCorners, qualityLevel, minDistance),
 * Source Files/feature_processing.cpp:93.  corners: room for max_out (x, y) pairs; returns
 * the number written; max_corners <= 0 means no limit; n_candidates (optional) receives the
 * number of thresholded local maxima. */
int orc_good_features(const uint8_t* img, int w, int h, size_t stride, int max_corners, double quality,
                      double min_distance, float* corners, int max_out, int* n_candidates);

#ifdef __cplusplus
}
#endif
#endif
