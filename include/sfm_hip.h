/*
 * sfm_hip.h -- C ABI of the MI355X (gfx950) implementation of the Lowcash/SfM
 * per-frame hot path: descriptor matching, pyramidal Lucas-Kanade tracking and
 * two-view DLT triangulation with reprojection filtering.
 *
 * Every entry point replaces one call the reference makes into OpenCV from its
 * wrapper layer; the reference location is cited as
 * "<reference file>:<line>" (paths relative to the reference tree, whose source
 * directories are "Source Files/" and "Header Files/").
 *
 * Conventions
 *   - Plain pointers and sizes only.  No exceptions cross this boundary.
 *   - Every function returns 0 on success, a negative SFM_E_* code otherwise;
 *     sfm_last_error() then describes the failure.
 *   - Functions without a suffix take HOST pointers: they copy inputs to the
 *     GPU, run, copy results back and return when the results are in place.
 *   - Functions ending in _dev take DEVICE pointers, enqueue their work on the
 *     context's stream and return without synchronising; inputs and outputs
 *     must stay valid until sfm_ctx_synchronize() (or the caller's own stream
 *     synchronisation) has returned.
 *   - A context is not thread-safe; use one context per host thread.  The
 *     reference calls this path from its single main-loop thread
 *     (Source Files/app_solver.cpp:151).
 */
#ifndef SFM_HIP_H
#define SFM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SFM_OK 0
#define SFM_E_INVALID (-1)   /* bad argument */
#define SFM_E_HIP (-2)       /* a HIP runtime call failed */
#define SFM_E_NOMEM (-3)     /* device or host allocation failed */
#define SFM_E_UNSUPPORTED (-4)

/* cv::NormTypes the reference selects in DescriptorMatcher::DescriptorMatcher,
 * Source Files/feature_processing.cpp:108-115. */
#define SFM_NORM_HAMMING 0 /* "BRUTEFORCE_HAMMING" */
#define SFM_NORM_L2 1      /* "BRUTEFORCE" (the default branch) */
#define SFM_NORM_L2SQR 2   /* "BRUTEFORCE_SL2" */

/* cv::TermCriteria type bits (Source Files/app_solver.cpp:108). */
#define SFM_TERM_COUNT 1
#define SFM_TERM_EPS 2

/* Same memory layout as cv::DMatch (16 bytes). */
typedef struct sfm_dmatch {
    int32_t queryIdx;
    int32_t trainIdx;
    int32_t imgIdx;
    float distance;
} sfm_dmatch;

/* Compact match record of the exchange between GPUs (Hamming matching of frames with at most
 * 65536 rows): prev row, curr row, Hamming distance.  6 bytes. */
typedef struct sfm_match16 {
    uint16_t queryIdx;
    uint16_t trainIdx;
    uint16_t distance;
} sfm_match16;

typedef struct sfm_ctx sfm_ctx;

/* ---- context ------------------------------------------------------------ */

/* Creates a context on HIP device `device` with a stream of its own. */
int sfm_ctx_create(int device, sfm_ctx** out);
/* As above but enqueues on the caller's hipStream_t (not owned). */
int sfm_ctx_create_on_stream(int device, void* hip_stream, sfm_ctx** out);
void sfm_ctx_destroy(sfm_ctx* ctx);
int sfm_ctx_synchronize(sfm_ctx* ctx);
/* Grows the context's device workspace to at least `bytes` so that later _dev
 * calls needing no more than that do not allocate. */
int sfm_ctx_reserve(sfm_ctx* ctx, size_t bytes);
/* Message of the last failure on this context (or of the last failed
 * sfm_ctx_create when ctx is NULL).  Never NULL. */
const char* sfm_last_error(const sfm_ctx* ctx);
/* "sfm_hip <version> gfx950" */
const char* sfm_version(void);

/* ---- measurement aid (no counterpart in the reference) -------------------
 * With profiling on, every kernel launch of this context is bracketed by HIP
 * events on the context's stream.  sfm_ctx_profile_read synchronises the
 * stream and returns the summed duration and the launch count of one kernel
 * family (ids 0 .. sfm_profile_kernel_count()-1) since profiling was last
 * switched (switching on or off clears the record). */
int sfm_ctx_set_profiling(sfm_ctx* ctx, int enabled);
int sfm_profile_kernel_count(void);
const char* sfm_profile_kernel_name(int kernel_id);
int sfm_ctx_profile_read(sfm_ctx* ctx, int kernel_id, double* total_ms, int* launches);

/* ---- descriptor matching ------------------------------------------------ */

/* Brute-force 2-nearest-neighbour search: replaces
 * matcher->knnMatch(lDesc, rDesc, knnMatches, 2), Source Files/
 * feature_processing.cpp:121.  `norm` is SFM_NORM_*.  For SFM_NORM_HAMMING rows
 * are `d` bytes of CV_8U and strides are in bytes; for the L2 norms rows are
 * `d` floats and strides are in bytes too.  idx and dist are [nq][2]; equal
 * distances keep the lower train index first; a slot without a neighbour
 * (nt < 2) holds idx -1 and dist FLT_MAX.  nt must be below 2^18 (OpenCV's own
 * limit for one train image).  L2 distances are the sequential float sum
 * d += (a-b)*(a-b) in element order (square root taken for SFM_NORM_L2), whatever the
 * inputs: the matrix cores only select candidates, and a query whose candidates cannot be
 * proven to contain the two nearest rows is recomputed against every train row. */
int sfm_knn2(sfm_ctx* ctx, int norm, const void* q, int nq, size_t q_stride,
             const void* t, int nt, size_t t_stride, int d, int32_t* idx, float* dist);
int sfm_knn2_dev(sfm_ctx* ctx, int norm, const void* d_q, int nq, size_t q_stride,
                 const void* d_t, int nt, size_t t_stride, int d, int32_t* d_idx,
                 float* d_dist);

/* Measurement aid (no counterpart in the reference): how many queries of the last L2 / L2SQR
 * call on this context (sfm_knn2*, sfm_ratio_matches, sfm_robust_matches, sfm_match_pairs_dev)
 * could not be proven exact from the matrix-core candidates and were recomputed against every
 * train row.  (With SFM_L2_COUNT_STRETCHES=1 in the environment it also counts the queries for
 * which one stretch of 256 train rows was recomputed row by row.)  Synchronises the stream. */
int sfm_l2_exhaustive_count(sfm_ctx* ctx, int* count);

/* DescriptorMatcher::ratioMaches, Source Files/feature_processing.cpp:118-130:
 * 2-NN followed by  k[0].distance < ratio * k[1].distance.  out has room for nq
 * matches in ascending queryIdx order; *out_count receives their number.  A
 * train set with fewer than two rows yields no match (the reference reads k[1]
 * out of bounds there). */
int sfm_ratio_matches(sfm_ctx* ctx, int norm, const void* q, int nq, size_t q_stride,
                      const void* t, int nt, size_t t_stride, int d, float ratio,
                      sfm_dmatch* out, int* out_count);

/* DescriptorMatcher::findRobustMatches up to and including the cross-check,
 * Source Files/feature_processing.cpp:148-176: ratio matches prev->curr and
 * curr->prev, then every backward match whose forward counterpart agrees, in
 * the order of the backward list (ascending curr index).  out has room for
 * min(n_prev, n_curr) matches (queryIdx = prev row, trainIdx = curr row). */
int sfm_robust_matches(sfm_ctx* ctx, int norm, const void* prev, int n_prev,
                       size_t prev_stride, const void* curr, int n_curr, size_t curr_stride,
                       int d, float ratio, sfm_dmatch* out, int* out_count);

/* The ratio test and cross-check of findRobustMatches (Source Files/feature_processing.cpp:
 * 126-129 and :159-176) on two 2-NN searches that exist on the device: d_fwd_* is
 * sfm_knn2_dev(prev -> curr), [n_prev][2]; d_bwd_* is sfm_knn2_dev(curr -> prev), [n_curr][2].
 * Writes up to `cap` matches in findRobustMatches' order and their number (not clipped to cap)
 * to *d_match_count.  With sfm_knn2_dev on row slabs this splits ONE large frame pair over
 * several GPUs: a 2-NN search is independent per query row, so every GPU searches a slab of prev
 * rows against all of curr and a slab of curr rows against all of prev, the [rows][2] results
 * are gathered, and this step runs on the gathered arrays (the reference re-matches the current
 * frame against a handful of earlier views, Source Files/tracking.cpp:127-135: fewer pairs than
 * GPUs). */
int sfm_ratio_cross_dev(sfm_ctx* ctx, const int32_t* d_fwd_idx, const float* d_fwd_dist, int n_prev,
                        const int32_t* d_bwd_idx, const float* d_bwd_dist, int n_curr, float ratio,
                        sfm_dmatch* d_matches, int cap, int32_t* d_match_count);

/* The same over many frame pairs in one launch sequence (the reference runs
 * it once per pair: Source Files/app_solver.cpp:404, Source Files/
 * tracking.cpp:135).  d_desc holds n_frames descriptor blocks, block f at
 * d_desc + f*frame_stride bytes, rows contiguous (`d` bytes or floats each);
 * d_counts[f] (device, may be NULL = rows_per_frame for all) is the row count of
 * frame f.  d_pairs is a device array of n_pairs (prev_frame, curr_frame)
 * int32 pairs.  Pair p writes up to `cap` matches at d_matches + p*cap and its
 * match count at d_match_counts[p] (the count is not clipped to cap). */
int sfm_match_pairs_dev(sfm_ctx* ctx, int norm, const void* d_desc, int n_frames,
                        size_t frame_stride, int rows_per_frame, const int32_t* d_counts,
                        int d, const int32_t* d_pairs, int n_pairs, float ratio,
                        sfm_dmatch* d_matches, int cap, int32_t* d_match_counts);
/* Workspace bytes sfm_match_pairs_dev needs for these sizes (sfm_ctx_reserve). */
size_t sfm_match_pairs_workspace(int norm, int n_frames, int rows_per_frame, int d, int n_pairs);

/* sfm_match_pairs_dev for windowed / all-pairs matching sharded over GPUs (the reference
 * re-matches the current frame against a window of earlier views, Source Files/
 * tracking.cpp:127-135, one findRobustMatches per pair): SFM_NORM_HAMMING only, rows_per_frame
 * <= 65536.  Instead of `cap` cv::DMatch slots per pair the cross-checked matches of all pairs
 * are written back to back, in pair order and within a pair in findRobustMatches' order, as
 * sfm_match16 records: pair p occupies d_out[d_offsets[p] .. d_offsets[p+1]); d_offsets has
 * n_pairs + 1 int64 entries (the last is the total), d_pair_counts n_pairs int32 entries.
 * Records at or beyond out_cap are not written (the offsets still tell the true total, so the
 * caller can detect the overflow and repeat with a larger buffer).  Pairs are processed 1024 at
 * a time through the context's workspace. */
int sfm_match_pairs_compact_dev(sfm_ctx* ctx, int norm, const void* d_desc, int n_frames,
                                size_t frame_stride, int rows_per_frame, const int32_t* d_counts,
                                int d, const int32_t* d_pairs, int n_pairs, float ratio,
                                sfm_match16* d_out, size_t out_cap, int32_t* d_pair_counts,
                                int64_t* d_offsets);
size_t sfm_match_pairs_compact_workspace(int norm, int n_frames, int rows_per_frame, int d, int n_pairs);

/* Reassembly of gathered match lists (no counterpart in the reference, which is one process):
 * segment s = d_len[s] elements of elem_bytes bytes (even) copied from d_src + d_src_off[s]
 * elements to d_dst + d_dst_off[s] elements.  Segments must not overlap in d_dst. */
int sfm_gather_segments_dev(sfm_ctx* ctx, const void* d_src, const int64_t* d_src_off,
                            const int64_t* d_dst_off, const int32_t* d_len, int n_segments,
                            int elem_bytes, void* d_dst);

/* ---- pyramidal Lucas-Kanade --------------------------------------------- */

/* OptFlow::computeFlow, Source Files/feature_processing.cpp:238-249:
 * cv::SparsePyrLKOpticalFlow::calc with flags 0 on two 8-bit single-channel
 * images, then status[i] = 0 where err[i] > max_error.  The tracker settings
 * are those OptFlow::OptFlow hands to SparsePyrLKOpticalFlow::create
 * (Source Files/feature_processing.cpp:228-236): window, max_level, the term
 * criteria (clamped as OpenCV does: count to [0,100], epsilon to [0,10]) and
 * OpenCV's default minEigThreshold 1e-4.  Points are interleaved (x, y) floats.
 * err (may be NULL) receives the per-point residual.  The normal-equation sums
 * are accumulated exactly in integers (OpenCV: float, build-dependent order).
 * Against a float-accumulating run of the same algorithm: a point whose two runs take the
 * same number of iterations on every pyramid level agrees within 1e-3 px (measured: 3.4e-4)
 * with the same status; the other points (a stop test decided differently under the two
 * roundings: 0 to 0.34 % of the points on the tested shapes, 0.32 % of 5000 points at
 * 1280x720) agree within the termination epsilon.
 * Windows up to 41x41 are supported.  The two images may have different row pitches, as
 * calc accepts them (one of them a region of a larger image): prev_stride and curr_stride are
 * the bytes between rows of each. */
int sfm_pyrlk(sfm_ctx* ctx, const uint8_t* prev, const uint8_t* curr, int width, int height,
              size_t prev_stride, size_t curr_stride, const float* prev_pts, int n, int win_w, int win_h,
              int max_level, int crit_type, int max_count, double epsilon,
              double min_eig_threshold, float max_error, float* curr_pts, uint8_t* status,
              float* err);

/* Batched device form: n_pairs image pairs, pair p at d_prev + p*image_stride /
 * d_curr + p*image_stride (rows `stride` bytes apart in both), n points per pair at
 * d_prev_pts + p*n*2.  Outputs are laid out likewise. */
int sfm_pyrlk_batch_dev(sfm_ctx* ctx, const uint8_t* d_prev, const uint8_t* d_curr,
                        int n_pairs, size_t image_stride, int width, int height,
                        size_t stride, const float* d_prev_pts, int n, int win_w, int win_h,
                        int max_level, int crit_type, int max_count, double epsilon,
                        double min_eig_threshold, float max_error, float* d_curr_pts,
                        uint8_t* d_status, float* d_err);
size_t sfm_pyrlk_workspace(int n_pairs, int width, int height, int win_w, int win_h,
                           int max_level);

/* The same in two steps, for sequences: the reference tracks frame k -> k+1 on every
 * iteration (Source Files/app_solver.cpp:68, :366), so frame k+1's pyramid serves as the next
 * image of one call and as the previous image of the following one.  OpenCV offers this split as
 * cv::buildOpticalFlowPyramid + calc on pyramids; the reference does not use it.
 * sfm_lk_build_pyramids_dev writes one pyramid (every level with its reflect-101 border) per
 * frame, sfm_lk_pyramid_bytes() bytes each, back to back into caller-owned device memory
 * (16-byte aligned); frame f at d_frames + f*image_stride, rows `stride` bytes apart.  A pyramid
 * depends on the image size, the window and max_level, and is valid for any later call with the
 * same values. */
size_t sfm_lk_pyramid_bytes(int width, int height, int win_w, int win_h, int max_level);
int sfm_lk_build_pyramids_dev(sfm_ctx* ctx, const uint8_t* d_frames, int n_frames, size_t image_stride,
                              int width, int height, size_t stride, int win_w, int win_h, int max_level,
                              uint8_t* d_pyramids);
/* sfm_pyrlk_batch_dev on pyramids that exist: pair p tracks its n points from pyramid
 * d_pairs[2p] to pyramid d_pairs[2p+1] of the n_pyramids in d_pyramids (d_pairs: device array of
 * int32; an index outside [0, n_pyramids) is clamped into it).  Results are bit-identical to
 * sfm_pyrlk_batch_dev on the frames the pyramids were built from. */
int sfm_pyrlk_pyramids_dev(sfm_ctx* ctx, const uint8_t* d_pyramids, int n_pyramids, const int32_t* d_pairs,
                           int n_pairs, int width, int height, const float* d_prev_pts, int n, int win_w,
                           int win_h, int max_level, int crit_type, int max_count, double epsilon,
                           double min_eig_threshold, float max_error, float* d_curr_pts, uint8_t* d_status,
                           float* d_err);
size_t sfm_pyrlk_pyramids_workspace(int n_pairs, int width, int height, int win_w, int win_h, int max_level);

/* The filters between computeFlow and findCameraPose (ProcesingAdds::filterPointsByStatusMask
 * and filterPointsByBoundary, Source Files/feature_processing.cpp:332-358, called at
 * Source Files/app_solver.cpp:69-70), on device-resident tracking results of n_pairs image
 * pairs (n slots each): correspondence i of a pair is kept when d_status is set and the current
 * point lies in the rectangle (x <= pt.x < x + w, y <= pt.y < y + h as cv::Rect::contains;
 * w or h <= 0: no boundary test); kept correspondences are written in order to the front of
 * the pair's n output slots and counted in d_counts.  d_valid (optional) gives the number of
 * meaningful input slots per pair (e.g. the corner count of sfm_good_features_dev).
 * The reference's loops never advance their index correction, so after the first removal they
 * test one element and erase another (and can index past the end); this entry point
 * implements the filter they evidently intend.  Not in place. */
int sfm_filter_tracked_dev(sfm_ctx* ctx, const float* d_prev_pts, const float* d_curr_pts, const uint8_t* d_status,
                           const int32_t* d_valid, int n_pairs, int n, int bound_x, int bound_y, int bound_w,
                           int bound_h, float* d_prev_out, float* d_curr_out, int32_t* d_counts);

/* ---- triangulation ------------------------------------------------------ */

/* Reconstruction::triangulateCloud with pointsToRGBCloud, Source Files/
 * reconstruction.cpp:41-73 and :6-39: cv::triangulatePoints(K*prev_pose,
 * K*curr_pose, ...) per correspondence (4x4 homogeneous system, right singular
 * vector of the smallest singular value, divided by w), then reprojection with
 * the separately supplied R, t, and
 *   mask = err < max_proj_err && depth > min_dist && depth < max_dist.
 * K and R are row-major 3x3, poses row-major 3x4, t has 3 entries, points are
 * interleaved (x, y) floats.  X is [n][3] doubles.  bgr (may be NULL) is the
 * 8-bit 3-channel image colours are read from at round(curr point), clamped to
 * the image (the reference does not check bounds); rgb is [n][3] and untouched
 * when bgr is NULL.  err (may be NULL) receives the reprojection error. */
int sfm_triangulate(sfm_ctx* ctx, const double* K, const double* prev_pose,
                    const double* curr_pose, const double* R, const double* t,
                    const float* prev_pts, const float* curr_pts, int n, const uint8_t* bgr,
                    int img_w, int img_h, size_t img_stride, float min_dist, float max_dist,
                    float max_proj_err, double* X, uint8_t* rgb, uint8_t* mask, double* err);
/* Device form: the five small matrices stay host pointers (they travel as
 * kernel arguments); points, image and outputs are device pointers. */
int sfm_triangulate_dev(sfm_ctx* ctx, const double* K, const double* prev_pose,
                        const double* curr_pose, const double* R, const double* t,
                        const float* d_prev_pts, const float* d_curr_pts, int n,
                        const uint8_t* d_bgr, int img_w, int img_h, size_t img_stride,
                        float min_dist, float max_dist, float max_proj_err, double* d_X,
                        uint8_t* d_rgb, uint8_t* d_mask, double* d_err);

/* sfm_triangulate_dev for callers that cannot allocate device memory (code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc); it has no counterpart in the
 * reference.  Same arguments, same kernels and bit-identical results, but the lists of deferred
 * points live in a fixed array in the library's device image (about 4 MiB, held by every GPU the
 * library runs on until the process ends) instead of the context's grow-only workspace, and
 * the call makes no hipMalloc / hipFree and creates no events (it is not covered by
 * sfm_ctx_set_profiling).  n may be at most sfm_triangulate_noalloc_max_points() (2^20); more
 * is rejected with SFM_E_INVALID.  All calls on one GPU share that GPU's array: calls that
 * overlap on one GPU (two contexts / streams) race; calls on one stream are safe. */
int sfm_triangulate_noalloc_dev(sfm_ctx* ctx, const double* K, const double* prev_pose,
                                const double* curr_pose, const double* R, const double* t,
                                const float* d_prev_pts, const float* d_curr_pts, int n,
                                const uint8_t* d_bgr, int img_w, int img_h, size_t img_stride,
                                float min_dist, float max_dist, float max_proj_err, double* d_X,
                                uint8_t* d_rgb, uint8_t* d_mask, double* d_err);
int sfm_triangulate_noalloc_max_points(void);

/* Device-resident hand-off from matching to triangulation: correspondence i is
 * (prev key point d_matches[i].queryIdx, curr key point d_matches[i].trainIdx), i.e. the
 * gather the reference does on the host after the cross-check (Source Files/
 * feature_processing.cpp:164-165, :196-197) followed by its quadratic vector-to-matrix
 * conversion (Header Files/common.h:47-57, called at Source Files/reconstruction.cpp:50-54),
 * fused into the triangulation of Source Files/reconstruction.cpp:41-73.  d_matches and
 * d_count are what sfm_match_pairs_dev wrote for one pair (d_count may be NULL: all `cap`
 * entries are used); outputs are sized for `cap` and filled for the first
 * min(*d_count, cap) entries.  A key point record starts with its (x, y) floats, records are
 * `stride` bytes apart: 28 for an array of cv::KeyPoint, 8 for plain points. */
int sfm_triangulate_matches_dev(sfm_ctx* ctx, const double* K, const double* prev_pose,
                                const double* curr_pose, const double* R, const double* t,
                                const sfm_dmatch* d_matches, const int32_t* d_count, int cap,
                                const void* d_prev_keypts, size_t prev_keypt_stride,
                                const void* d_curr_keypts, size_t curr_keypt_stride,
                                const uint8_t* d_bgr, int img_w, int img_h, size_t img_stride,
                                float min_dist, float max_dist, float max_proj_err, double* d_X,
                                uint8_t* d_rgb, uint8_t* d_mask, double* d_err);

/* ---- epipolar filter ------------------------------------------------------ */

/* cv::findFundamentalMat(prevAligPts, currAligPts, inliersMask), the epipolar filter of
 * DescriptorMatcher::findRobustMatches at Source Files/feature_processing.cpp:187-188:
 * FM_RANSAC on 7-point samples with OpenCV's generator, sampling rules, error measure
 * (larger squared distance to the two epipolar lines), adaptive iteration count and
 * regimes (fewer than 7 points: nothing found; exactly 7: one direct solve, every point an
 * inlier; 8 to 14: least median of squares; 15 and more: RANSAC).  The reference calls it
 * with thresh 3, confidence 0.99, max_iters 1000; out-of-range thresh / confidence are
 * replaced by those defaults as OpenCV does.  Points are interleaved (x, y) floats.
 * mask[i] = 1 for inliers of the chosen matrix; it is all 0 when nothing is found (OpenCV
 * leaves the caller's mask untouched then; the reference passes zeros).  F (may be NULL)
 * receives the chosen 3x3 matrix, row-major.  The null space of each 7-point system is
 * taken by elimination rather than by OpenCV's SVD; the candidate matrices agree with
 * OpenCV's up to rounding. */
int sfm_fundamental(sfm_ctx* ctx, const float* pts1, const float* pts2, int n, double thresh,
                    double confidence, int max_iters, double* F, uint8_t* mask, int* found);
/* The same for n_problems independent point sets on the device: set p starts at
 * d_pts + p * problem_stride points and holds d_counts[p] (or count_uniform when d_counts is
 * NULL) correspondences; d_F is [n_problems][9] (may be NULL), d_mask [n_problems]
 * [problem_stride], d_found [n_problems]. */
int sfm_fundamental_dev(sfm_ctx* ctx, const float* d_pts1, const float* d_pts2, size_t problem_stride,
                        const int32_t* d_counts, int count_uniform, int n_problems, double thresh,
                        double confidence, int max_iters, double* d_F, uint8_t* d_mask,
                        int32_t* d_found);
size_t sfm_fundamental_workspace(int n_problems, int max_iters);

/* The epipolar filter of findRobustMatches for many frame pairs, on the device, in place
 * (Source Files/feature_processing.cpp:187-204 and :223 with useEpipolarFilter == true):
 * gathers the matched key points of every pair (d_matches / d_match_counts as
 * sfm_match_pairs_dev wrote them, key point arrays of frame f at d_keypts + f *
 * frame_stride bytes, records keypt_stride bytes apart starting with (x, y) floats), runs
 * cv::findFundamentalMat per pair and keeps the inlier matches in order, updating the
 * counts.  A pair with fewer than 7 matches keeps none, as in the reference (OpenCV finds no
 * matrix and the zero-initialised mask drops everything).  d_mask (may be NULL) receives the
 * [n_pairs][cap] inlier masks of the unfiltered lists. */
int sfm_epipolar_filter_pairs_dev(sfm_ctx* ctx, const void* d_keypts, size_t frame_stride,
                                  size_t keypt_stride, const int32_t* d_pairs, int n_pairs,
                                  sfm_dmatch* d_matches, int cap, int32_t* d_match_counts, double thresh,
                                  double confidence, int max_iters, uint8_t* d_mask);
size_t sfm_epipolar_filter_pairs_workspace(int n_pairs, int cap, int max_iters);

/* ---- corner detection (producer of the tracker's input) --------------------------------- */

/* cv::goodFeaturesToTrack(imGray, _corners, maxCorners, qualityLevel, minDistance) of
 * FeatureDetector::generateFlowFeatures, Source Files/feature_processing.cpp:87-99
 * (Shi-Tomasi: no mask, blockSize 3, Sobel aperture 3, useHarrisDetector false).
 * img: w x h 8-bit gray, `stride` bytes between rows, host memory.  corners: room for `cap`
 * (x, y) float pairs; *count receives the number found, strongest first.  max_corners == 0
 * means no limit (OpenCV's convention); quality > 0, min_distance >= 0 as OpenCV asserts.
 * The image must be at least 3 x 3 and at most 2^26 pixels. */
int sfm_good_features(sfm_ctx* ctx, const uint8_t* img, int w, int h, size_t stride, int max_corners,
                      double quality, double min_distance, float* corners, int cap, int* count);

/* The same on a device-resident image, enqueued on the context's stream without a host
 * round trip.  d_info receives three int32: corners written, local maxima above the
 * threshold, and 1 if those exceeded the candidate capacity (then the corners are those of the
 * candidates that fitted and the call should be repeated with a larger max_candidates;
 * 0 selects w*h/4 + 1024, enough unless the image has plateaus of equal corner strength). */
int sfm_good_features_dev(sfm_ctx* ctx, const uint8_t* d_img, int w, int h, size_t stride, int max_corners,
                          double quality, double min_distance, int max_candidates, float* d_corners,
                          int cap, int32_t* d_info);
size_t sfm_good_features_workspace(int w, int h, int max_candidates);

/* The same for up to 64 equally sized device-resident images per call (image i at d_imgs +
 * i * image_stride bytes; corners of image i at d_corners + i * cap * 2 floats, its three
 * counters at d_info + 3 i).  One set of launches serves the whole batch: for single video
 * frames most of the time is the gaps between some thirty short kernels. */
int sfm_good_features_batch_dev(sfm_ctx* ctx, const uint8_t* d_imgs, int n_images, size_t image_stride, int w, int h,
                                size_t stride, int max_corners, double quality, double min_distance,
                                int max_candidates, float* d_corners, int cap, int32_t* d_info);
size_t sfm_good_features_batch_workspace(int n_images, int w, int h, int max_candidates);

/* cv::cornerMinEigenVal(image, eig, 3, 3), the first stage of the above: d_eig is [h][w]
 * float, dense. */
int sfm_min_eigen_val_dev(sfm_ctx* ctx, const uint8_t* d_img, int w, int h, size_t stride, float* d_eig);

/* ---- relative pose from tracked points ---------------------------------------------------- */

#define SFM_POSE_RANSAC 0 /* cv::RANSAC */
#define SFM_POSE_LMEDS 1  /* cv::LMEDS  */

/* cv::findEssentialMat(prevPts, currPts, cameraK, method, prob, threshold, mask) of
 * Tracking::findCameraPose, Source Files/tracking.cpp:94 (five-point solver in a RANSAC or
 * LMedS registrator; max_iters 1000 is OpenCV's default).  pts: n (x, y) float pairs in
 * pixels, host memory; K: 3x3 row-major camera matrix.  E: room for 10 matrices of 9 doubles;
 * *n_matrices receives how many were written: 1 when a model was found, 0 when none, and
 * every solution of the direct solve when n == 5 (OpenCV stacks them into a 3k x 3 matrix,
 * which the reference then rejects).  mask: n bytes, 1 for inliers. */
int sfm_find_essential(sfm_ctx* ctx, const float* pts1, const float* pts2, int n, const double* K, int method,
                       double prob, double threshold, int max_iters, double* E, uint8_t* mask, int* n_matrices);

/* Many device-resident point sets per call (problem p at p * problem_stride points; counts
 * from d_counts or count_uniform).  d_E: [n_problems][10][9] doubles; d_mask:
 * [n_problems][problem_stride]; d_found: matrices written per problem.  K is read on the host. */
int sfm_find_essential_dev(sfm_ctx* ctx, const float* d_pts1, const float* d_pts2, size_t problem_stride,
                           const int32_t* d_counts, int count_uniform, int n_problems, const double* K, int method,
                           double prob, double threshold, int max_iters, double* d_E, uint8_t* d_mask,
                           int32_t* d_found);
size_t sfm_find_essential_workspace(int n_problems, size_t problem_stride, int max_iters);

/* cv::recoverPose(E, prevPts, currPts, cameraK, R, t, mask) of Tracking::findCameraPose,
 * Source Files/tracking.cpp:110 (distanceThresh 50, OpenCV's default).  E: 9 doubles; pts:
 * n (x, y) float pairs in pixels; mask_io: n bytes, on input the inlier mask of
 * sfm_find_essential (all ones for "no mask"), on output cleared where the chosen pose fails
 * the cheirality test; *good receives the number of points that pass (the function's return
 * value in OpenCV).  R: 9 doubles row-major, t: 3 doubles. */
int sfm_recover_pose(sfm_ctx* ctx, const double* E, const float* pts1, const float* pts2, int n, const double* K,
                     double* R, double* t, uint8_t* mask_io, int* good);

/* Many device-resident point sets per call.  d_E: problem p's matrix at d_E + p * E_stride
 * doubles (E_stride 90 chains after sfm_find_essential_dev); d_found (optional): a problem is
 * only processed when its entry is exactly 1; max_first_residual > 0 adds the reference's
 * tests: more than five correspondences (Source Files/tracking.cpp:92) and
 * | p1^T K^-T E K^-1 p0 | <= limit on the first correspondence (Source Files/
 * tracking.cpp:99-107, 1e-3 there).  Problems that are skipped get d_good = 0. */
int sfm_recover_pose_dev(sfm_ctx* ctx, const double* d_E, size_t E_stride, const int32_t* d_found,
                         const float* d_pts1, const float* d_pts2, size_t problem_stride, const int32_t* d_counts,
                         int count_uniform, int n_problems, const double* K, double distance_thresh,
                         double max_first_residual, double* d_R, double* d_t, uint8_t* d_mask_io, int32_t* d_good);
size_t sfm_recover_pose_workspace(int n_problems, size_t problem_stride);

/* Tracking::findCameraPose, Source Files/tracking.cpp:91-113, for many device-resident point
 * sets without a host round trip: sfm_find_essential_dev, the 3x3 and residual checks, then
 * sfm_recover_pose_dev.  d_inliers[p] is numInliers (0 where the reference returns false
 * before recoverPose); the caller compares it with minInliers.  d_E: [n_problems][10][9]. */
int sfm_find_camera_pose_dev(sfm_ctx* ctx, const float* d_pts1, const float* d_pts2, size_t problem_stride,
                             const int32_t* d_counts, int count_uniform, int n_problems, const double* K, int method,
                             double prob, double threshold, double* d_E, double* d_R, double* d_t, uint8_t* d_mask,
                             int32_t* d_found, int32_t* d_inliers);
size_t sfm_find_camera_pose_workspace(int n_problems, size_t problem_stride);

#ifdef __cplusplus
}
#endif
#endif /* SFM_HIP_H */
