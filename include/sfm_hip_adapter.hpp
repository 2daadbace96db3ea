/*
 * sfm_hip_adapter.hpp -- header-only C++ host side over the C ABI in sfm_hip.h.
 *
 * It gives the three wrapper classes of the reference the same public
 * interface they have there, with bodies that call the MI355X library:
 *
 *   sfm_hip::DescriptorMatcher  Header Files/feature_processing.h:33-57
 *   sfm_hip::OptFlow            Header Files/feature_processing.h:59-100
 *   sfm_hip::Reconstruction     Header Files/reconstruction.h:140-159
 *                               (constructor and triangulateCloud only)
 *
 * With OpenCV's headers on the include path the classes take and return the
 * OpenCV types the reference passes (cv::Mat, cv::KeyPoint, cv::DMatch, ...),
 * so Source Files/app_solver.cpp compiles against them unchanged.  Without
 * OpenCV (define SFM_ADAPTER_NO_OPENCV, or simply have no <opencv2/core.hpp>)
 * a small set of stand-in types with the same member names and memory layout
 * is used instead, which is how this repository tests the adapter.
 *
 * Objects are cheap to copy: the reference passes them by value
 * (Header Files/app_solver.h:120, Header Files/tracking.h:88), so the GPU
 * context is held through a shared handle.
 */
#ifndef SFM_HIP_ADAPTER_HPP
#define SFM_HIP_ADAPTER_HPP

#include <algorithm>
#include <cctype>
#include <cstdint>
#include <cstring>
#include <cmath>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "sfm_hip.h"

#if !defined(SFM_ADAPTER_NO_OPENCV) && defined(__has_include)
#if __has_include(<opencv2/core.hpp>)
#define SFM_ADAPTER_HAVE_OPENCV 1
#endif
#endif

#ifdef SFM_ADAPTER_HAVE_OPENCV
#include <opencv2/calib3d.hpp>
#include <opencv2/core.hpp>
#include <opencv2/features2d.hpp>
#include <opencv2/video/tracking.hpp>
namespace sfm_hip {
namespace cvx = cv;
inline bool mat_is_u8(const cv::Mat& m) { return m.depth() == CV_8U; }
inline bool mat_is_f32(const cv::Mat& m) { return m.depth() == CV_32F; }
inline int mat_channels(const cv::Mat& m) { return m.channels(); }
inline size_t mat_step(const cv::Mat& m) { return m.step[0]; }
inline cv::Mat mat_u8_column(int rows) { return cv::Mat(rows, 1, CV_8U); }
[[noreturn]] inline void raise(const std::string& what) {
    CV_Error(cv::Error::StsError, what);
}
}  // namespace sfm_hip
#else
namespace sfm_hip {
/* Stand-ins for the OpenCV types at the boundary: same member names, and for
 * the plain structs the same layout, as OpenCV 4's. */
namespace cvx {
struct Point2f {
    float x, y;
    Point2f() : x(0), y(0) {}
    Point2f(float x_, float y_) : x(x_), y(y_) {}
};
struct Size {
    int width, height;
    Size() : width(0), height(0) {}
    Size(int w, int h) : width(w), height(h) {}
};
struct KeyPoint { /* 28 bytes, as cv::KeyPoint */
    Point2f pt;
    float size, angle, response;
    int octave, class_id;
    KeyPoint() : size(0), angle(-1), response(0), octave(0), class_id(-1) {}
    KeyPoint(float x, float y, float s) : pt(x, y), size(s), angle(-1), response(0), octave(0), class_id(-1) {}
};
struct DMatch { /* 16 bytes, as cv::DMatch */
    int queryIdx, trainIdx, imgIdx;
    float distance;
    DMatch() : queryIdx(-1), trainIdx(-1), imgIdx(-1), distance(3.402823466e+38f) {}
};
template <typename T, int N>
struct Vec {
    T val[N];
    Vec() { for (int i = 0; i < N; ++i) val[i] = T(0); }
    T& operator[](int i) { return val[i]; }
    const T& operator[](int i) const { return val[i]; }
};
typedef Vec<double, 3> Vec3d;
typedef Vec<unsigned char, 3> Vec3b;
template <int R, int C>
struct Matx {
    double val[R * C];
    Matx() { for (int i = 0; i < R * C; ++i) val[i] = 0.0; }
    double& operator()(int r, int c) { return val[r * C + c]; }
    const double& operator()(int r, int c) const { return val[r * C + c]; }
};
typedef Matx<3, 3> Matx33d;
typedef Matx<3, 4> Matx34d;
typedef Matx<3, 1> Matx31d;
struct TermCriteria {
    enum Type { COUNT = 1, MAX_ITER = COUNT, EPS = 2 };
    int type, maxCount;
    double epsilon;
    TermCriteria() : type(0), maxCount(0), epsilon(0) {}
    TermCriteria(int t, int c, double e) : type(t), maxCount(c), epsilon(e) {}
};
/* Dense row-major 2-D array of u8 or f32 with 1 or 3 channels. */
struct Mat {
    enum Depth { U8 = 0, F32 = 5 };
    int rows, cols, depth_, channels_;
    size_t step;
    unsigned char* data;
    std::shared_ptr<std::vector<unsigned char>> storage;
    Mat() : rows(0), cols(0), depth_(U8), channels_(1), step(0), data(nullptr) {}
    Mat(int r, int c, Depth d, int ch = 1)
        : rows(r), cols(c), depth_(d), channels_(ch),
          step(size_t(c) * ch * (d == F32 ? 4 : 1)),
          storage(std::make_shared<std::vector<unsigned char>>(size_t(r) * step)) {
        data = storage->data();
    }
    /* Wraps memory the caller keeps alive. */
    Mat(int r, int c, Depth d, int ch, void* p, size_t step_bytes)
        : rows(r), cols(c), depth_(d), channels_(ch), step(step_bytes), data(static_cast<unsigned char*>(p)) {}
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
};
template <typename T>
using Ptr = std::shared_ptr<T>;
/* The three calls FeatureDetector::generateFeatures makes on cv::Feature2D. */
struct Feature2D {
    virtual ~Feature2D() {}
    virtual void detect(const Mat& image, std::vector<KeyPoint>& keypoints) = 0;
    virtual void compute(const Mat& image, std::vector<KeyPoint>& keypoints, Mat& descriptors) = 0;
    virtual void detectAndCompute(const Mat& image, const Mat& mask, std::vector<KeyPoint>& keypoints,
                                  Mat& descriptors) = 0;
};
inline Mat noArray() { return Mat(); }
}  // namespace cvx
inline bool mat_is_u8(const cvx::Mat& m) { return m.depth_ == cvx::Mat::U8; }
inline bool mat_is_f32(const cvx::Mat& m) { return m.depth_ == cvx::Mat::F32; }
inline int mat_channels(const cvx::Mat& m) { return m.channels_; }
inline size_t mat_step(const cvx::Mat& m) { return m.step; }
inline cvx::Mat mat_u8_column(int rows) { return cvx::Mat(rows, 1, cvx::Mat::U8); }
[[noreturn]] inline void raise(const std::string& what) { throw std::runtime_error(what); }
}  // namespace sfm_hip
#endif

namespace sfm_hip {

static_assert(sizeof(cvx::DMatch) == sizeof(sfm_dmatch), "DMatch must be 16 bytes");
static_assert(sizeof(cvx::Point2f) == 8, "Point2f must be two floats");

/* Shared GPU context; one per device is enough for the reference's single
 * main-loop thread (Source Files/app_solver.cpp:151). */
class Handle {
public:
    explicit Handle(int device = 0) {
        sfm_ctx* raw = nullptr;
        if (sfm_ctx_create(device, &raw) != SFM_OK) raise(std::string("sfm_ctx_create: ") + sfm_last_error(nullptr));
        ctx_.reset(raw, sfm_ctx_destroy);
    }
    sfm_ctx* get() const { return ctx_.get(); }
    void check(int status, const char* call) const {
        if (status != SFM_OK) raise(std::string(call) + ": " + sfm_last_error(ctx_.get()));
    }
    static Handle& shared() {
        static Handle h(0);
        return h;
    }

private:
    std::shared_ptr<sfm_ctx> ctx_;
};

/* Header Files/feature_processing.h:33-57.  `matcher` (a cv::Ptr to OpenCV's
 * matcher there) becomes the norm the constructor selects. */
class DescriptorMatcher {
public:
    const bool m_isVisDebug;
    const float m_ratioThreshold;
    const cvx::Size m_visDebugWinSize;
    int normType; /* SFM_NORM_* */
    Handle handle;

    /* Epipolar filter (points -> inlier mask), the step at
     * Source Files/feature_processing.cpp:187-188.  Defaults to the library's
     * restatement of cv::findFundamentalMat (RANSAC, 3 px, 0.99) on the GPU;
     * assign another callable (cv::findFundamentalMat itself, say) to replace it. */
    std::function<void(const std::vector<cvx::Point2f>&, const std::vector<cvx::Point2f>&, std::vector<uint8_t>&)>
        epipolarFilter;

    /* Source Files/feature_processing.cpp:101-116. */
    DescriptorMatcher(std::string method, const float ratioThreshold, const bool isVisDebug = false,
                      const cvx::Size visDebugWinSize = cvx::Size(), Handle h = Handle::shared())
        : m_isVisDebug(isVisDebug), m_ratioThreshold(ratioThreshold),
          m_visDebugWinSize(cvx::Size(visDebugWinSize.width * 2, visDebugWinSize.height * 3)), handle(h) {
        std::for_each(method.begin(), method.end(), [](char& c) { c = char(::toupper(c)); });
        if (method == "BRUTEFORCE_HAMMING") normType = SFM_NORM_HAMMING;
        else if (method == "BRUTEFORCE_SL2") normType = SFM_NORM_L2SQR;
        else if (method == "FLANNBASED") raise("FLANNBASED matching is approximate and not provided by sfm_hip");
        else normType = SFM_NORM_L2;
        static_assert(sizeof(cvx::Point2f) == 2 * sizeof(float), "Point2f must be two packed floats");
        epipolarFilter = [h](const std::vector<cvx::Point2f>& a, const std::vector<cvx::Point2f>& b,
                             std::vector<uint8_t>& m) {
            if (a.size() != b.size()) raise("epipolar filter: point lists differ in length");
            m.assign(a.size(), 0);
            double F[9];
            int found = 0;
            h.check(sfm_fundamental(h.get(), reinterpret_cast<const float*>(a.data()),
                                     reinterpret_cast<const float*>(b.data()), int(a.size()), 3.0, 0.99, 1000, F,
                                     m.data(), &found),
                     "sfm_fundamental");
        };
    }

    /* Source Files/feature_processing.cpp:118-130. */
    void ratioMaches(const cvx::Mat lDesc, const cvx::Mat rDesc, std::vector<cvx::DMatch>& matches) {
        matches.clear();
        if (lDesc.empty()) return;
        const int d = checkedDim(lDesc, rDesc);
        std::vector<cvx::DMatch> out(size_t(lDesc.rows));
        int count = 0;
        handle.check(sfm_ratio_matches(handle.get(), normType, lDesc.data, lDesc.rows, mat_step(lDesc), rDesc.data,
                                       rDesc.rows, mat_step(rDesc), d, m_ratioThreshold,
                                       reinterpret_cast<sfm_dmatch*>(out.data()), &count),
                     "sfm_ratio_matches");
        out.resize(size_t(count));
        matches.swap(out);
    }

    /* Source Files/feature_processing.cpp:144-226.  The two ratio matches and
     * the cross-check run on the GPU as one call; the lists are appended to,
     * as in the reference. */
    void findRobustMatches(std::vector<cvx::KeyPoint> prevKeyPts, std::vector<cvx::KeyPoint> currKeyPts,
                           cvx::Mat prevDesc, cvx::Mat currDesc, std::vector<cvx::Point2f>& prevAligPts,
                           std::vector<cvx::Point2f>& currAligPts, std::vector<cvx::DMatch>& matches,
                           std::vector<int>& prevPtsToKeyIdx, std::vector<int>& currPtsToKeyIdx,
                           cvx::Mat /*debugPrevFrame*/, cvx::Mat /*debugCurrFrame*/,
                           bool useEpipolarFilter) {
        if (!prevDesc.empty() && !currDesc.empty()) {
            const int d = checkedDim(prevDesc, currDesc);
            std::vector<cvx::DMatch> cross(size_t(std::min(prevDesc.rows, currDesc.rows)));
            int count = 0;
            handle.check(sfm_robust_matches(handle.get(), normType, prevDesc.data, prevDesc.rows, mat_step(prevDesc),
                                            currDesc.data, currDesc.rows, mat_step(currDesc), d, m_ratioThreshold,
                                            reinterpret_cast<sfm_dmatch*>(cross.data()), &count),
                         "sfm_robust_matches");
            for (int i = 0; i < count; ++i) {
                const cvx::DMatch& m = cross[size_t(i)];
                prevAligPts.push_back(prevKeyPts.at(size_t(m.queryIdx)).pt);
                currAligPts.push_back(currKeyPts.at(size_t(m.trainIdx)).pt);
                matches.push_back(m);
            }
        }
        if (prevAligPts.empty() || currAligPts.empty()) return;

        std::vector<uint8_t> inliers(matches.size(), 1);
        if (useEpipolarFilter) {
            if (!epipolarFilter) raise("useEpipolarFilter needs DescriptorMatcher::epipolarFilter to be set");
            epipolarFilter(prevAligPts, currAligPts, inliers);
        }
        std::vector<cvx::DMatch> keptMatches;
        std::vector<cvx::Point2f> keptPrev, keptCurr;
        for (size_t m = 0; m < matches.size(); ++m) {
            if (inliers[m] || !useEpipolarFilter) {
                keptPrev.push_back(prevKeyPts.at(size_t(matches[m].queryIdx)).pt);
                keptCurr.push_back(currKeyPts.at(size_t(matches[m].trainIdx)).pt);
                prevPtsToKeyIdx.push_back(matches[m].queryIdx);
                currPtsToKeyIdx.push_back(matches[m].trainIdx);
                keptMatches.push_back(matches[m]);
            }
        }
        std::swap(matches, keptMatches);
        std::swap(prevAligPts, keptPrev);
        std::swap(currAligPts, keptCurr);
    }

private:
    /* OpenCV asserts that both descriptor sets have one type and width and
     * that the type suits the norm; the same conditions raise here. */
    int checkedDim(const cvx::Mat& l, const cvx::Mat& r) const {
        if (mat_channels(l) != 1 || mat_channels(r) != 1) raise("descriptors must be single-channel");
        if (l.cols != r.cols) raise("descriptor widths differ");
        if (normType == SFM_NORM_HAMMING) {
            if (!mat_is_u8(l) || !mat_is_u8(r)) raise("Hamming matching needs CV_8U descriptors");
        } else if (!mat_is_f32(l) || !mat_is_f32(r)) {
            raise("L2 matching needs CV_32F descriptors");
        }
        return l.cols;
    }
};

/* Header Files/feature_processing.h:59-73. */
class OptFlowAddSettings {
public:
    float maxError, qualLvl, minDist;
    unsigned maxCorn, minFeatures;
    void setMaxError(float v) { maxError = v; }
    void setMaxCorners(unsigned v) { maxCorn = v; }
    void setQualityLvl(float v) { qualLvl = v; }
    void setMinDistance(float v) { minDist = v; }
    void setMinFeatures(unsigned v) { minFeatures = v; }
};

/* What cv::SparsePyrLKOpticalFlow::create keeps
 * (Source Files/feature_processing.cpp:229), with OpenCV's accessor names. */
struct PyrLKSettings {
    cvx::Size winSize;
    int maxLevel;
    cvx::TermCriteria criteria;
    int flags;
    double minEigThreshold;
    cvx::Size getWinSize() const { return winSize; }
    void setWinSize(cvx::Size s) { winSize = s; }
    int getMaxLevel() const { return maxLevel; }
    void setMaxLevel(int l) { maxLevel = l; }
    cvx::TermCriteria getTermCriteria() const { return criteria; }
    void setTermCriteria(const cvx::TermCriteria& c) { criteria = c; }
    int getFlags() const { return flags; }
    double getMinEigThreshold() const { return minEigThreshold; }
    void setMinEigThreshold(double v) { minEigThreshold = v; }
    PyrLKSettings* operator->() { return this; }
    const PyrLKSettings* operator->() const { return this; }
};

/* Header Files/feature_processing.h:75-100 (constructor and computeFlow). */
class OptFlow {
public:
    PyrLKSettings optFlow;
    OptFlowAddSettings additionalSettings;
    std::vector<unsigned char> statusMask; /* std::vector<uchar> in the reference */
    Handle handle;

    /* Source Files/feature_processing.cpp:228-236. */
    OptFlow(cvx::TermCriteria termcrit, int winSize, int maxLevel, float maxError, unsigned maxCorners,
            float qualityLevel, float minCornersDistance, unsigned minFeatures, Handle h = Handle::shared())
        : handle(h) {
        optFlow.winSize = cvx::Size(winSize, winSize);
        optFlow.maxLevel = maxLevel;
        optFlow.criteria = termcrit;
        optFlow.flags = 0;
        optFlow.minEigThreshold = 1e-4;
        additionalSettings.setMaxError(maxError);
        additionalSettings.setMaxCorners(maxCorners);
        additionalSettings.setQualityLvl(qualityLevel);
        additionalSettings.setMinDistance(minCornersDistance);
        additionalSettings.setMinFeatures(minFeatures);
    }

    /* Source Files/feature_processing.cpp:238-249. */
    void computeFlow(cvx::Mat imPrevGray, cvx::Mat imCurrGray, std::vector<cvx::Point2f>& prevPts,
                     std::vector<cvx::Point2f>& currPts, std::vector<unsigned char>& statusMask_) {
        const size_t n = prevPts.size();
        currPts.resize(n);
        statusMask_.resize(n);
        if (n == 0) return;
        if (!mat_is_u8(imPrevGray) || mat_channels(imPrevGray) != 1 || !mat_is_u8(imCurrGray) ||
            mat_channels(imCurrGray) != 1)
            raise("computeFlow needs 8-bit single-channel images");
        if (imPrevGray.rows != imCurrGray.rows || imPrevGray.cols != imCurrGray.cols)
            raise("computeFlow needs two images of one size");
        /* the row pitches may differ (one image a region of a larger one), as calc accepts */
        handle.check(sfm_pyrlk(handle.get(), imPrevGray.data, imCurrGray.data, imPrevGray.cols, imPrevGray.rows,
                               mat_step(imPrevGray), mat_step(imCurrGray),
                               reinterpret_cast<const float*>(prevPts.data()), int(n),
                               optFlow.winSize.width, optFlow.winSize.height, optFlow.maxLevel,
                               int(optFlow.criteria.type), optFlow.criteria.maxCount, optFlow.criteria.epsilon,
                               optFlow.minEigThreshold, additionalSettings.maxError,
                               reinterpret_cast<float*>(currPts.data()), statusMask_.data(), nullptr),
                     "sfm_pyrlk");
    }
};

/* Header Files/feature_processing.h:8-31.  generateFlowFeatures
 * (Source Files/feature_processing.cpp:87-99) runs on the GPU.  The key point detectors and
 * extractors of generateFeatures are OpenCV's own CPU code and stay with it: the public
 * `detector` / `extractor` handles are kept, the caller supplies them (the reference's
 * constructor picks them by name from OpenCV + opencv_contrib, :26-74), and
 * generateFeatures forwards to them exactly as Source Files/feature_processing.cpp:77-85
 * does. */
class FeatureDetector {
public:
    cvx::Ptr<cvx::Feature2D> detector, extractor;
    Handle handle;

    explicit FeatureDetector(std::string /*method*/ = "AKAZE", Handle h = Handle::shared()) : handle(h) {}
    FeatureDetector(std::string /*method*/, cvx::Ptr<cvx::Feature2D> detector_, cvx::Ptr<cvx::Feature2D> extractor_,
                    Handle h = Handle::shared())
        : detector(detector_), extractor(extractor_), handle(h) {}

    /* Source Files/feature_processing.cpp:77-85. */
    void generateFeatures(cvx::Mat& imGray, std::vector<cvx::KeyPoint>& keyPts, cvx::Mat& descriptor) {
        if (!detector || !extractor)
            raise("FeatureDetector::generateFeatures needs detector and extractor to be set (OpenCV's own objects)");
        if (detector != extractor) {
            detector->detect(imGray, keyPts);
            extractor->compute(imGray, keyPts, descriptor);
        } else {
            detector->detectAndCompute(imGray, cvx::noArray(), keyPts, descriptor);
        }
    }

    /* cv::goodFeaturesToTrack(imGray, _corners, maxCorners, qualityLevel, minDistance);
     * new corners are appended, points already in `corners` are kept. */
    void generateFlowFeatures(cvx::Mat& imGray, std::vector<cvx::Point2f>& corners, int maxCorners,
                              double qualityLevel, double minDistance) {
        if (!mat_is_u8(imGray) || mat_channels(imGray) != 1) raise("generateFlowFeatures needs an 8-bit single-channel image");
        if (maxCorners < 0) raise("maxCorners must not be negative");
        const size_t cap = maxCorners > 0 ? size_t(maxCorners) : size_t(imGray.rows) * size_t(imGray.cols);
        std::vector<cvx::Point2f> found(cap);
        int count = 0;
        handle.check(sfm_good_features(handle.get(), imGray.data, imGray.cols, imGray.rows, mat_step(imGray), maxCorners,
                                       qualityLevel, minDistance, reinterpret_cast<float*>(found.data()), int(cap),
                                       &count),
                     "sfm_good_features");
        corners.insert(corners.end(), found.begin(), found.begin() + count);
    }
};

/* Header Files/tracking.h:17-32 (settings and result of the relative-pose step;
 * drawRecoveredPose is visualisation and stays with the reference). */
class RecoveryPose {
public:
    int recPoseMethod; /* SFM_POSE_RANSAC / SFM_POSE_LMEDS (cv::RANSAC / cv::LMEDS in the reference) */
    std::string poseEstMethod;
    const double prob, threshold;
    const unsigned minInliers, numIter;
    const bool useExtrinsicGuess;
    cvx::Matx33d R;
    cvx::Matx31d t;
    cvx::Mat mask;

    /* Source Files/tracking.cpp:3-18: "RANSAC" selects RANSAC, anything else LMedS. */
    RecoveryPose(std::string method, const double prob_, const double threshold_, const unsigned minInliers_,
                 std::string poseEstMethod_, const bool useExtrinsicGuess_, const unsigned numIter_)
        : poseEstMethod(poseEstMethod_), prob(prob_), threshold(threshold_), minInliers(minInliers_),
          numIter(numIter_), useExtrinsicGuess(useExtrinsicGuess_) {
        std::for_each(method.begin(), method.end(), [](char& c) { c = char(::toupper(c)); });
        recPoseMethod = method == "RANSAC" ? SFM_POSE_RANSAC : SFM_POSE_LMEDS;
        R(0, 0) = R(1, 1) = R(2, 2) = 1.0;
        t(0, 0) = 1.0; /* cv::Matx31d::eye() */
    }
};

/* Header Files/tracking.h:82: the relative-pose member of Tracking. */
class Tracking {
public:
    /* Source Files/tracking.cpp:91-113: findEssentialMat, the 3x3 and residual checks,
     * recoverPose; R, t and mask are left in recPose. */
    static bool findCameraPose(RecoveryPose& recPose, std::vector<cvx::Point2f> prevPts,
                               std::vector<cvx::Point2f> currPts, const cvx::Matx33d& cameraK, int minInliers,
                               int& numInliers, Handle handle = Handle::shared()) {
        if (prevPts.size() <= 5 || currPts.size() <= 5) return false;
        if (prevPts.size() != currPts.size()) raise("findCameraPose: point lists differ in length");
        const int n = int(prevPts.size());
        double E[90];
        int n_matrices = 0;
        recPose.mask = mat_u8_column(n);
        handle.check(sfm_find_essential(handle.get(), reinterpret_cast<const float*>(prevPts.data()),
                                        reinterpret_cast<const float*>(currPts.data()), n, cameraK.val,
                                        recPose.recPoseMethod, recPose.prob, recPose.threshold, 1000, E,
                                        recPose.mask.data, &n_matrices),
                     "sfm_find_essential");
        if (n_matrices != 1) return false; /* the reference's E.cols == 3 && E.rows == 3 */
        /* | p1^T K^-T E K^-1 p0 | of the first correspondence, K upper triangular without skew */
        const double fx = cameraK(0, 0), fy = cameraK(1, 1), cx = cameraK(0, 2), cy = cameraK(1, 2);
        const double a[3] = {(double(prevPts[0].x) - cx) / fx, (double(prevPts[0].y) - cy) / fy, 1.0};
        const double b[3] = {(double(currPts[0].x) - cx) / fx, (double(currPts[0].y) - cy) / fy, 1.0};
        double err = 0;
        for (int r = 0; r < 3; ++r) err += b[r] * (E[3 * r] * a[0] + E[3 * r + 1] * a[1] + E[3 * r + 2] * a[2]);
        if (std::fabs(err) > 1e-03) return false;
        handle.check(sfm_recover_pose(handle.get(), E, reinterpret_cast<const float*>(prevPts.data()),
                                      reinterpret_cast<const float*>(currPts.data()), n, cameraK.val, recPose.R.val,
                                      recPose.t.val, recPose.mask.data, &numInliers),
                     "sfm_recover_pose");
        return numInliers > minInliers;
    }
};

/* Header Files/reconstruction.h:140-159 (constructor and triangulateCloud).
 * Camera is the reference's CameraParameters (Header Files/camera.h:11-42) or
 * anything else with a K33d member. */
class Reconstruction {
public:
    Reconstruction(const std::string triangulateMethod, const std::string baMethod, const double baMaxRMSE,
                   const float minDistance, const float maxDistance, const float maxProjectionError,
                   const bool useNormalizePts, Handle h = Handle::shared())
        : m_triangulateMethod(triangulateMethod), m_baMethod(baMethod), m_baMaxRMSE(baMaxRMSE),
          m_minDistance(minDistance), m_maxDistance(maxDistance), m_maxProjectionError(maxProjectionError),
          m_useNormalizePts(useNormalizePts), handle(h) {}

    /* Source Files/reconstruction.cpp:41-73 with :6-39.  points3D and
     * pointsRGB are cleared and refilled, mask is appended to.  The undistort
     * step there has no distortion coefficients, so both settings of
     * useNormalizePts give the same points.  With triangulateMethod "DLT" the
     * reference triangulates nothing (its call is commented out) and hands an
     * empty matrix to cv::projectPoints; that configuration raises here. */
    template <class Camera>
    void triangulateCloud(Camera camera, const std::vector<cvx::Point2f> prevPts,
                          const std::vector<cvx::Point2f> currPts, const cvx::Mat colorImage,
                          std::vector<cvx::Vec3d>& points3D, std::vector<cvx::Vec3b>& pointsRGB,
                          std::vector<bool>& mask, const cvx::Matx34d prevPose, const cvx::Matx34d currPose,
                          cvx::Matx33d& R, cvx::Matx31d& t) {
        if (prevPts.empty() || currPts.empty()) return;
        if (m_triangulateMethod == "DLT")
            raise("triangulateMethod \"DLT\" is a dead branch in the reference; use any other name");
        if (prevPts.size() != currPts.size()) raise("triangulateCloud needs as many prev as curr points");
        if (colorImage.empty() || !mat_is_u8(colorImage) || mat_channels(colorImage) != 3)
            raise("triangulateCloud needs an 8-bit 3-channel colour image");
        const int n = int(prevPts.size());
        double K[9], P1[12], P2[12], Rm[9], tv[3];
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) {
                K[r * 3 + c] = camera.K33d(r, c);
                Rm[r * 3 + c] = R(r, c);
            }
            for (int c = 0; c < 4; ++c) {
                P1[r * 4 + c] = prevPose(r, c);
                P2[r * 4 + c] = currPose(r, c);
            }
            tv[r] = t(r, 0);
        }
        points3D.resize(size_t(n));
        pointsRGB.resize(size_t(n));
        std::vector<uint8_t> m(static_cast<size_t>(n));
        static_assert(sizeof(cvx::Vec3d) == 24 && sizeof(cvx::Vec3b) == 3, "Vec3d / Vec3b layout");
        handle.check(sfm_triangulate(handle.get(), K, P1, P2, Rm, tv, reinterpret_cast<const float*>(prevPts.data()),
                                     reinterpret_cast<const float*>(currPts.data()), n, colorImage.data,
                                     colorImage.cols, colorImage.rows, mat_step(colorImage), m_minDistance,
                                     m_maxDistance, m_maxProjectionError, reinterpret_cast<double*>(points3D.data()),
                                     reinterpret_cast<uint8_t*>(pointsRGB.data()), m.data(), nullptr),
                     "sfm_triangulate");
        for (int i = 0; i < n; ++i) mask.push_back(m[size_t(i)] != 0);
    }

private:
    const std::string m_triangulateMethod, m_baMethod;
    const double m_baMaxRMSE;
    const float m_minDistance, m_maxDistance, m_maxProjectionError;
    const bool m_useNormalizePts;
    Handle handle;
};

#ifdef SFM_ADAPTER_HAVE_OPENCV
/* ---- Seam 2: OpenCV's own operator interfaces ---------------------------------------------
 * DescriptorMatcher::matcher and OptFlow::optFlow are public cv::Ptr members of the reference's
 * wrappers (Header Files/feature_processing.h:44, :79).  A maintainer who keeps the wrappers
 * as they are can assign these instead of cv::DescriptorMatcher::create(...) /
 * cv::SparsePyrLKOpticalFlow::create(...):
 *     descMatcher.matcher = cv::makePtr<sfm_hip::HipBFMatcher>(cv::NORM_HAMMING);
 *     optFlow.optFlow     = cv::makePtr<sfm_hip::HipSparsePyrLK>(cv::Size(21, 21), 3, crit);
 * The wrappers' own host loops (ratio test, cross-check, error gate) then run unchanged. */

/* cv::BFMatcher(normType, crossCheck = false) for k = 2 without masks: what
 * matcher->knnMatch(lDesc, rDesc, knnMatches, 2) needs (Source Files/feature_processing.cpp:121). */
class HipBFMatcher : public cv::DescriptorMatcher {
public:
    explicit HipBFMatcher(int cvNormType, Handle h = Handle::shared()) : cvNorm_(cvNormType), handle_(h) {
        if (cvNormType == cv::NORM_HAMMING) norm_ = SFM_NORM_HAMMING;
        else if (cvNormType == cv::NORM_L2) norm_ = SFM_NORM_L2;
        else if (cvNormType == cv::NORM_L2SQR) norm_ = SFM_NORM_L2SQR;
        else CV_Error(cv::Error::StsBadArg, "HipBFMatcher: NORM_HAMMING, NORM_L2 or NORM_L2SQR");
    }
    bool isMaskSupported() const override { return false; }
    cv::Ptr<cv::DescriptorMatcher> clone(bool emptyTrainData = false) const override {
        cv::Ptr<HipBFMatcher> m = cv::makePtr<HipBFMatcher>(cvNorm_, handle_);
        if (!emptyTrainData) m->trainDescCollection = trainDescCollection;
        return m;
    }

protected:
    void knnMatchImpl(cv::InputArray queryDescriptors, std::vector<std::vector<cv::DMatch>>& matches, int k,
                      cv::InputArrayOfArrays masks = cv::noArray(), bool compactResult = false) override {
        if (k != 2 || trainDescCollection.size() != 1 || !masks.empty())
            CV_Error(cv::Error::StsNotImplemented, "HipBFMatcher: k = 2, one train image, no masks");
        const cv::Mat Q = queryDescriptors.getMat(), T = trainDescCollection[0];
        matches.clear();
        if (Q.empty()) return;
        if (Q.channels() != 1 || T.channels() != 1 || (!T.empty() && (Q.cols != T.cols || Q.depth() != T.depth())) ||
            Q.depth() != (norm_ == SFM_NORM_HAMMING ? CV_8U : CV_32F))
            CV_Error(cv::Error::StsBadArg, "HipBFMatcher: descriptor type or width does not suit the norm");
        std::vector<int32_t> idx(2 * size_t(Q.rows));
        std::vector<float> dist(2 * size_t(Q.rows));
        const size_t t_step = T.empty() ? size_t(Q.step[0]) : size_t(T.step[0]);
        handle_.check(sfm_knn2(handle_.get(), norm_, Q.data, Q.rows, Q.step[0], T.data, T.rows, t_step, Q.cols,
                               idx.data(), dist.data()),
                      "sfm_knn2");
        matches.reserve(size_t(Q.rows));
        for (int i = 0; i < Q.rows; ++i) {
            std::vector<cv::DMatch> row;
            for (int j = 0; j < 2 && idx[2 * size_t(i) + j] >= 0; ++j)
                row.push_back(cv::DMatch(i, idx[2 * size_t(i) + j], 0, dist[2 * size_t(i) + j]));
            if (!compactResult || !row.empty()) matches.push_back(row);
        }
    }
    void radiusMatchImpl(cv::InputArray, std::vector<std::vector<cv::DMatch>>&, float,
                         cv::InputArrayOfArrays = cv::noArray(), bool = false) override {
        CV_Error(cv::Error::StsNotImplemented, "HipBFMatcher: radius matching is not on the reference's path");
    }

private:
    int cvNorm_, norm_;
    Handle handle_;
};

/* cv::SparsePyrLKOpticalFlow with flags 0 (Source Files/feature_processing.cpp:229, :241). */
class HipSparsePyrLK : public cv::SparsePyrLKOpticalFlow {
public:
    HipSparsePyrLK(cv::Size winSize = cv::Size(21, 21), int maxLevel = 3,
                   cv::TermCriteria crit = cv::TermCriteria(cv::TermCriteria::COUNT + cv::TermCriteria::EPS, 30, 0.01),
                   int flags = 0, double minEigThreshold = 1e-4, Handle h = Handle::shared())
        : winSize_(winSize), maxLevel_(maxLevel), crit_(crit), flags_(flags), minEig_(minEigThreshold), handle_(h) {}

    cv::Size getWinSize() const override { return winSize_; }
    void setWinSize(cv::Size winSize) override { winSize_ = winSize; }
    int getMaxLevel() const override { return maxLevel_; }
    void setMaxLevel(int maxLevel) override { maxLevel_ = maxLevel; }
    cv::TermCriteria getTermCriteria() const override { return crit_; }
    void setTermCriteria(cv::TermCriteria& crit) override { crit_ = crit; }
    int getFlags() const override { return flags_; }
    void setFlags(int flags) override { flags_ = flags; }
    double getMinEigThreshold() const override { return minEig_; }
    void setMinEigThreshold(double minEigThreshold) override { minEig_ = minEigThreshold; }

    void calc(cv::InputArray prevImg, cv::InputArray nextImg, cv::InputArray prevPts, cv::InputOutputArray nextPts,
              cv::OutputArray status, cv::OutputArray err = cv::noArray()) override {
        if (flags_ != 0) CV_Error(cv::Error::StsNotImplemented, "HipSparsePyrLK: flags must be 0");
        const cv::Mat prev = prevImg.getMat(), next = nextImg.getMat(), pts = prevPts.getMat();
        const int n = int(pts.total());
        if (pts.type() != CV_32FC2 || !pts.isContinuous())
            CV_Error(cv::Error::StsBadArg, "HipSparsePyrLK: points must be a continuous array of Point2f");
        nextPts.create(pts.size(), pts.type(), -1, true);
        status.create(n, 1, CV_8U, -1, true);
        if (err.needed()) err.create(n, 1, CV_32F, -1, true);
        if (n == 0) return;
        if (prev.type() != CV_8UC1 || next.type() != CV_8UC1 || prev.rows != next.rows || prev.cols != next.cols)
            CV_Error(cv::Error::StsBadArg, "HipSparsePyrLK: two 8-bit single-channel images of one size");
        cv::Mat out = nextPts.getMat(), st = status.getMat(), er = err.needed() ? err.getMat() : cv::Mat();
        /* no error gate here: OptFlow::computeFlow applies its own (feature_processing.cpp:245-248) */
        handle_.check(sfm_pyrlk(handle_.get(), prev.data, next.data, prev.cols, prev.rows, prev.step[0], next.step[0],
                                pts.ptr<float>(), n, winSize_.width, winSize_.height, maxLevel_, crit_.type,
                                crit_.maxCount, crit_.epsilon, minEig_, 3.402823466e+38f, out.ptr<float>(),
                                st.ptr<unsigned char>(), er.empty() ? nullptr : er.ptr<float>()),
                      "sfm_pyrlk");
    }

private:
    cv::Size winSize_;
    int maxLevel_;
    cv::TermCriteria crit_;
    int flags_;
    double minEig_;
    Handle handle_;
};
#endif /* SFM_ADAPTER_HAVE_OPENCV */

}  // namespace sfm_hip
#endif /* SFM_HIP_ADAPTER_HPP */
