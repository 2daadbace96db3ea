// Drives the C++ adapter (include/sfm_hip_adapter.hpp) the way the reference's
// main loop drives its wrapper classes (Source Files/app_solver.cpp:105-118
// construction, :182 computeFlow, :404 findRobustMatches, :435 triangulateCloud)
// on inputs a test writes to a file, and writes the results to another file for
// the test to compare.  All values little-endian, packed.
//
//   adapter_check match  in out
//   adapter_check flow   in out
//   adapter_check tri    in out
//   adapter_check errors            (exit 0 when every misuse raised)
#define SFM_ADAPTER_NO_OPENCV 1
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "sfm_hip_adapter.hpp"

namespace cvx = sfm_hip::cvx;

struct Reader {
    FILE* f;
    explicit Reader(const char* path) : f(std::fopen(path, "rb")) {
        if (!f) { std::perror(path); std::exit(2); }
    }
    ~Reader() { std::fclose(f); }
    void raw(void* p, size_t n) {
        if (n && std::fread(p, 1, n, f) != n) { std::fprintf(stderr, "short read\n"); std::exit(2); }
    }
    int i32() { int32_t v; raw(&v, 4); return v; }
    float f32() { float v; raw(&v, 4); return v; }
    double f64() { double v; raw(&v, 8); return v; }
};

struct Writer {
    FILE* f;
    explicit Writer(const char* path) : f(std::fopen(path, "wb")) {
        if (!f) { std::perror(path); std::exit(2); }
    }
    ~Writer() { std::fclose(f); }
    void raw(const void* p, size_t n) {
        if (n && std::fwrite(p, 1, n, f) != n) { std::fprintf(stderr, "short write\n"); std::exit(2); }
    }
    void i32(int v) { int32_t w = v; raw(&w, 4); }
};

static int run_match(const char* in, const char* out) {
    Reader r(in);
    const int norm = r.i32(), n_prev = r.i32(), n_curr = r.i32(), d = r.i32();
    const float ratio = r.f32();
    const int use_filter = r.i32();
    const cvx::Mat::Depth depth = norm == SFM_NORM_HAMMING ? cvx::Mat::U8 : cvx::Mat::F32;
    cvx::Mat prevDesc(n_prev, d, depth), currDesc(n_curr, d, depth);
    r.raw(prevDesc.data, size_t(n_prev) * prevDesc.step);
    r.raw(currDesc.data, size_t(n_curr) * currDesc.step);
    std::vector<cvx::KeyPoint> prevKp, currKp;
    if (use_filter == 2) {  // key point positions follow the descriptors: the library's own filter runs on them
        std::vector<float> xy(size_t(2) * size_t(n_prev + n_curr));
        r.raw(xy.data(), xy.size() * 4);
        for (int i = 0; i < n_prev; ++i) prevKp.emplace_back(xy[2 * i], xy[2 * i + 1], 7.f);
        for (int i = n_prev; i < n_prev + n_curr; ++i) currKp.emplace_back(xy[2 * i], xy[2 * i + 1], 7.f);
    } else {
        for (int i = 0; i < n_prev; ++i) prevKp.emplace_back(float(i) + 0.25f, float(2 * i), 7.f);
        for (int i = 0; i < n_curr; ++i) currKp.emplace_back(float(3 * i), float(i) + 0.5f, 7.f);
    }

    const char* method = norm == SFM_NORM_HAMMING ? "BruteForce_Hamming" : norm == SFM_NORM_L2SQR ? "bruteforce_sl2" : "BruteForce";
    sfm_hip::DescriptorMatcher matcher(method, ratio);
    sfm_hip::DescriptorMatcher byValue = matcher;  // the reference passes it by value
    if (use_filter == 1) {
        // a caller's own filter in place of the default: drops every third match
        byValue.epipolarFilter = [](const std::vector<cvx::Point2f>& a, const std::vector<cvx::Point2f>&, std::vector<uint8_t>& m) {
            m.assign(a.size(), 1);
            for (size_t i = 0; i < m.size(); i += 3) m[i] = 0;
        };
    }

    std::vector<cvx::DMatch> fwd(3);  // ratioMaches clears what it is given
    byValue.ratioMaches(prevDesc, currDesc, fwd);

    std::vector<cvx::Point2f> prevAlig, currAlig;
    std::vector<cvx::DMatch> matches;
    std::vector<int> prevIdx, currIdx;
    byValue.findRobustMatches(prevKp, currKp, prevDesc, currDesc, prevAlig, currAlig, matches, prevIdx, currIdx,
                              cvx::Mat(), cvx::Mat(), use_filter != 0);

    Writer w(out);
    w.i32(int(fwd.size()));
    w.raw(fwd.data(), fwd.size() * sizeof(cvx::DMatch));
    w.i32(int(matches.size()));
    w.raw(matches.data(), matches.size() * sizeof(cvx::DMatch));
    w.i32(int(prevAlig.size()));
    w.raw(prevAlig.data(), prevAlig.size() * 8);
    w.raw(currAlig.data(), currAlig.size() * 8);
    w.i32(int(prevIdx.size()));
    w.raw(prevIdx.data(), prevIdx.size() * 4);
    w.raw(currIdx.data(), currIdx.size() * 4);
    return 0;
}

static int run_flow(const char* in, const char* out) {
    Reader r(in);
    const int w_ = r.i32(), h = r.i32(), n = r.i32(), win = r.i32(), maxLevel = r.i32();
    const int critType = r.i32(), maxCount = r.i32();
    const double eps = r.f64();
    const float maxError = r.f32();
    cvx::Mat prev(h, w_, cvx::Mat::U8), curr(h, w_, cvx::Mat::U8);
    r.raw(prev.data, size_t(h) * w_);
    r.raw(curr.data, size_t(h) * w_);
    std::vector<cvx::Point2f> prevPts(static_cast<size_t>(n)), currPts(5);
    r.raw(prevPts.data(), size_t(n) * 8);

    // Source Files/app_solver.cpp:108-109
    sfm_hip::OptFlow optFlow(cvx::TermCriteria(critType, maxCount, eps), win, maxLevel, maxError, 2000u, 0.01f, 10.f, 30u);
    sfm_hip::OptFlow byValue = optFlow;
    byValue.computeFlow(prev, curr, prevPts, currPts, byValue.statusMask);

    Writer w(out);
    w.i32(int(currPts.size()));
    w.raw(currPts.data(), currPts.size() * 8);
    w.raw(byValue.statusMask.data(), byValue.statusMask.size());
    w.i32(byValue.optFlow->getMaxLevel());
    w.i32(byValue.optFlow->getWinSize().width);
    return 0;
}

static int run_corners(const char* in, const char* out) {
    Reader r(in);
    const int w_ = r.i32(), h = r.i32(), maxCorners = r.i32();
    const double quality = r.f64(), minDistance = r.f64();
    cvx::Mat img(h, w_, cvx::Mat::U8);
    r.raw(img.data, size_t(h) * w_);
    sfm_hip::FeatureDetector detector("ORB");
    sfm_hip::FeatureDetector byValue = detector;  // the reference passes it by value
    std::vector<cvx::Point2f> corners = {cvx::Point2f(1.5f, 2.5f)};  // kept, new ones are appended
    byValue.generateFlowFeatures(img, corners, maxCorners, quality, minDistance);
    Writer w(out);
    w.i32(int(corners.size()));
    w.raw(corners.data(), corners.size() * 8);
    return 0;
}

static int run_pose(const char* in, const char* out) {
    Reader r(in);
    const int n = r.i32(), minInliers = r.i32();
    const double prob = r.f64(), threshold = r.f64();
    cvx::Matx33d K;
    r.raw(K.val, 72);
    std::vector<cvx::Point2f> p1(static_cast<size_t>(n)), p2(static_cast<size_t>(n));
    r.raw(p1.data(), size_t(n) * 8);
    r.raw(p2.data(), size_t(n) * 8);
    // Source Files/app_solver.cpp constructs it from the peMethod / peProb / peThresh / peMinInl settings
    sfm_hip::RecoveryPose recPose("ransac", prob, threshold, unsigned(minInliers), "SOLVEPNP_P3P", false, 100u);
    int numInliers = -1;
    const bool ok = sfm_hip::Tracking::findCameraPose(recPose, p1, p2, K, minInliers, numInliers);
    // five points or fewer: false before anything is computed (Source Files/tracking.cpp:92)
    int untouched = -7;
    std::vector<cvx::Point2f> few(p1.begin(), p1.begin() + 5);
    if (sfm_hip::Tracking::findCameraPose(recPose, few, few, K, minInliers, untouched) || untouched != -7) return 3;
    Writer w(out);
    w.i32(ok ? 1 : 0);
    w.i32(numInliers);
    w.raw(recPose.R.val, 72);
    w.raw(recPose.t.val, 24);
    w.i32(recPose.mask.rows);
    w.raw(recPose.mask.data, size_t(recPose.mask.rows));
    return 0;
}

struct Camera {  // the part of CameraParameters triangulateCloud reads
    cvx::Matx33d K33d;
};

static int run_tri(const char* in, const char* out) {
    Reader r(in);
    const int n = r.i32(), w_ = r.i32(), h = r.i32();
    Camera cam;
    cvx::Matx34d prevPose, currPose;
    cvx::Matx33d R;
    cvx::Matx31d t;
    r.raw(cam.K33d.val, 72);
    r.raw(prevPose.val, 96);
    r.raw(currPose.val, 96);
    r.raw(R.val, 72);
    r.raw(t.val, 24);
    const float minD = r.f32(), maxD = r.f32(), maxErr = r.f32();
    std::vector<cvx::Point2f> p1(static_cast<size_t>(n)), p2(static_cast<size_t>(n));
    r.raw(p1.data(), size_t(n) * 8);
    r.raw(p2.data(), size_t(n) * 8);
    cvx::Mat bgr(h, w_, cvx::Mat::U8, 3);
    r.raw(bgr.data, size_t(h) * w_ * 3);

    sfm_hip::Reconstruction recon("ITERATIVE", "", 1.0, minD, maxD, maxErr, true);
    std::vector<cvx::Vec3d> cloud(2);      // cleared by the call
    std::vector<cvx::Vec3b> colours(2);    // cleared by the call
    std::vector<bool> mask(1, true);       // appended to
    recon.triangulateCloud(cam, p1, p2, bgr, cloud, colours, mask, prevPose, currPose, R, t);

    // empty input leaves everything untouched (Source Files/reconstruction.cpp:42)
    std::vector<cvx::Vec3d> keep(4);
    std::vector<cvx::Vec3b> keepc(4);
    std::vector<bool> keepm(4);
    recon.triangulateCloud(cam, std::vector<cvx::Point2f>(), p2, bgr, keep, keepc, keepm, prevPose, currPose, R, t);
    if (keep.size() != 4 || keepc.size() != 4 || keepm.size() != 4) return 3;

    Writer w(out);
    w.i32(int(cloud.size()));
    w.raw(cloud.data(), cloud.size() * 24);
    w.raw(colours.data(), colours.size() * 3);
    w.i32(int(mask.size()));
    for (bool b : mask) { uint8_t v = b; w.raw(&v, 1); }
    return 0;
}

template <class F>
static int must_raise(const char* what, F&& f) {
    try {
        f();
    } catch (const std::exception& e) {
        std::printf("raised as expected (%s): %s\n", what, e.what());
        return 0;
    }
    std::printf("NOT raised: %s\n", what);
    return 1;
}

static int run_errors() {
    int bad = 0;
    cvx::Mat u8(8, 32, cvx::Mat::U8), f32(8, 32, cvx::Mat::F32), u8w(8, 64, cvx::Mat::U8);
    std::vector<cvx::DMatch> m;
    sfm_hip::DescriptorMatcher ham("BRUTEFORCE_HAMMING", 0.5f), l2("anything else", 0.5f);
    if (ham.normType != SFM_NORM_HAMMING || l2.normType != SFM_NORM_L2) { std::printf("ctor mapping wrong\n"); ++bad; }
    bad += must_raise("float descriptors with Hamming", [&] { ham.ratioMaches(f32, f32, m); });
    bad += must_raise("byte descriptors with L2", [&] { l2.ratioMaches(u8, u8, m); });
    bad += must_raise("width mismatch", [&] { ham.ratioMaches(u8, u8w, m); });
    bad += must_raise("FLANN", [&] { sfm_hip::DescriptorMatcher f("FlannBased", 0.5f); });
    bad += must_raise("epipolar filter unset", [&] {
        std::vector<cvx::KeyPoint> kp(8);
        std::vector<cvx::Point2f> a, b;
        std::vector<int> ia, ib;
        // identical, distinct rows give cross-checked matches, so the filter is reached
        for (int i = 0; i < 8; ++i)
            for (int j = 0; j < 32; ++j) u8.data[i * 32 + j] = uint8_t(i == (j & 7) ? 0xFF : 0);
        sfm_hip::DescriptorMatcher noFilter = ham;
        noFilter.epipolarFilter = nullptr;
        noFilter.findRobustMatches(kp, kp, u8, u8, a, b, m, ia, ib, cvx::Mat(), cvx::Mat(), true);
    });
    // generateFeatures forwards to the caller's detector / extractor as the reference does
    // (feature_processing.cpp:77-85): one object -> detectAndCompute, two -> detect + compute
    {
        struct Counting : cvx::Feature2D {
            int detects = 0, computes = 0, both = 0;
            void detect(const cvx::Mat&, std::vector<cvx::KeyPoint>& k) override { ++detects; k.resize(3); }
            void compute(const cvx::Mat&, std::vector<cvx::KeyPoint>&, cvx::Mat& d) override {
                ++computes;
                d = cvx::Mat(3, 32, cvx::Mat::U8);
            }
            void detectAndCompute(const cvx::Mat&, const cvx::Mat&, std::vector<cvx::KeyPoint>& k, cvx::Mat& d) override {
                ++both;
                k.resize(5);
                d = cvx::Mat(5, 32, cvx::Mat::U8);
            }
        };
        auto one = std::make_shared<Counting>(), two = std::make_shared<Counting>();
        cvx::Mat img(16, 16, cvx::Mat::U8), desc;
        std::vector<cvx::KeyPoint> kp;
        sfm_hip::FeatureDetector same("ORB", one, one), split("FAST", one, two), none("AKAZE");
        same.generateFeatures(img, kp, desc);
        if (one->both != 1 || one->detects != 0 || kp.size() != 5 || desc.rows != 5) { std::printf("detectAndCompute path wrong\n"); ++bad; }
        split.generateFeatures(img, kp, desc);
        if (one->detects != 1 || two->computes != 1 || two->both != 0 || desc.rows != 3) { std::printf("detect + compute path wrong\n"); ++bad; }
        if (same.detector != one || split.extractor != two) { std::printf("public handles wrong\n"); ++bad; }
        bad += must_raise("generateFeatures without a detector", [&] { none.generateFeatures(img, kp, desc); });
    }
    sfm_hip::OptFlow of(cvx::TermCriteria(3, 30, 0.01), 21, 3, 10.f, 100u, 0.01f, 10.f, 30u);
    bad += must_raise("colour image into computeFlow", [&] {
        cvx::Mat c3(32, 32, cvx::Mat::U8, 3);
        std::vector<cvx::Point2f> p(1, cvx::Point2f(16, 16)), q;
        of.computeFlow(c3, c3, p, q, of.statusMask);
    });
    sfm_hip::Reconstruction dlt("DLT", "", 1.0, 1e-4f, 250.f, 3.f, true);
    bad += must_raise("dead DLT branch", [&] {
        Camera cam;
        cvx::Matx34d P;
        cvx::Matx33d R;
        cvx::Matx31d t;
        std::vector<cvx::Point2f> p(1);
        std::vector<cvx::Vec3d> X;
        std::vector<cvx::Vec3b> c;
        std::vector<bool> k;
        dlt.triangulateCloud(cam, p, p, cvx::Mat(4, 4, cvx::Mat::U8, 3), X, c, k, P, P, R, t);
    });
    return bad;
}

int main(int argc, char** argv) {
    const std::string mode = argc > 1 ? argv[1] : "";
    try {
        if (mode == "errors") return run_errors();
        if (argc < 4) { std::fprintf(stderr, "usage: adapter_check match|flow|tri in out | errors\n"); return 2; }
        if (mode == "match") return run_match(argv[2], argv[3]);
        if (mode == "flow") return run_flow(argv[2], argv[3]);
        if (mode == "corners") return run_corners(argv[2], argv[3]);
        if (mode == "pose") return run_pose(argv[2], argv[3]);
        if (mode == "tri") return run_tri(argv[2], argv[3]);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "adapter_check: %s\n", e.what());
        return 4;
    }
    return 2;
}
