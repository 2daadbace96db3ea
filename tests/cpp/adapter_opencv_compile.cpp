// Compile-only check of the OpenCV branch of include/sfm_hip_adapter.hpp against the
// declarations-only headers in tests/cpp/opencv_decl (no OpenCV is installed where the tests
// run).  Every member a reference call site uses is named once, with the reference's own
// argument types, so that a wrong name, namespace or signature fails here.
#include "sfm_hip_adapter.hpp"

#ifndef SFM_ADAPTER_HAVE_OPENCV
#error "the OpenCV branch of the adapter was not selected"
#endif

struct Camera {
    cv::Matx33d K33d;
};

void reference_call_sites(cv::Mat gray0, cv::Mat gray1, cv::Mat color, cv::Mat desc0, cv::Mat desc1,
                          std::vector<cv::KeyPoint> kp0, std::vector<cv::KeyPoint> kp1, cv::Mat cameraK33) {
    // app_solver.cpp:105-118
    cv::Ptr<cv::FeatureDetector> orb;
    sfm_hip::FeatureDetector featDetector("ORB", orb, orb);
    sfm_hip::DescriptorMatcher descMatcher("BRUTEFORCE_HAMMING", 0.5f, false, cv::Size(640, 480));
    cv::TermCriteria crit(cv::TermCriteria::COUNT | cv::TermCriteria::EPS, 250, 0.0);
    sfm_hip::OptFlow optFlow(crit, 30, 5, 10.0f, 300, 0.01f, 20.0f, 100);
    sfm_hip::Reconstruction reconstruction("ITERATIVE", "DENSE_SCHUR", 1.0, 0.0001f, 250.0f, 3.0f, true);
    sfm_hip::RecoveryPose recPose("RANSAC", 0.99, 0.5, 10, "SOLVEPNP_P3P", false, 250);

    // feature_processing.cpp:77-99
    featDetector.generateFeatures(gray0, kp0, desc0);
    std::vector<cv::Point2f> prevPts, currPts;
    featDetector.generateFlowFeatures(gray0, prevPts, 300, 0.01, 20.0);

    // app_solver.cpp:68, :366
    optFlow.computeFlow(gray0, gray1, prevPts, currPts, optFlow.statusMask);
    std::vector<uchar>& status = optFlow.statusMask;   // the reference's element type
    (void)status;
    (void)optFlow.optFlow->getWinSize();
    optFlow.additionalSettings.setMaxError(5.0f);

    // app_solver.cpp:404, tracking.cpp:135
    std::vector<cv::Point2f> prevAlig, currAlig;
    std::vector<cv::DMatch> matches;
    std::vector<int> prevIdx, currIdx;
    descMatcher.findRobustMatches(kp0, kp1, desc0, desc1, prevAlig, currAlig, matches, prevIdx, currIdx, gray0, gray1,
                                  true);
    descMatcher.ratioMaches(desc0, desc1, matches);

    // tracking.cpp:91-113 (cv::Mat cameraK converts through Matx33d at the call site)
    cv::Matx33d K;
    int numInliers = 0;
    (void)sfm_hip::Tracking::findCameraPose(recPose, prevAlig, currAlig, K, 10, numInliers);

    // app_solver.cpp:435
    Camera camera;
    std::vector<cv::Vec3d> points3D;
    std::vector<cv::Vec3b> pointsRGB;
    std::vector<bool> mask;
    cv::Matx34d prevPose, currPose;
    cv::Matx33d R;
    cv::Matx31d t;
    reconstruction.triangulateCloud(camera, prevAlig, currAlig, color, points3D, pointsRGB, mask, prevPose, currPose,
                                    R, t);

    // Seam 2: OpenCV's own operator interfaces, assigned to the reference wrappers' public handles
    cv::Ptr<cv::DescriptorMatcher> matcher = cv::makePtr<sfm_hip::HipBFMatcher>(cv::NORM_HAMMING);
    std::vector<std::vector<cv::DMatch>> knn;
    matcher->knnMatch(desc0, desc1, knn, 2);             // feature_processing.cpp:121
    cv::Ptr<cv::SparsePyrLKOpticalFlow> lk = cv::makePtr<sfm_hip::HipSparsePyrLK>(cv::Size(30, 30), 5, crit);
    std::vector<float> err;
    lk->calc(gray0, gray1, prevPts, currPts, status, err);   // feature_processing.cpp:241
    (void)cameraK33;
}
