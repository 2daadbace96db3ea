// Declarations-only mirror of cv::SparsePyrLKOpticalFlow (see ../core.hpp).
#ifndef SFM_TEST_OPENCV_DECL_VIDEO_TRACKING_HPP
#define SFM_TEST_OPENCV_DECL_VIDEO_TRACKING_HPP
#include <opencv2/core.hpp>

namespace cv {

class SparseOpticalFlow : public Algorithm {
public:
    virtual void calc(InputArray prevImg, InputArray nextImg, InputArray prevPts, InputOutputArray nextPts,
                      OutputArray status, OutputArray err = noArray()) = 0;
};

class SparsePyrLKOpticalFlow : public SparseOpticalFlow {
public:
    virtual Size getWinSize() const = 0;
    virtual void setWinSize(Size winSize) = 0;
    virtual int getMaxLevel() const = 0;
    virtual void setMaxLevel(int maxLevel) = 0;
    virtual TermCriteria getTermCriteria() const = 0;
    virtual void setTermCriteria(TermCriteria& crit) = 0;
    virtual int getFlags() const = 0;
    virtual void setFlags(int flags) = 0;
    virtual double getMinEigThreshold() const = 0;
    virtual void setMinEigThreshold(double minEigThreshold) = 0;
    static Ptr<SparsePyrLKOpticalFlow> create(Size winSize = Size(21, 21), int maxLevel = 3,
                                              TermCriteria crit = TermCriteria(TermCriteria::COUNT + TermCriteria::EPS,
                                                                               30, 0.01),
                                              int flags = 0, double minEigThreshold = 1e-4);
};

}  // namespace cv
#endif
