// Declarations-only mirror (see core.hpp in this directory); the adapter only includes it.
#ifndef SFM_TEST_OPENCV_DECL_CALIB3D_HPP
#define SFM_TEST_OPENCV_DECL_CALIB3D_HPP
#include <opencv2/core.hpp>
namespace cv {
enum { LMEDS = 4, RANSAC = 8 };
}
#endif
