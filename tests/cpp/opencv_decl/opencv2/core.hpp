// Declarations-only mirror of the part of OpenCV 4's public core API that
// include/sfm_hip_adapter.hpp touches: same names, same namespaces (uchar is global,
// everything else in cv), same member names and call signatures.  Nothing here is
// implemented; tests/test_cpu_host.py only COMPILES the adapter's OpenCV branch against it
// (no OpenCV is installed where the tests run), so that a name the adapter gets wrong shows up
// before a maintainer's first real build.  Written from OpenCV's documented interface.
#ifndef SFM_TEST_OPENCV_DECL_CORE_HPP
#define SFM_TEST_OPENCV_DECL_CORE_HPP

#include <cstddef>
#include <memory>
#include <string>
#include <vector>

typedef unsigned char uchar;
typedef unsigned short ushort;
typedef std::string String_;

#define CV_CN_SHIFT 3
#define CV_8U 0
#define CV_8S 1
#define CV_16U 2
#define CV_16S 3
#define CV_32S 4
#define CV_32F 5
#define CV_64F 6
#define CV_MAKETYPE(depth, cn) ((depth) + (((cn)-1) << CV_CN_SHIFT))
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_8UC3 CV_MAKETYPE(CV_8U, 3)
#define CV_32FC1 CV_MAKETYPE(CV_32F, 1)
#define CV_32FC2 CV_MAKETYPE(CV_32F, 2)

namespace cv {

typedef std::string String;

template <typename T>
using Ptr = std::shared_ptr<T>;
template <typename T, typename... A>
Ptr<T> makePtr(A&&... a) {
    return std::make_shared<T>(std::forward<A>(a)...);
}

namespace Error {
enum Code { StsOk = 0, StsError = -2, StsBadArg = -5, StsNotImplemented = -213, StsAssert = -215 };
}
class Exception : public std::exception {
public:
    Exception(int code, const String& err, const String& func, const String& file, int line);
    const char* what() const noexcept override;
    String msg;
    int code;
};
[[noreturn]] void error(int code, const String& err, const char* func, const char* file, int line);
#define CV_Error(code, msg) cv::error(code, msg, __func__, __FILE__, __LINE__)
#define CV_Assert(expr) \
    do {                \
        if (!(expr)) cv::error(cv::Error::StsAssert, #expr, __func__, __FILE__, __LINE__); \
    } while (0)

template <typename T>
class Point_ {
public:
    Point_();
    Point_(T x, T y);
    T x, y;
};
typedef Point_<float> Point2f;
typedef Point_<int> Point;

template <typename T>
class Size_ {
public:
    Size_();
    Size_(T width, T height);
    T width, height;
};
typedef Size_<int> Size;

template <typename T, int m, int n>
class Matx {
public:
    Matx();
    static Matx eye();
    const T& operator()(int row, int col) const;
    T& operator()(int row, int col);
    T val[m * n];
};
typedef Matx<double, 3, 3> Matx33d;
typedef Matx<double, 3, 4> Matx34d;
typedef Matx<double, 3, 1> Matx31d;

template <typename T, int cn>
class Vec : public Matx<T, cn, 1> {
public:
    Vec();
    const T& operator[](int i) const;
    T& operator[](int i);
};
typedef Vec<double, 3> Vec3d;
typedef Vec<uchar, 3> Vec3b;

class TermCriteria {
public:
    enum Type { COUNT = 1, MAX_ITER = COUNT, EPS = 2 };
    TermCriteria();
    TermCriteria(int type, int maxCount, double epsilon);
    int type;
    int maxCount;
    double epsilon;
};

class KeyPoint {
public:
    KeyPoint();
    KeyPoint(float x, float y, float size, float angle = -1, float response = 0, int octave = 0, int class_id = -1);
    Point2f pt;
    float size;
    float angle;
    float response;
    int octave;
    int class_id;
};

class DMatch {
public:
    DMatch();
    DMatch(int queryIdx, int trainIdx, float distance);
    DMatch(int queryIdx, int trainIdx, int imgIdx, float distance);
    int queryIdx;
    int trainIdx;
    int imgIdx;
    float distance;
};

struct MatStep {
    MatStep();
    const size_t& operator[](int i) const;
    size_t& operator[](int i);
    operator size_t() const;
    size_t* p;
    size_t buf[2];
};

class Mat {
public:
    Mat();
    Mat(int rows, int cols, int type);
    Mat(int rows, int cols, int type, void* data, size_t step = 0);
    Mat(const Mat&);
    Mat& operator=(const Mat&);
    ~Mat();
    int type() const;
    int depth() const;
    int channels() const;
    bool empty() const;
    size_t total() const;
    bool isContinuous() const;
    Size size() const;
    void create(int rows, int cols, int type);
    template <typename T> T* ptr(int row = 0);
    template <typename T> const T* ptr(int row = 0) const;
    int flags;
    int dims;
    int rows, cols;
    uchar* data;
    MatStep step;
};

class _InputArray {
public:
    _InputArray();
    _InputArray(const Mat& m);
    template <typename T> _InputArray(const std::vector<T>& vec);
    Mat getMat(int idx = -1) const;
    void getMatVector(std::vector<Mat>& mv) const;
    int type(int i = -1) const;
    int depth(int i = -1) const;
    int channels(int i = -1) const;
    size_t total(int i = -1) const;
    bool empty() const;
};
class _OutputArray : public _InputArray {
public:
    _OutputArray();
    _OutputArray(Mat& m);
    template <typename T> _OutputArray(std::vector<T>& vec);
    bool needed() const;
    void create(int rows, int cols, int type, int i = -1, bool allowTransposed = false, int fixedDepthMask = 0) const;
    void create(Size sz, int type, int i = -1, bool allowTransposed = false, int fixedDepthMask = 0) const;
};
class _InputOutputArray : public _OutputArray {
public:
    _InputOutputArray();
    _InputOutputArray(Mat& m);
    template <typename T> _InputOutputArray(std::vector<T>& vec);
};
typedef const _InputArray& InputArray;
typedef InputArray InputArrayOfArrays;
typedef const _OutputArray& OutputArray;
typedef OutputArray OutputArrayOfArrays;
typedef const _InputOutputArray& InputOutputArray;
InputOutputArray noArray();

class FileStorage;
class FileNode;
class Algorithm {
public:
    Algorithm();
    virtual ~Algorithm();
    virtual void clear();
    virtual void write(FileStorage& fs) const;
    virtual void read(const FileNode& fn);
    virtual bool empty() const;
    virtual void save(const String& filename) const;
    virtual String getDefaultName() const;
};

}  // namespace cv
#endif
