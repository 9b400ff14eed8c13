// opencv2/core.hpp -- DECLARATION-ONLY stand-in used by tests/test_cxx_adapter.py::test_opencv_front_end_compiles.
//
// This is NOT OpenCV and pins nothing.  (Round 6: extended by what tests/cxx/test_adapter_opencv.cpp and interpose_probe.cpp touch --
// Rect, Mat's ROI / external-data constructors, split / merge, getBuildInformation, CV_VERSION -- so that the programs which make
// first contact with a REAL OpenCV elsewhere at least go through a compiler here.)  OpenCV 4.4 is not installed in the authoring image, so the
// real-cv::Mat front end of cxx/mi_cv.hpp (namespace mi_cv) and the interposer (cxx/interpose/) had never been
// through a compiler.  This header declares, with OpenCV 4.4's public names and signatures, exactly the API
// surface those two files touch -- nothing is implemented -- so that a syntax / override check can run on the CPU:
// every pure virtual of cv::CLAHE must be overridden, the InputArray / OutputArray calls must exist, cv::error must
// take those arguments.  Objects are compiled (-c), never linked or run.
#ifndef MI_TEST_OPENCV_DECL_CORE_HPP_
#define MI_TEST_OPENCV_DECL_CORE_HPP_

#include <cstddef>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#define CV_VERSION_MAJOR 4
#define CV_VERSION_MINOR 4
#define CV_VERSION "4.4.0-declarations-only"
#define CV_OVERRIDE override
#define CV_EXPORTS
#define CV_EXPORTS_W
#define CV_CN_SHIFT 3
#define CV_8U 0
#define CV_16U 2
#define CV_MAKETYPE(depth, cn) (((depth) & 7) + (((cn) - 1) << CV_CN_SHIFT))
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_16UC1 CV_MAKETYPE(CV_16U, 1)
#define CV_8UC3 CV_MAKETYPE(CV_8U, 3)

namespace cv {

typedef std::string String;
typedef unsigned char uchar;

namespace Error { enum Code { StsNoMem = -4, StsBadArg = -5, StsAssert = -215, GpuNotSupported = -216, GpuApiCallError = -217 }; }

[[noreturn]] void error(int _code, const String& _err, const char* _func, const char* _file, int _line);

#define CV_Assert(expr) do { if (!!(expr)) ; else cv::error(cv::Error::StsAssert, #expr, __func__, __FILE__, __LINE__); } while (0)

template <typename _Tp> class Size_ {
public:
    Size_();
    Size_(_Tp _width, _Tp _height);
    _Tp width, height;
};
typedef Size_<int> Size2i;
typedef Size2i Size;

template <typename _Tp> class Rect_ {
public:
    Rect_();
    Rect_(_Tp _x, _Tp _y, _Tp _width, _Tp _height);
    _Tp x, y, width, height;
};
typedef Rect_<int> Rect;

const String& getBuildInformation();

template <typename T> struct Ptr : public std::shared_ptr<T> {
    Ptr() = default;
    Ptr(const std::shared_ptr<T>& o) : std::shared_ptr<T>(o) {}
    template <typename Y> Ptr(const Ptr<Y>& o) : std::shared_ptr<T>(o) {}
};
template <typename _Tp, typename... A1> static inline Ptr<_Tp> makePtr(const A1&... a1) { return Ptr<_Tp>(std::make_shared<_Tp>(a1...)); }

struct MatStep {
    operator size_t() const;
    size_t* p;
};

class Mat {
public:
    Mat();
    Mat(int rows, int cols, int type);
    Mat(int rows, int cols, int type, void* data, size_t step = 0);
    Mat(const Mat& m, const Rect& roi);
    Mat operator()(const Rect& roi) const;
    Mat clone() const;
    int type() const;
    Size size() const;
    bool empty() const;
    bool isContinuous() const;
    size_t elemSize() const;
    template <typename _Tp> _Tp* ptr(int i0 = 0);
    template <typename _Tp> const _Tp* ptr(int i0 = 0) const;
    int flags, dims, rows, cols;
    uchar* data;
    MatStep step;
};

class _InputArray {
public:
    _InputArray();
    _InputArray(const Mat& m);
    _InputArray(const std::vector<Mat>& vec);
    Mat getMat(int idx = -1) const;
    int type(int i = -1) const;
    bool empty() const;
};
class _OutputArray : public _InputArray {
public:
    _OutputArray();
    _OutputArray(Mat& m);
    _OutputArray(std::vector<Mat>& vec);
    void create(Size sz, int type, int i = -1, bool allowTransposed = false, int fixedDepthMask = 0) const;
};
typedef const _InputArray& InputArray;
typedef const _OutputArray& OutputArray;
typedef InputArray InputArrayOfArrays;
typedef OutputArray OutputArrayOfArrays;

void split(InputArray m, OutputArrayOfArrays mv);
void merge(InputArrayOfArrays mv, OutputArray dst);

class FileStorage;
class FileNode;

class Algorithm {
public:
    Algorithm();
    virtual ~Algorithm();
    virtual void clear() {}
    virtual void write(FileStorage& fs) const { (void)fs; }
    virtual void read(const FileNode& fn) { (void)fn; }
    virtual bool empty() const { return false; }
    virtual void save(const String& filename) const;
    virtual String getDefaultName() const;
};

}  // namespace cv
#endif
