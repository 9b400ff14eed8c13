// mi_cv_interpose.cpp -- LD_PRELOAD interposer for UNMODIFIED OpenCV 4.4 programs.
//
// The reference's prebuilt binaries import exactly two functions of the hot path from libopencv_imgproc.so.4.4
// (`readelf --dyn-syms`, demangled):
//     cv::equalizeHist(cv::_InputArray const&, cv::_OutputArray const&)      OpenCVequalHist.cpp:145, nextimprovement.cpp:168
//     cv::createCLAHE(double, cv::Size_<int>)                               clahevideo.cpp:184/:497, clahe1frame.cpp:88
// A shared object that defines those two symbols and is loaded first (LD_PRELOAD) takes their place for every caller in
// the process, so a program built against OpenCV runs its luma op on the MI355X without being relinked:
//     LD_PRELOAD=libmi_cv_interpose.so ./histequalize ...
// Everything else (Mat, cvtColor, imread, GStreamer glue) still comes from the real OpenCV the program links.
//
// Build (needs the REAL OpenCV 4.x headers of the target machine: the classes' layout is part of the ABI):
//     make -C opencv-opencl_amd/cxx interpose        (pkg-config opencv4)
// The authoring image has no OpenCV; tests/test_cxx_adapter.py compiles this file against declaration-only headers as a
// syntax check, nothing more.
#define MI_CV_WITH_OPENCV
#include "../mi_cv.hpp"

#ifndef MI_CV_HAVE_OPENCV_FRONT_END
#error "mi_cv_interpose.cpp needs <opencv2/core.hpp> and <opencv2/imgproc.hpp> on the include path"
#endif

namespace cv {

void equalizeHist(InputArray src, OutputArray dst) { mi_cv::equalizeHist(src, dst); }

Ptr<CLAHE> createCLAHE(double clipLimit, Size tileGridSize) { return mi_cv::createCLAHE(clipLimit, tileGridSize); }

}  // namespace cv
