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
// An OpenCV built with FMA contraction (aarch64) computes CLAHE's float steps differently from an x86-64 baseline build: set
// MI_CV_CLAHE_FP_CONTRACT=1 in the program's environment there (mi_cv.hpp thread_ctx; tests/cxx/test_adapter_opencv tells which).
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

#include <atomic>

// How often each of the two was TAKEN: a program (tests/cxx/interpose_probe.cpp) finds this function with dlsym(RTLD_DEFAULT, ...) when
// the interposer is loaded, and does not when it is not -- the proof that its cv::equalizeHist / cv::createCLAHE calls really ran here.
// Low 32 bits: cv::equalizeHist calls; high 32 bits: cv::createCLAHE calls.
static std::atomic<unsigned long long> g_taken{0};
extern "C" unsigned long long mi_cv_interpose_calls(void) { return g_taken.load(std::memory_order_relaxed); }

namespace cv {

void equalizeHist(InputArray src, OutputArray dst)
{
    g_taken.fetch_add(1ull, std::memory_order_relaxed);
    mi_cv::equalizeHist(src, dst);
}

Ptr<CLAHE> createCLAHE(double clipLimit, Size tileGridSize)
{
    g_taken.fetch_add(1ull << 32, std::memory_order_relaxed);
    return mi_cv::createCLAHE(clipLimit, tileGridSize);
}

}  // namespace cv
