// interpose_probe.cpp -- an UNMODIFIED OpenCV program, as the reference's prebuilt binaries are: it includes and links OpenCV and
// nothing of this repository, and calls plain cv::equalizeHist (OpenCVequalHist.cpp:145) and cv::createCLAHE(...)->apply
// (clahevideo.cpp:184-195).  tests/test_opencv_pin.py runs it twice, plain and under LD_PRELOAD=libmi_cv_interpose.so: the bytes it
// computes must not change, and under the interposer `interposed_calls` must say that both calls were taken there (the interposer
// exports a counter; without it the symbol does not exist).  Built by `make -C tests/cxx opencv` when pkg-config knows opencv4.
#include <opencv2/core.hpp>
#include <opencv2/imgproc.hpp>

#include <cstdint>
#include <cstdio>
#include <dlfcn.h>

static uint64_t fnv(const cv::Mat& m)
{
    uint64_t h = 1469598103934665603ull;
    for (int y = 0; y < m.rows; ++y) {
        const unsigned char* p = m.ptr<unsigned char>(y);
        for (int x = 0; x < m.cols; ++x) { h ^= p[x]; h *= 1099511628211ull; }
    }
    return h;
}

int main()
{
    cv::Mat src(720, 1280, CV_8UC1), eq, cl;
    uint64_t s = 0x5EED0000ull;
    for (int y = 0; y < src.rows; ++y) {
        unsigned char* p = src.ptr<unsigned char>(y);
        for (int x = 0; x < src.cols; ++x) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; p[x] = (unsigned char)(96 + (x + y) / 16 % 64 + (int)((s >> 33) % 49) - 24); }
    }
    cv::equalizeHist(src, eq);
    cv::Ptr<cv::CLAHE> clahe = cv::createCLAHE(2.0, cv::Size(8, 8));
    clahe->apply(src, cl);
    typedef unsigned long long (*counter_fn)(void);
    const counter_fn counter = (counter_fn)dlsym(RTLD_DEFAULT, "mi_cv_interpose_calls");
    const unsigned long long taken = counter ? counter() : 0ull;
    printf("{\"equalize_fnv\": \"%016llx\", \"clahe_fnv\": \"%016llx\", \"interposer_loaded\": %s, \"equalizeHist_taken\": %llu, \"createCLAHE_taken\": %llu}\n",
           (unsigned long long)fnv(eq), (unsigned long long)fnv(cl), counter ? "true" : "false", taken & 0xffffffffull, taken >> 32);
    return 0;
}
