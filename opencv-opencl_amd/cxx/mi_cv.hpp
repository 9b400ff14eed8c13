// mi_cv.hpp -- header-only C++ adapter: the reference's cv::Mat-in / cv::Mat-out call surface on top
// of the C ABI (include/mi_lumaeq.h).  This is the host-side mirror of the interface the reference
// programs use for the hot path:
//
//   cv::equalizeHist(y_plane_in, y_plane_out);                       OpenCVequalHist.cpp:145
//   cv::equalizeHist(y_in /*view on input buffer*/, y_out /*view on output buffer*/);
//                                                                    nextimprovement.cpp:164-168
//   clahe = cv::createCLAHE(clip, cv::Size(t, t));                   clahevideo.cpp:184, :497
//   clahe->setClipLimit(..); clahe->setTilesGridSize(..);            clahevideo.cpp:187-188
//   clahe->apply(y_in /*ROI view, step = width*/, y_out);            clahevideo.cpp:195
//
// Semantics kept: src must be 8-bit single channel, otherwise an exception derived
// from std::exception is thrown (callers catch `const std::exception&`, OpenCVequalHist.cpp:189,
// clahevideo.cpp:273); an empty src is a no-op; dst is (re)created ONLY if its size/type does not
// already match, so a dst that wraps caller memory is written in place and never reallocated;
// src/dst may be ROI views (step > width) and may alias; the call is synchronous.
//
// Two front ends, same functions:
//   * namespace micv  -- a small self-contained Mat/Size/Rect/Ptr so the adapter (and its tests)
//     work where OpenCV is not installed (it is not in the authoring image);
//   * if <opencv2/core.hpp> is present and MI_CV_WITH_OPENCV is defined, overloads taking real
//     cv::Mat (`mi_cv::equalizeHist(const cv::Mat&, cv::Mat&)`, `mi_cv::createCLAHE`) are added;
//     a reference program switches by replacing `cv::equalizeHist` with `mi_cv::equalizeHist`
//     (INTEGRATION.md).
//
// Each calling thread lazily owns one mi_ctx per device ("one context per worker", the shape of
// OpenCLequalHist.cpp:142-152), so equalizeHist() is re-entrant across the reference's worker
// threads; a CLAHE object is not shared between threads (as in OpenCV).
#ifndef MI_CV_HPP_
#define MI_CV_HPP_

#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/mi_lumaeq.h"

namespace micv {

// ---- type codes (values as in OpenCV's interface.h so they mean the same thing to a reader) ----
constexpr int CV_8U = 0, CV_16U = 2, CV_32F = 5;
constexpr int MI_CN_SHIFT = 3;
constexpr int makeType(int depth, int cn) { return (depth & 7) + ((cn - 1) << MI_CN_SHIFT); }
constexpr int CV_8UC1 = makeType(CV_8U, 1), CV_8UC3 = makeType(CV_8U, 3), CV_16UC1 = makeType(CV_16U, 1),
              CV_32FC1 = makeType(CV_32F, 1);
inline int elemSizeOf(int type)
{
    static const int depth_bytes[8] = {1, 1, 2, 2, 4, 4, 8, 2};
    return depth_bytes[type & 7] * ((type >> MI_CN_SHIFT) + 1);
}

// ---- error type: mirrors cv::Exception's fields; derives from std::exception ----
class Exception : public std::exception {
public:
    Exception(int code_, std::string err_, std::string func_, std::string file_, int line_)
        : code(code_), err(std::move(err_)), func(std::move(func_)), file(std::move(file_)), line(line_)
    {
        msg = "mi_cv(" + std::to_string(code) + ") " + file + ":" + std::to_string(line) + ": error: " + err + " in function '" + func + "'";
    }
    const char* what() const noexcept override { return msg.c_str(); }
    int code;
    std::string err, func, file, msg;
    int line;
};
// OpenCV's numeric codes for the two cases that can occur on this path
constexpr int StsAssert = -215, StsBadArg = -5, StsNoMem = -4, GpuApiCallError = -217, GpuNotSupported = -216;

#define MI_CV_ERROR(code, text) throw ::micv::Exception((code), (text), __func__, __FILE__, __LINE__)
#define MI_CV_ASSERT(expr) \
    do { if (!(expr)) MI_CV_ERROR(::micv::StsAssert, #expr); } while (0)

struct Size {
    int width = 0, height = 0;
    Size() = default;
    Size(int w, int h) : width(w), height(h) {}
    int area() const { return width * height; }
    bool operator==(const Size& o) const { return width == o.width && height == o.height; }
    bool operator!=(const Size& o) const { return !(*this == o); }
};
struct Rect {
    int x = 0, y = 0, width = 0, height = 0;
    Rect() = default;
    Rect(int x_, int y_, int w, int h) : x(x_), y(y_), width(w), height(h) {}
};

template <class T> using Ptr = std::shared_ptr<T>;

// ---- Mat: 2-D, reference counted when it owns memory, a plain view when it wraps caller memory ----
class Mat {
public:
    static constexpr size_t AUTO_STEP = 0;
    int rows = 0, cols = 0;
    size_t step = 0;
    unsigned char* data = nullptr;

    Mat() = default;
    Mat(int r, int c, int type) { create(r, c, type); }
    Mat(Size s, int type) { create(s.height, s.width, type); }
    // wraps external memory (no copy, never freed): cv::Mat(rows, cols, type, data, step)
    Mat(int r, int c, int type, void* ext, size_t step_ = AUTO_STEP)
        : rows(r), cols(c), step(step_ == AUTO_STEP ? (size_t)c * elemSizeOf(type) : step_), data((unsigned char*)ext), type_(type)
    {
        MI_CV_ASSERT(r >= 0 && c >= 0 && step >= (size_t)c * elemSizeOf(type));
    }
    // ROI view: cv::Mat(const Mat&, const Rect&)
    Mat(const Mat& m, const Rect& roi) : rows(roi.height), cols(roi.width), step(m.step), type_(m.type_), owner_(m.owner_)
    {
        MI_CV_ASSERT(0 <= roi.x && 0 <= roi.width && roi.x + roi.width <= m.cols && 0 <= roi.y && 0 <= roi.height && roi.y + roi.height <= m.rows);
        data = m.data + (size_t)roi.y * m.step + (size_t)roi.x * elemSizeOf(m.type_);
    }
    Mat operator()(const Rect& roi) const { return Mat(*this, roi); }

    // (re)allocate only when size or type differ -- Mat::create semantics
    void create(int r, int c, int type)
    {
        MI_CV_ASSERT(r >= 0 && c >= 0);
        if (data && rows == r && cols == c && type_ == type) return;
        rows = r; cols = c; type_ = type; step = (size_t)c * elemSizeOf(type);
        const size_t bytes = step * (size_t)r;
        if (bytes == 0) { owner_.reset(); data = nullptr; return; }
        void* p = nullptr;
        if (posix_memalign(&p, 64, bytes) != 0) MI_CV_ERROR(StsNoMem, "allocation failed");
        owner_ = std::shared_ptr<unsigned char>((unsigned char*)p, [](unsigned char* q) { free(q); });
        data = owner_.get();
    }
    void create(Size s, int type) { create(s.height, s.width, type); }

    int type() const { return type_; }
    int channels() const { return (type_ >> MI_CN_SHIFT) + 1; }
    int depth() const { return type_ & 7; }
    size_t elemSize() const { return (size_t)elemSizeOf(type_); }
    Size size() const { return Size(cols, rows); }
    size_t total() const { return (size_t)rows * cols; }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    bool isContinuous() const { return rows <= 1 || step == (size_t)cols * elemSize(); }
    unsigned char* ptr(int y = 0) { return data + (size_t)y * step; }
    const unsigned char* ptr(int y = 0) const { return data + (size_t)y * step; }
    bool ownsMemory() const { return (bool)owner_; }

    Mat clone() const
    {
        Mat m;
        copyTo(m);
        return m;
    }
    void copyTo(Mat& dst) const
    {
        if (empty()) { dst = Mat(); return; }
        dst.create(rows, cols, type_);
        const size_t rb = (size_t)cols * elemSize();
        for (int y = 0; y < rows; ++y) memcpy(dst.ptr(y), ptr(y), rb);
    }
    Mat& setTo(unsigned char v)
    {
        const size_t rb = (size_t)cols * elemSize();
        for (int y = 0; y < rows; ++y) memset(ptr(y), v, rb);
        return *this;
    }

private:
    int type_ = CV_8UC1;
    std::shared_ptr<unsigned char> owner_;
};

// ---- per-thread contexts ----
namespace detail {
struct CtxDeleter { void operator()(mi_ctx* c) const { mi_ctx_destroy(c); } };
inline int& tls_device() { static thread_local int dev = 0; return dev; }
inline mi_ctx* thread_ctx()
{
    static thread_local std::map<int, std::unique_ptr<mi_ctx, CtxDeleter>> ctxs;
    const int dev = tls_device();
    auto it = ctxs.find(dev);
    if (it == ctxs.end()) {
        mi_ctx* c = nullptr;
        const mi_status st = mi_ctx_create(dev, &c);
        if (st != MI_OK)
            MI_CV_ERROR(st == MI_ERR_NO_DEVICE ? GpuNotSupported : GpuApiCallError,
                        std::string("mi_ctx_create(device=") + std::to_string(dev) + ") failed: " + mi_status_str(st) +
                            " (this backend has no CPU fallback)");
        it = ctxs.emplace(dev, std::unique_ptr<mi_ctx, CtxDeleter>(c)).first;
        // CLAHE's float steps as the program's OWN OpenCV build computes them: 0 = every multiply and add rounded (x86-64 baseline
        // builds, the default), 1 = GCC's FMA contraction (aarch64 builds such as the reference's board).  For programs that cannot
        // call setOption themselves -- an unmodified binary under the LD_PRELOAD interposer -- the environment decides;
        // tests/cxx/test_adapter_opencv finds out which of the two a given OpenCV needs.
        if (const char* e = std::getenv("MI_CV_CLAHE_FP_CONTRACT")) (void)mi_ctx_set_option(c, "clahe_fp_contract", std::atoi(e) != 0);
    }
    return it->second.get();
}
inline void check(mi_ctx* c, mi_status st, const char* what)
{
    if (st == MI_OK) return;
    const std::string detail = std::string(what) + ": " + mi_status_str(st) + " (" + mi_ctx_last_error_msg(c) + ")";
    switch (st) {
        case MI_ERR_BAD_ARG: MI_CV_ERROR(StsBadArg, detail);
        case MI_ERR_UNSUPPORTED: MI_CV_ERROR(StsAssert, detail);
        case MI_ERR_OOM: MI_CV_ERROR(StsNoMem, detail);
        case MI_ERR_NO_DEVICE: MI_CV_ERROR(GpuNotSupported, detail);
        default: MI_CV_ERROR(GpuApiCallError, detail);
    }
}
}  // namespace detail

// Device used by the calling thread's subsequent calls (a worker of an N-GPU pool calls this once).
inline void setDevice(int device) { detail::tls_device() = device; }
inline int getDevice() { return detail::tls_device(); }
inline int getDeviceCount() { return mi_device_count(); }
// Bind the CALLING thread to the CPUs next to `device` (its NUMA node) -- call it at the top of a worker, before the first
// micv:: call of that thread creates its context (mi_thread_bind_near_device; the reference's workers are not placed at all,
// OpenCVequalHist.cpp:397-402).  Returns the one-line description for a log; never throws for a platform without NUMA information.
inline std::string bindThreadNearDevice(int device)
{
    mi_numa_binding nb{};
    const mi_status st = mi_thread_bind_near_device(device, &nb);
    return st == MI_OK ? std::string(nb.why) : std::string("not bound: ") + mi_status_str(st);
}
// Library option of the calling thread's context on its current device (mi_ctx_set_option), e.g.
//   setOption("clahe_fp_contract", 1)  -- CLAHE interpolation with the fused multiply-adds a GCC build of OpenCV uses on
//   FMA targets (the reference's aarch64 board) instead of the separately rounded x86-64 baseline arithmetic.
inline void setOption(const char* name, int value)
{
    mi_ctx* c = detail::thread_ctx();
    detail::check(c, mi_ctx_set_option(c, name, value), "mi_ctx_set_option");
}

// ---- cv::equalizeHist(InputArray src, OutputArray dst) ----
inline void equalizeHist(const Mat& src, Mat& dst)
{
    MI_CV_ASSERT(src.type() == CV_8UC1);          // histogram.cpp: CV_Assert(_src.type() == CV_8UC1)
    if (src.empty()) return;
    const Mat s = src;                            // keep src alive/unchanged if dst is the same header
    dst.create(s.rows, s.cols, CV_8UC1);
    mi_ctx* c = detail::thread_ctx();
    detail::check(c, mi_equalize_hist_u8(c, s.data, s.step, dst.data, dst.step, s.cols, s.rows), "mi_equalize_hist_u8");
}

// ---- cv::CLAHE ----
class CLAHE {
public:
    virtual ~CLAHE() = default;
    virtual void apply(const Mat& src, Mat& dst) = 0;
    virtual void setClipLimit(double clipLimit) = 0;
    virtual double getClipLimit() const = 0;
    virtual void setTilesGridSize(Size tileGridSize) = 0;
    virtual Size getTilesGridSize() const = 0;
    virtual void collectGarbage() = 0;
};

namespace detail {
class CLAHE_Impl final : public CLAHE {
public:
    CLAHE_Impl(double clip, int tx, int ty) : clip_(clip), tx_(tx), ty_(ty) {}
    void apply(const Mat& src, Mat& dst) override
    {
        MI_CV_ASSERT(src.type() == CV_8UC1 || src.type() == CV_16UC1);     // clahe.cpp: CV_Assert(8UC1 || 16UC1)
        if (src.empty()) return;
        MI_CV_ASSERT(tx_ >= 1 && ty_ >= 1);
        const Mat s = src;
        dst.create(s.rows, s.cols, s.type());
        mi_ctx* c = thread_ctx();
        if (s.type() == CV_8UC1)
            check(c, mi_clahe_u8(c, s.data, s.step, dst.data, dst.step, s.cols, s.rows, clip_, tx_, ty_), "mi_clahe_u8");
        else                                                          // 65 536-bin path
            check(c, mi_clahe_u16(c, reinterpret_cast<const uint16_t*>(s.data), s.step, reinterpret_cast<uint16_t*>(dst.data), dst.step,
                                  s.cols, s.rows, clip_, tx_, ty_), "mi_clahe_u16");
    }
    void setClipLimit(double v) override { clip_ = v; }
    double getClipLimit() const override { return clip_; }
    void setTilesGridSize(Size s) override { tx_ = s.width; ty_ = s.height; }
    Size getTilesGridSize() const override { return Size(tx_, ty_); }
    void collectGarbage() override {}             // scratch lives in the per-thread mi_ctx, reused across frames
private:
    double clip_;
    int tx_, ty_;
};
}  // namespace detail

// cv::createCLAHE(double clipLimit = 40.0, Size tileGridSize = Size(8, 8))
inline Ptr<CLAHE> createCLAHE(double clipLimit = 40.0, Size tileGridSize = Size(8, 8))
{
    return std::make_shared<detail::CLAHE_Impl>(clipLimit, tileGridSize.width, tileGridSize.height);
}

// ---- colour-domain neighbours: the calls around the luma op in the reference's image benches ----
// cv::cvtColor(bgr, yuv, cv::COLOR_BGR2YUV) ... cv::cvtColor(yuv, bgr, cv::COLOR_YUV2BGR)   singlecolor.cpp:39/:66,
//                                                                                           clahe1frame.cpp:83/:102
constexpr int COLOR_BGR2YUV = MI_COLOR_BGR2YUV, COLOR_YUV2BGR = MI_COLOR_YUV2BGR;     // OpenCV's numeric values (82, 84)
// 4:2:0 codes: cv::cvtColor(bgr, yuv, cv::COLOR_BGR2YUV_I420) prepares the bench input at 1frameMeasure.cpp:32;
// COLOR_YUV2BGR_NV12 is its NV12 inverse (BASELINE config 5 read literally).  OpenCV's numeric values (128, 93).
constexpr int COLOR_BGR2YUV_I420 = MI_COLOR_BGR2YUV_I420, COLOR_YUV2BGR_NV12 = MI_COLOR_YUV2BGR_NV12;

inline void cvtColor(const Mat& src, Mat& dst, int code)
{
    MI_CV_ASSERT(code == COLOR_BGR2YUV || code == COLOR_YUV2BGR || code == COLOR_BGR2YUV_I420 || code == COLOR_YUV2BGR_NV12);
    if (code == COLOR_BGR2YUV_I420 || code == COLOR_YUV2BGR_NV12) {
        const bool enc = code == COLOR_BGR2YUV_I420;
        MI_CV_ASSERT(src.type() == (enc ? CV_8UC3 : CV_8UC1));
        if (src.empty()) return;
        const Mat s = src;                                               // keeps the data alive if dst aliases src
        const int w = s.cols, h = enc ? s.rows : s.rows * 2 / 3;
        MI_CV_ASSERT(w % 2 == 0 && (enc ? s.rows % 2 == 0 : s.rows % 3 == 0));   // OpenCV's own size checks
        dst.create(enc ? h * 3 / 2 : h, w, enc ? CV_8UC1 : CV_8UC3);
        mi_ctx* c = detail::thread_ctx();
        detail::check(c, mi_cvt_color_420_u8(c, s.data, s.step, dst.data, dst.step, w, h, code), "mi_cvt_color_420_u8");
        return;
    }
    MI_CV_ASSERT(src.type() == CV_8UC3);
    if (src.empty()) return;
    const Mat s = src;
    dst.create(s.rows, s.cols, CV_8UC3);
    mi_ctx* c = detail::thread_ctx();
    detail::check(c, mi_cvt_color_u8c3(c, s.data, s.step, dst.data, dst.step, s.cols, s.rows, code), "mi_cvt_color_u8c3");
}

// cv::split / cv::merge for 8-bit images (singlecolor.cpp:44/:61, clahe1frame.cpp:85/:100).  Pure byte shuffles of
// host Mats, kept on the host like Mat::clone/copyTo; the GPU pipeline below fuses them into the conversions instead.
inline void split(const Mat& src, std::vector<Mat>& planes)
{
    MI_CV_ASSERT(src.depth() == CV_8U);
    const int cn = src.channels();
    planes.resize((size_t)cn);
    for (int k = 0; k < cn; ++k) planes[(size_t)k].create(src.rows, src.cols, CV_8UC1);
    for (int y = 0; y < src.rows; ++y) {
        const unsigned char* s = src.ptr(y);
        for (int k = 0; k < cn; ++k) {
            unsigned char* d = planes[(size_t)k].ptr(y);
            for (int x = 0; x < src.cols; ++x) d[x] = s[(size_t)x * cn + k];
        }
    }
}
inline void merge(const std::vector<Mat>& planes, Mat& dst)
{
    MI_CV_ASSERT(!planes.empty());
    const int cn = (int)planes.size();
    for (const Mat& p : planes) MI_CV_ASSERT(p.type() == CV_8UC1 && p.rows == planes[0].rows && p.cols == planes[0].cols);
    const std::vector<Mat> keep = planes;                         // dst may alias one of the planes' headers
    dst.create(keep[0].rows, keep[0].cols, makeType(CV_8U, cn));
    for (int y = 0; y < dst.rows; ++y) {
        unsigned char* d = dst.ptr(y);
        for (int k = 0; k < cn; ++k) {
            const unsigned char* s = keep[(size_t)k].ptr(y);
            for (int x = 0; x < dst.cols; ++x) d[(size_t)x * cn + k] = s[x];
        }
    }
}

// One call for the whole sequence BGR2YUV -> split -> equalizeHist / CLAHE on Y -> merge -> YUV2BGR
// (singlecolor.cpp:39-66; clahe1frame.cpp:83-102), entirely on the GPU.
inline void equalizeHistLumaBGR(const Mat& bgr, Mat& dst)
{
    MI_CV_ASSERT(bgr.type() == CV_8UC3);
    if (bgr.empty()) return;
    const Mat s = bgr;
    dst.create(s.rows, s.cols, CV_8UC3);
    mi_ctx* c = detail::thread_ctx();
    detail::check(c, mi_bgr_luma_op_u8c3(c, s.data, s.step, dst.data, dst.step, s.cols, s.rows, MI_OP_EQUALIZE, 0.0, 1, 1), "mi_bgr_luma_op_u8c3");
}
inline void claheLumaBGR(const Mat& bgr, Mat& dst, double clipLimit, Size tiles)
{
    MI_CV_ASSERT(bgr.type() == CV_8UC3);
    if (bgr.empty()) return;
    const Mat s = bgr;
    dst.create(s.rows, s.cols, CV_8UC3);
    mi_ctx* c = detail::thread_ctx();
    detail::check(c, mi_bgr_luma_op_u8c3(c, s.data, s.step, dst.data, dst.step, s.cols, s.rows, MI_OP_CLAHE, clipLimit, tiles.width, tiles.height), "mi_bgr_luma_op_u8c3");
}

// ---- the reference's own device-vs-CPU check, 1frameMeasure.cpp:91-94, call for call:
//          cv::absdiff(y_ocv, y_fpga, diff);   float err_per;   xf::cv::analyzeDiff(diff, 1, err_per);
// absdiff: dst = |a - b| (CV_8UC1, same size).  analyzeDiff: err_per = percentage of pixels of `diff` that EXCEED err_thresh; the
// smallest / largest difference are returned through the optional pointers (Vitis Vision prints them).
inline void absdiff(const Mat& a, const Mat& b, Mat& dst)
{
    MI_CV_ASSERT(a.type() == CV_8UC1 && b.type() == CV_8UC1);
    MI_CV_ASSERT(a.rows == b.rows && a.cols == b.cols);
    if (a.empty()) return;
    const Mat x = a, y = b;
    dst.create(x.rows, x.cols, CV_8UC1);
    mi_ctx* c = detail::thread_ctx();
    mi_diff_stats st{};
    detail::check(c, mi_analyze_diff_u8(c, x.data, x.step, y.data, y.step, dst.data, dst.step, x.cols, x.rows, 0, &st), "mi_analyze_diff_u8");
}
inline void analyzeDiff(const Mat& diff, int err_thresh, float& err_per, int* min_diff = nullptr, int* max_diff = nullptr)
{
    MI_CV_ASSERT(diff.type() == CV_8UC1);
    err_per = 0.f;
    if (diff.empty()) return;
    mi_ctx* c = detail::thread_ctx();
    mi_diff_stats st{};
    detail::check(c, mi_analyze_diff_u8(c, diff.data, diff.step, nullptr, 0, nullptr, 0, diff.cols, diff.rows, err_thresh, &st), "mi_analyze_diff_u8");
    err_per = 100.f * (float)st.above / (float)st.total;
    if (min_diff) *min_diff = (int)st.min_diff;
    if (max_diff) *max_diff = (int)st.max_diff;
}

// Pin a recycled frame-buffer pool once so the host forms DMA straight from / into it (mi_host_register).
inline void registerHostBuffer(void* ptr, size_t bytes)
{
    const mi_status st = mi_host_register(ptr, bytes);
    if (st != MI_OK)
        MI_CV_ERROR(st == MI_ERR_NO_DEVICE ? GpuNotSupported : GpuApiCallError,
                    std::string("mi_host_register: ") + mi_status_str(st) + (st == MI_ERR_NO_DEVICE ? " (no HIP device: this backend has no CPU fallback)" : ""));
}
// Unpin it again BEFORE its memory is released.  Two forms:
//   tryUnregisterHostBuffer  never throws (destructors, clean-up paths): MI_OK = unpinned; MI_ERR_BUSY = a pipe still has a transfer queued on
//                            it, the buffer STAYS registered (wait for the pending frames or destroy the pool first, then ask again);
//                            MI_ERR_BAD_ARG = not the start of a registered range; MI_ERR_HIP = the runtime refused, still registered.
//   unregisterHostBuffer     returns false on the busy case and THROWS on every other failure -- not for destructors.
inline mi_status tryUnregisterHostBuffer(void* ptr) noexcept { return mi_host_unregister(ptr); }
inline bool unregisterHostBuffer(void* ptr)
{
    const mi_status st = tryUnregisterHostBuffer(ptr);
    if (st == MI_ERR_BUSY) return false;
    if (st != MI_OK) MI_CV_ERROR(st == MI_ERR_BAD_ARG ? StsBadArg : GpuApiCallError, std::string("mi_host_unregister: ") + mi_status_str(st));
    return true;
}

// ---- whole NV12 frame helpers (what every caller of the reference does around the call) ----
enum UVMode { UV_FILL128 = MI_UV_FILL128, UV_COPY = MI_UV_COPY };

// in/out: tightly packed NV12 (W*H + W*H/2 bytes).  Replaces the clone + equalizeHist + memcpy(Y) +
// memset/memcpy(UV) sequence of OpenCVequalHist.cpp:140-162 / ColoropenCVCwqualHist.cpp:146-165.
inline void equalizeHistNV12(const unsigned char* in, unsigned char* out, int width, int height, UVMode uv)
{
    mi_ctx* c = detail::thread_ctx();
    detail::check(c, mi_equalize_hist_nv12(c, in, out, width, height, (mi_uv_mode)uv), "mi_equalize_hist_nv12");
}
inline void claheNV12(const unsigned char* in, unsigned char* out, int width, int height, UVMode uv,
                      double clipLimit, Size tiles)
{
    mi_ctx* c = detail::thread_ctx();
    detail::check(c, mi_clahe_nv12(c, in, out, width, height, (mi_uv_mode)uv, clipLimit, tiles.width, tiles.height), "mi_clahe_nv12");
}

// BASELINE.json config 5 read literally: cvtColor(COLOR_YUV2BGR_NV12) -> split -> equalizeHist on B, G and R -> merge ->
// cvtColor(COLOR_BGR2YUV_I420) + U/V interleave, NV12 in -> NV12 out in one call (no file of the reference does this;
// ColoropenCVCwqualHist.cpp itself is equalizeHistNV12(..., UV_COPY)).  Width and height must be even.
inline void equalizeHistChannelsNV12(const unsigned char* in, unsigned char* out, int width, int height)
{
    mi_ctx* c = detail::thread_ctx();
    detail::check(c, mi_nv12_bgr_equalize(c, in, out, width, height), "mi_nv12_bgr_equalize");
}

}  // namespace micv

// ---- real OpenCV front end (only when the including program already uses OpenCV) ----
#if defined(MI_CV_WITH_OPENCV) && defined(__has_include)
#if __has_include(<opencv2/core.hpp>) && __has_include(<opencv2/imgproc.hpp>)
#include <opencv2/core.hpp>
#include <opencv2/imgproc.hpp>      // cv::CLAHE
#define MI_CV_HAVE_OPENCV_FRONT_END 1
namespace mi_cv {
inline void throw_cv(mi_ctx* c, mi_status st, const char* what)
{
    if (st == MI_OK) return;
    cv::error(st == MI_ERR_BAD_ARG ? cv::Error::StsBadArg : (st == MI_ERR_UNSUPPORTED ? cv::Error::StsAssert : cv::Error::GpuApiCallError),
              std::string(what) + ": " + mi_status_str(st) + " (" + mi_ctx_last_error_msg(c) + ")", what, __FILE__, __LINE__);
}
// drop-in for cv::equalizeHist(InputArray, OutputArray) on host Mats
inline void equalizeHist(cv::InputArray _src, cv::OutputArray _dst)
{
    CV_Assert(_src.type() == CV_8UC1);
    if (_src.empty()) return;
    cv::Mat src = _src.getMat();
    _dst.create(src.size(), src.type());          // no reallocation when dst already matches (nextimprovement.cpp:164-168)
    cv::Mat dst = _dst.getMat();
    mi_ctx* c = micv::detail::thread_ctx();
    throw_cv(c, mi_equalize_hist_u8(c, src.data, src.step, dst.data, dst.step, src.cols, src.rows), "mi_equalize_hist_u8");
}
class CLAHE_MI final : public cv::CLAHE {
public:
    CLAHE_MI(double clip, cv::Size t) : clip_(clip), tiles_(t) {}
    void apply(cv::InputArray _src, cv::OutputArray _dst) CV_OVERRIDE
    {
        CV_Assert(_src.type() == CV_8UC1 || _src.type() == CV_16UC1);       // clahe.cpp accepts exactly these two
        if (_src.empty()) return;
        cv::Mat src = _src.getMat();
        _dst.create(src.size(), src.type());
        cv::Mat dst = _dst.getMat();
        mi_ctx* c = micv::detail::thread_ctx();
        if (src.type() == CV_8UC1)
            throw_cv(c, mi_clahe_u8(c, src.data, src.step, dst.data, dst.step, src.cols, src.rows, clip_, tiles_.width, tiles_.height), "mi_clahe_u8");
        else
            throw_cv(c, mi_clahe_u16(c, src.ptr<uint16_t>(), src.step, dst.ptr<uint16_t>(), dst.step, src.cols, src.rows, clip_, tiles_.width, tiles_.height), "mi_clahe_u16");
    }
    void setClipLimit(double v) CV_OVERRIDE { clip_ = v; }
    double getClipLimit() const CV_OVERRIDE { return clip_; }
    void setTilesGridSize(cv::Size s) CV_OVERRIDE { tiles_ = s; }
    cv::Size getTilesGridSize() const CV_OVERRIDE { return tiles_; }
    void collectGarbage() CV_OVERRIDE {}
private:
    double clip_;
    cv::Size tiles_;
};
inline cv::Ptr<cv::CLAHE> createCLAHE(double clipLimit = 40.0, cv::Size tileGridSize = cv::Size(8, 8))
{
    return cv::makePtr<CLAHE_MI>(clipLimit, tileGridSize);
}
}  // namespace mi_cv
#endif
#endif

#endif  // MI_CV_HPP_
