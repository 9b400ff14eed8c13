// test_adapter.cpp -- C++ parity test of the cv::Mat-style adapter (opencv-opencl_amd/cxx/mi_cv.hpp,
// mi_pool.hpp) against the CPU oracle.  Written the way the reference's programs use the call
// (Mat views over NV12 buffers, ROI, preallocated / external dst, try/catch of std::exception).
// Test code: links oracle/build/liblumaeq_oracle.so as the checker.  Needs a GPU.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../../opencv-opencl_amd/cxx/mi_cv.hpp"
#include "../../opencv-opencl_amd/cxx/mi_pool.hpp"

extern "C" {
int orc_equalize_hist_u8(const uint8_t*, size_t, uint8_t*, size_t, int, int);
int orc_clahe_u8(const uint8_t*, size_t, uint8_t*, size_t, int, int, double, int, int);
int orc_nv12_frame(const uint8_t*, uint8_t*, int, int, int, int, double, int, int);
int orc_bgr_luma_op(const uint8_t*, uint8_t*, int, int, int, double, int, int);
int orc_bgr2yuv_u8(const uint8_t*, size_t, uint8_t*, size_t, int, int);
int orc_nv12_bgr_equalize(const uint8_t*, uint8_t*, int, int);
int orc_set_fp_contract(int);
int orc_bgr_to_i420(const uint8_t*, uint8_t*, int, int);
int orc_nv12_to_bgr(const uint8_t*, uint8_t*, int, int);
int orc_clahe_u16(const uint16_t*, size_t, uint16_t*, size_t, int, int, double, int, int);
}

static int failures = 0;
#define EXPECT(c) do { if (!(c)) { printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #c); ++failures; } } while (0)

static void fill(std::vector<uint8_t>& v, uint64_t seed)
{
    uint64_t s = seed * 0x9E3779B97F4A7C15ull + 1;
    for (auto& b : v) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; b = (uint8_t)(96 + (s >> 33) % 80); }
}

int main()
{
    using namespace micv;
    const int W = 640, H = 360;
    const size_t ysz = (size_t)W * H, uvsz = ysz / 2;
    std::vector<uint8_t> nv12(ysz + uvsz), ref(ysz + uvsz);
    fill(nv12, 1);

    // --- OpenCVequalHist.cpp:140-145: Mat over the NV12 buffer, Y ROI clone, preallocated output
    {
        Mat nv12_input(H * 3 / 2, W, CV_8UC1, nv12.data());
        Mat y_plane_in = nv12_input(Rect(0, 0, W, H)).clone();
        Mat y_plane_out(H, W, CV_8UC1);
        const unsigned char* before = y_plane_out.data;
        equalizeHist(y_plane_in, y_plane_out);
        EXPECT(y_plane_out.data == before);                          // no reallocation when dst matches
        orc_equalize_hist_u8(nv12.data(), W, ref.data(), W, W, H);
        EXPECT(memcmp(y_plane_out.data, ref.data(), ysz) == 0);
    }
    // --- nextimprovement.cpp:164-168: views straight over input and OUTPUT buffers (external dst)
    {
        std::vector<uint8_t> outbuf(ysz + uvsz, 0xEE);
        Mat y_in(H, W, CV_8UC1, nv12.data());
        Mat y_out(H, W, CV_8UC1, outbuf.data());
        equalizeHist(y_in, y_out);
        EXPECT(y_out.data == outbuf.data() && !y_out.ownsMemory());
        EXPECT(memcmp(outbuf.data(), ref.data(), ysz) == 0);
        EXPECT(outbuf[ysz] == 0xEE);                                 // nothing written past the Y plane
    }
    // --- dst empty -> created; in place; ROI with step > width
    {
        Mat y_in(H, W, CV_8UC1, nv12.data());
        Mat dst;
        equalizeHist(y_in, dst);
        EXPECT(dst.rows == H && dst.cols == W && dst.ownsMemory() && memcmp(dst.data, ref.data(), ysz) == 0);
        Mat roi = y_in(Rect(17, 9, 301, 200));
        std::vector<uint8_t> r2((size_t)301 * 200);
        orc_equalize_hist_u8(roi.data, roi.step, r2.data(), 301, 301, 200);
        Mat roi_out;
        equalizeHist(roi, roi_out);
        EXPECT(roi_out.isContinuous() && memcmp(roi_out.data, r2.data(), r2.size()) == 0);
        Mat inplace = y_in.clone();
        equalizeHist(inplace, inplace);
        EXPECT(memcmp(inplace.data, ref.data(), ysz) == 0);
        Mat empty;
        Mat untouched(3, 3, CV_8UC1);
        untouched.setTo(5);
        equalizeHist(empty, untouched);                              // empty src: no-op
        EXPECT(untouched.data[4] == 5);
    }
    // --- 1frameMeasure.cpp:38-44, :91-100 written call for call: the CPU result against the device result through
    //     absdiff + analyzeDiff(diff, 1, err_per); here the "CPU" plane is the oracle's, and the bar is 0 differences at all
    {
        Mat y_plane(H, W, CV_8UC1, nv12.data());
        Mat y_ocv(H, W, CV_8UC1, ref.data());                        // orc_equalize_hist_u8 above
        Mat y_dev(H, W, CV_8UC1);
        Mat diff(H, W, CV_8UC1);
        equalizeHist(y_plane, y_dev);
        absdiff(y_ocv, y_dev, diff);
        float err_per = -1.f; int dmin = -1, dmax = -1;
        analyzeDiff(diff, 1, err_per, &dmin, &dmax);
        EXPECT(err_per == 0.0f && dmin == 0 && dmax == 0);
        y_dev.data[5 * y_dev.step + 7] ^= 1;                         // within the reference's tolerance
        y_dev.data[9 * y_dev.step + 11] ^= 0x40;                     // far outside it
        absdiff(y_ocv, y_dev, diff);
        analyzeDiff(diff, 1, err_per, &dmin, &dmax);
        EXPECT(diff.data[5 * diff.step + 7] == 1 && diff.data[9 * diff.step + 11] == 0x40);
        EXPECT(dmin == 0 && dmax == 0x40 && err_per == 100.f * 1.f / (float)(W * H));
        analyzeDiff(diff, 0, err_per);
        EXPECT(err_per == 100.f * 2.f / (float)(W * H));
    }
    // --- wrong type throws something derived from std::exception (OpenCVequalHist.cpp:189)
    {
        bool threw = false;
        try { Mat bgr(4, 4, CV_8UC3), o; equalizeHist(bgr, o); } catch (const std::exception& e) { threw = true; EXPECT(strstr(e.what(), "CV_8UC1") != nullptr); }
        EXPECT(threw);
    }
    // --- clahevideo.cpp:178-195: createCLAHE, setters, apply on an ROI VIEW of the NV12 buffer
    {
        Mat nv12_in(H * 3 / 2, W, CV_8UC1, nv12.data());
        Mat y_in = nv12_in(Rect(0, 0, W, H));
        Mat y_out(H, W, CV_8UC1);
        Ptr<CLAHE> clahe = createCLAHE(2.0, Size(8, 8));
        clahe->setClipLimit(2.0);
        clahe->setTilesGridSize(Size(8, 8));
        EXPECT(clahe->getClipLimit() == 2.0 && clahe->getTilesGridSize() == Size(8, 8));
        clahe->apply(y_in, y_out);
        std::vector<uint8_t> r(ysz);
        orc_clahe_u8(nv12.data(), W, r.data(), W, W, H, 2.0, 8, 8);
        EXPECT(memcmp(y_out.data, r.data(), ysz) == 0);
        // clahe1frame.cpp defaults: clip 3.0, 4x4, odd-sized image (padding path)
        Mat odd = y_in(Rect(0, 0, 639, 359)).clone();
        Ptr<CLAHE> c2 = createCLAHE(3.0, Size(4, 4));
        Mat odd_out(odd.size(), odd.type());
        c2->apply(odd, odd_out);
        std::vector<uint8_t> r3((size_t)639 * 359);
        orc_clahe_u8(odd.data, odd.step, r3.data(), 639, 639, 359, 3.0, 4, 4);
        EXPECT(memcmp(odd_out.data, r3.data(), r3.size()) == 0);
        c2->collectGarbage();
        EXPECT(createCLAHE()->getClipLimit() == 40.0 && createCLAHE()->getTilesGridSize() == Size(8, 8));
    }
    // --- singlecolor.cpp:39-66 written call for call: cvtColor -> split -> equalizeHist(Y) -> merge -> cvtColor,
    //     and the one-call GPU form of the same sequence; clahe1frame.cpp:83-102 likewise
    {
        const int CW = 321, CH = 199;                                  // odd size: CLAHE padding path
        std::vector<uint8_t> bgrbuf((size_t)CW * CH * 3), want((size_t)CW * CH * 3);
        fill(bgrbuf, 7);
        Mat bgr_image(CH, CW, CV_8UC3, bgrbuf.data());
        Mat yuv_image;
        cvtColor(bgr_image, yuv_image, COLOR_BGR2YUV);
        std::vector<uint8_t> yref((size_t)CW * CH * 3);
        orc_bgr2yuv_u8(bgrbuf.data(), (size_t)CW * 3, yref.data(), (size_t)CW * 3, CW, CH);
        EXPECT(memcmp(yuv_image.data, yref.data(), yref.size()) == 0);
        std::vector<Mat> yuv_channels;
        split(yuv_image, yuv_channels);
        Mat y_equalized;
        equalizeHist(yuv_channels[0], y_equalized);
        std::vector<Mat> enhanced = {y_equalized, yuv_channels[1], yuv_channels[2]};
        Mat enhanced_yuv, enhanced_bgr;
        merge(enhanced, enhanced_yuv);
        cvtColor(enhanced_yuv, enhanced_bgr, COLOR_YUV2BGR);
        orc_bgr_luma_op(bgrbuf.data(), want.data(), CW, CH, 0, 0.0, 1, 1);
        EXPECT(enhanced_bgr.type() == CV_8UC3 && memcmp(enhanced_bgr.data, want.data(), want.size()) == 0);
        Mat fused;
        equalizeHistLumaBGR(bgr_image, fused);
        EXPECT(memcmp(fused.data, want.data(), want.size()) == 0);
        Mat fused_clahe;
        claheLumaBGR(bgr_image, fused_clahe, 3.0, Size(4, 4));
        orc_bgr_luma_op(bgrbuf.data(), want.data(), CW, CH, 1, 3.0, 4, 4);
        EXPECT(memcmp(fused_clahe.data, want.data(), want.size()) == 0);
        bool threw = false;
        try { Mat g(4, 4, CV_8UC1), o; cvtColor(g, o, COLOR_BGR2YUV); } catch (const std::exception&) { threw = true; }
        EXPECT(threw);
    }
    // --- the other arithmetic flavour of CLAHE::apply (GCC's FMA contraction, as OpenCV is built for the reference's aarch64 board)
    {
        Mat y_in(H, W, CV_8UC1, nv12.data());
        std::vector<uint8_t> r0((size_t)W * H), r1((size_t)W * H);
        orc_clahe_u8(nv12.data(), W, r0.data(), W, W, H, 2.0, 8, 8);
        orc_set_fp_contract(1);
        orc_clahe_u8(nv12.data(), W, r1.data(), W, W, H, 2.0, 8, 8);
        orc_set_fp_contract(0);
        Ptr<CLAHE> cl = createCLAHE(2.0, Size(8, 8));
        Mat o0, o1;
        cl->apply(y_in, o0);
        setOption("clahe_fp_contract", 1);
        cl->apply(y_in, o1);
        setOption("clahe_fp_contract", 0);
        EXPECT(memcmp(o0.data, r0.data(), r0.size()) == 0);
        EXPECT(memcmp(o1.data, r1.data(), r1.size()) == 0);
        bool threw = false;
        try { setOption("no_such_option", 1); } catch (const std::exception&) { threw = true; }
        EXPECT(threw);
    }
    // --- BASELINE config 5 read literally: NV12 -> BGR -> equalizeHist on B, G, R -> NV12 in one call
    {
        const int CW = 322, CH = 178;                                    // W % 16 != 0: block-per-lane path
        std::vector<uint8_t> f((size_t)CW * CH * 3 / 2), got(f.size()), want(f.size());
        uint64_t s = 1234;
        for (auto& v : f) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (uint8_t)(60 + (s >> 40) % 120); }
        equalizeHistChannelsNV12(f.data(), got.data(), CW, CH);
        orc_nv12_bgr_equalize(f.data(), want.data(), CW, CH);
        EXPECT(memcmp(got.data(), want.data(), want.size()) == 0);
        bool threw = false;
        try { equalizeHistChannelsNV12(f.data(), got.data(), 321, CH); } catch (const std::exception&) { threw = true; }
        EXPECT(threw);
        // 1frameMeasure.cpp:30-36: cvtColor(bgr, yuv, COLOR_BGR2YUV_I420), then a Y-plane view over yuv.data
        std::vector<uint8_t> bgrv((size_t)CW * CH * 3), i420((size_t)CW * CH * 3 / 2), back((size_t)CW * CH * 3);
        for (auto& v : bgrv) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (uint8_t)(s >> 40); }
        Mat bgr(CH, CW, CV_8UC3, bgrv.data());
        Mat yuv;
        cvtColor(bgr, yuv, COLOR_BGR2YUV_I420);
        orc_bgr_to_i420(bgrv.data(), i420.data(), CW, CH);
        EXPECT(yuv.rows == CH * 3 / 2 && yuv.cols == CW && yuv.type() == CV_8UC1 && memcmp(yuv.data, i420.data(), i420.size()) == 0);
        Mat y_plane(CH, CW, CV_8UC1, yuv.data);
        EXPECT(memcmp(y_plane.data, i420.data(), (size_t)CW * CH) == 0);
        Mat nv(CH * 3 / 2, CW, CV_8UC1, f.data()), bgr2;
        cvtColor(nv, bgr2, COLOR_YUV2BGR_NV12);
        orc_nv12_to_bgr(f.data(), back.data(), CW, CH);
        EXPECT(bgr2.rows == CH && bgr2.type() == CV_8UC3 && memcmp(bgr2.data, back.data(), back.size()) == 0);
    }
    // --- 16-bit CLAHE through the same cv::CLAHE-shaped object
    {
        const int CW = 203, CH = 97;
        std::vector<uint16_t> src16((size_t)CW * CH), ref16((size_t)CW * CH);
        uint64_t s = 99;
        for (auto& v : src16) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (uint16_t)(2000 + (s >> 40) % 3000); }
        Mat m16(CH, CW, CV_16UC1, src16.data());
        Mat out16;
        Ptr<CLAHE> c16 = createCLAHE(2.0, Size(8, 8));
        c16->apply(m16, out16);
        orc_clahe_u16(src16.data(), (size_t)CW * 2, ref16.data(), (size_t)CW * 2, CW, CH, 2.0, 8, 8);
        EXPECT(out16.type() == CV_16UC1 && memcmp(out16.data, ref16.data(), ref16.size() * 2) == 0);
    }
    // --- worker pool: 3 workers, 24 frames, in-order delivery, both UV modes (A7, A8)
    for (int uv = 0; uv < 2; ++uv) {
        const int N = 24;
        std::vector<std::vector<uint8_t>> in(N, std::vector<uint8_t>(ysz + uvsz)), out(N, std::vector<uint8_t>(ysz + uvsz));
        for (int k = 0; k < N; ++k) fill(in[k], 100 + k);
        std::vector<uint64_t> order;
        {
            FramePool pool(3, W, H, FramePool::EQUALIZE, uv ? UV_COPY : UV_FILL128, [&](const FrameJob& j) { order.push_back(j.index); EXPECT(j.ok); });
            for (int k = 0; k < N; ++k) pool.submit(in[k].data(), out[k].data());
            pool.finish();
            EXPECT(pool.stats().frames_out.load() == (uint64_t)N && pool.stats().processing_errors.load() == 0);
        }
        EXPECT((int)order.size() == N);
        for (int k = 0; k < (int)order.size(); ++k) EXPECT(order[k] == (uint64_t)k);
        for (int k = 0; k < N; ++k) {
            orc_nv12_frame(in[k].data(), ref.data(), W, H, uv, 0, 0.0, 0, 0);
            EXPECT(memcmp(out[k].data(), ref.data(), ysz + uvsz) == 0);
        }
    }
    // --- per-frame error handling of the pool (drop-and-count, OpenCVequalHist.cpp:117/:135/:189): a frame that cannot be
    //     processed (null buffer) is reported through the sink with ok == false, counted, and the stream goes on in order
    {
        const int N = 6;
        std::vector<std::vector<uint8_t>> in(N, std::vector<uint8_t>(ysz + uvsz)), out(N, std::vector<uint8_t>(ysz + uvsz));
        for (int k = 0; k < N; ++k) fill(in[k], 300 + k);
        std::vector<int> oks;
        {
            FramePool pool(2, W, H, FramePool::EQUALIZE, UV_FILL128, [&](const FrameJob& j) { oks.push_back(j.ok ? 1 : 0); EXPECT(j.index == oks.size() - 1); });
            for (int k = 0; k < N; ++k) pool.submit(k == 2 ? nullptr : in[k].data(), out[k].data());
            pool.finish();
            EXPECT(pool.stats().processing_errors.load() == 1 && pool.stats().frames_out.load() == (uint64_t)N);
        }
        EXPECT(oks.size() == (size_t)N && oks[2] == 0 && oks[0] == 1 && oks[5] == 1);
        orc_nv12_frame(in[5].data(), ref.data(), W, H, 0, 0, 0.0, 0, 0);
        EXPECT(memcmp(out[5].data(), ref.data(), ysz + uvsz) == 0);
    }
    // --- a sink that runs the adapter ITSELF on the pool thread (the reference's appsink callback chain does more OpenCV work on
    //     delivered frames): the worker's pipe has frames pending at that moment, and a context with pending pipe frames answers
    //     MI_ERR_BUSY -- so the pool worker must own a private context and leave the thread's default one to the sink
    {
        const int N = 10;
        std::vector<std::vector<uint8_t>> in(N, std::vector<uint8_t>(ysz + uvsz)), out(N, std::vector<uint8_t>(ysz + uvsz));
        for (int k = 0; k < N; ++k) fill(in[k], 500 + k);
        int sink_bad = 0, sink_calls = 0;
        std::vector<uint8_t> want_twice(ysz), once(ysz);
        {
            FramePool pool(1, W, H, FramePool::EQUALIZE, UV_FILL128, [&](const FrameJob& j) {
                EXPECT(j.ok);
                try {
                    Mat y(H, W, CV_8UC1, j.out), again;               // equalize the delivered Y plane once more, on this very thread
                    equalizeHist(y, again);
                    orc_equalize_hist_u8(j.out, W, want_twice.data(), W, W, H);
                    if (memcmp(again.data, want_twice.data(), ysz) != 0) ++sink_bad;
                    ++sink_calls;
                } catch (const std::exception& e) { printf("sink: %s\n", e.what()); ++sink_bad; }
            }, 2.0, Size(8, 8), 16, 4);
            for (int k = 0; k < N; ++k) pool.submit(in[k].data(), out[k].data());
            pool.finish();
            EXPECT(pool.stats().processing_errors.load() == 0);
        }
        EXPECT(sink_calls == N && sink_bad == 0);
    }
    // --- the reference's worker pattern without the pool (OpenCVequalHist.cpp:397-402: 1..8 threads, each calling cv::equalizeHist on
    //     its own Mats; clahevideo.cpp: one CLAHE object per thread): equalizeHist is re-entrant across threads, distinct CLAHE objects
    //     are independent.  Eight threads -> eight per-thread contexts on one device (four get the fused kernel, four the three-kernel
    //     path, kMaxFusedCtxPerDevice): every result against the oracle.
    {
        const int T = 8, ITER = 12;
        std::vector<int> bad(T, 0);
        std::vector<std::thread> th;
        for (int w = 0; w < T; ++w)
            th.emplace_back([&, w] {
                try {
                    std::vector<uint8_t> src(ysz), want(ysz), want_c(ysz);
                    Ptr<CLAHE> clahe = createCLAHE(2.0 + w, Size(8, 8));
                    for (int it = 0; it < ITER; ++it) {
                        fill(src, 9000 + 100 * w + it);
                        Mat y_in(H, W, CV_8UC1, src.data()), y_out, y_cl;
                        equalizeHist(y_in, y_out);
                        clahe->apply(y_in, y_cl);
                        orc_equalize_hist_u8(src.data(), W, want.data(), W, W, H);
                        orc_clahe_u8(src.data(), W, want_c.data(), W, W, H, 2.0 + w, 8, 8);
                        if (memcmp(y_out.data, want.data(), ysz) != 0 || memcmp(y_cl.data, want_c.data(), ysz) != 0) ++bad[w];
                    }
                } catch (const std::exception& e) { printf("thread %d: %s\n", w, e.what()); bad[w] = 1000; }
            });
        for (auto& t : th) t.join();
        for (int w = 0; w < T; ++w) EXPECT(bad[w] == 0);
    }
    printf(failures ? "test_adapter: %d FAILURES\n" : "test_adapter: all checks passed\n", failures);
    return failures ? 1 : 0;
}
