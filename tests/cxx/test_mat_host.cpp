// test_mat_host.cpp -- CPU-only checks of the adapter's Mat/Size/Rect/split/merge semantics (no GPU call is made, so
// it runs in the authoring container, under ASan/UBSan).  The GPU-backed calls are covered by test_adapter.cpp.
#include <cstdio>
#include <cstring>
#include <vector>
#include "../../opencv-opencl_amd/cxx/mi_cv.hpp"

static int failures = 0;
#define EXPECT(c) do { if (!(c)) { printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #c); ++failures; } } while (0)

int main()
{
    using namespace micv;
    // create(): allocate once, keep the buffer when size/type already match (Mat::create semantics)
    Mat a(7, 13, CV_8UC1);
    const unsigned char* p = a.data;
    a.create(7, 13, CV_8UC1);
    EXPECT(a.data == p && a.ownsMemory() && a.isContinuous() && a.step == 13 && a.total() == 91);
    a.create(8, 13, CV_8UC1);
    EXPECT(a.rows == 8 && a.data != nullptr);
    // external memory is wrapped, never owned
    std::vector<unsigned char> buf(20 * 30);
    for (size_t i = 0; i < buf.size(); ++i) buf[i] = (unsigned char)(i * 7);
    Mat ext(20, 30, CV_8UC1, buf.data());
    EXPECT(!ext.ownsMemory() && ext.data == buf.data() && ext.step == 30);
    // ROI view shares memory, keeps the parent's step (OpenCVequalHist.cpp:140-141: Mat over NV12, Rect for Y)
    Mat roi = ext(Rect(5, 3, 10, 6));
    EXPECT(roi.rows == 6 && roi.cols == 10 && roi.step == 30 && roi.data == buf.data() + 3 * 30 + 5 && !roi.isContinuous());
    Mat cl = roi.clone();
    EXPECT(cl.ownsMemory() && cl.isContinuous() && cl.step == 10);
    for (int y = 0; y < 6; ++y) EXPECT(memcmp(cl.ptr(y), roi.ptr(y), 10) == 0);
    cl.setTo(9);
    EXPECT(buf[3 * 30 + 5] == (unsigned char)((3 * 30 + 5) * 7));                // clone is independent
    bool threw = false;
    try { Mat bad = ext(Rect(25, 0, 10, 5)); (void)bad; } catch (const std::exception&) { threw = true; }
    EXPECT(threw);
    // empty Mat
    Mat e;
    EXPECT(e.empty() && e.type() == CV_8UC1);
    Mat e2;
    e.copyTo(e2);
    EXPECT(e2.empty());
    // split / merge round trip on CV_8UC3 (singlecolor.cpp:44 / :61)
    Mat c3(5, 4, CV_8UC3);
    for (int y = 0; y < 5; ++y) for (int x = 0; x < 12; ++x) c3.ptr(y)[x] = (unsigned char)(y * 12 + x);
    EXPECT(c3.channels() == 3 && c3.elemSize() == 3 && c3.step == 12);
    std::vector<Mat> planes;
    split(c3, planes);
    EXPECT(planes.size() == 3 && planes[1].type() == CV_8UC1 && planes[1].ptr(2)[3] == (unsigned char)(2 * 12 + 3 * 3 + 1));
    Mat back;
    merge(planes, back);
    EXPECT(back.type() == CV_8UC3 && memcmp(back.data, c3.data, 60) == 0);
    // type checks happen before any device work: wrong types throw std::exception-derived errors without a GPU
    threw = false;
    try { Mat f32(4, 4, CV_32FC1), o; equalizeHist(f32, o); } catch (const Exception& ex) { threw = ex.code == StsAssert; }
    EXPECT(threw);
    threw = false;
    try { Mat g(4, 4, CV_8UC1), o; cvtColor(g, o, COLOR_BGR2YUV); } catch (const std::exception&) { threw = true; }
    EXPECT(threw);
    Mat nothing, out(3, 3, CV_8UC1);
    out.setTo(4);
    equalizeHist(nothing, out);                                                 // empty src: no-op, no device needed
    EXPECT(out.data[0] == 4);
    Ptr<CLAHE> cl2 = createCLAHE();
    EXPECT(cl2->getClipLimit() == 40.0 && cl2->getTilesGridSize() == Size(8, 8));
    cl2->setTilesGridSize(Size(4, 2)); cl2->setClipLimit(3.5);
    EXPECT(cl2->getTilesGridSize() == Size(4, 2) && cl2->getClipLimit() == 3.5);
    printf(failures ? "test_mat_host: %d FAILURES\n" : "test_mat_host: all checks passed\n", failures);
    return failures ? 1 : 0;
}
