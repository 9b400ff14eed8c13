// test_adapter_opencv.cpp -- FIRST CONTACT WITH A REAL OpenCV, self-verifying (parity is unpinned until this has run).
//
// Real cv::Mat through the front end INTEGRATION.md section 2 hands a maintainer -- mi_cv::equalizeHist(cv::InputArray, cv::OutputArray),
// mi_cv::createCLAHE(...)->apply -- against cv::equalizeHist / cv::createCLAHE of the OpenCV this program is linked with, on the same
// frames, byte for byte:
//   * the five BASELINE.json configurations (1080p and 4K NV12 luma equalizeHist, 4K CLAHE 8x8 clip 2.0, frames of the 4K stream,
//     config 5 in both readings: Y equalize + UV passthrough, and NV12 -> BGR -> equalizeHist on B, G, R -> NV12);
//   * 1919 x 1079 (hun.png's shape: the REFLECT_101 pad quirk of CLAHE);
//   * every known answer of tests/golden/kat.json, through OpenCV AND through this library: the first pins SURVEY Appendix A itself;
//   * a ROI view with step > width                                     (reference: clahevideo.cpp:179);
//   * a caller-owned dst that must be written in place, never reallocated   (nextimprovement.cpp:164-168);
//   * a CV_8UC3 input, which must throw something `catch (const std::exception&)` catches   (OpenCVequalHist.cpp:189);
//   * CV_16UC1 CLAHE on 12-bit, 14-bit and full-range content.
// CLAHE's float steps depend on how the OpenCV at hand was compiled (separately rounded multiply / add on x86-64 baseline builds, GCC's
// FMA contraction on aarch64 -- SURVEY App. A): the program finds out which of the library's two arithmetic modes matches and says so.
//
//   test_adapter_opencv <pin.json> <kat.txt>      exit 0 iff every check passed; the JSON record is written either way
// kat.txt: one known answer per line, written from kat.json by tests/test_opencv_pin.py:
//   <id> <equalize|clahe> <rows> <cols> <clip> <tiles_x> <tiles_y> <src bytes ...> | <dst bytes ...>
//
// Built by `make -C tests/cxx opencv` when `pkg-config --exists opencv4` (this image and the GPU boxes of this pool have no OpenCV:
// there, tests/test_opencv_pin.py compiles this file against the declaration-only headers in tests/cxx/opencv_decl -- a syntax
// check that pins nothing -- and the GPU test skips).  Test code; needs a GPU; does NOT link the oracle: OpenCV is the checker here.
#define MI_CV_WITH_OPENCV
#include "../../opencv-opencl_amd/cxx/mi_cv.hpp"
#ifndef MI_CV_HAVE_OPENCV_FRONT_END
#error "test_adapter_opencv.cpp needs <opencv2/core.hpp> and <opencv2/imgproc.hpp> on the include path (pkg-config --cflags opencv4)"
#endif

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

namespace {

struct Check { std::string name; bool ok; std::string note; };
std::vector<Check> g_checks;
void record(const std::string& name, bool ok, const std::string& note = "")
{
    g_checks.push_back({name, ok, note});
    printf("%s  %s%s%s\n", ok ? "ok  " : "FAIL", name.c_str(), note.empty() ? "" : "  -- ", note.c_str());
    fflush(stdout);
}

// SURVEY 8d's generator, simplified: xorshift bytes shaped like the D1 (uniform) and D2 (natural low-contrast) distributions
void fill_plane(cv::Mat& m, uint64_t seed, int kind)
{
    uint64_t s = seed * 0x9E3779B97F4A7C15ull + 0x5EED0000ull;
    for (int y = 0; y < m.rows; ++y) {
        unsigned char* p = m.ptr<unsigned char>(y);
        for (int x = 0; x < m.cols; ++x) {
            s ^= s << 13; s ^= s >> 7; s ^= s << 17;
            const unsigned r = (unsigned)(s >> 33);
            if (kind == 0) p[x] = (unsigned char)r;
            else { int v = 96 + (x + y) / 16 % 64 + (int)(r % 49) - 24; p[x] = (unsigned char)(v < 16 ? 16 : (v > 200 ? 200 : v)); }
        }
    }
}
void fill_plane16(cv::Mat& m, uint64_t seed, unsigned range)
{
    uint64_t s = seed * 0x9E3779B97F4A7C15ull + 0x5EED0016ull;
    for (int y = 0; y < m.rows; ++y) {
        unsigned short* p = m.ptr<unsigned short>(y);
        for (int x = 0; x < m.cols; ++x) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; p[x] = (unsigned short)((unsigned)(s >> 33) % range); }
    }
}
bool same(const cv::Mat& a, const cv::Mat& b)
{
    if (a.rows != b.rows || a.cols != b.cols || a.type() != b.type()) return false;
    const size_t row = (size_t)a.cols * a.elemSize();
    for (int y = 0; y < a.rows; ++y) if (memcmp(a.ptr<unsigned char>(y), b.ptr<unsigned char>(y), row) != 0) return false;
    return true;
}
void set_fp_contract(int on)
{
    mi_ctx* c = micv::detail::thread_ctx();
    if (mi_ctx_set_option(c, "clahe_fp_contract", on) != MI_OK) { fprintf(stderr, "mi_ctx_set_option(clahe_fp_contract) failed\n"); exit(2); }
}
bool clahe_agrees(const cv::Mat& src, double clip, int tx, int ty)
{
    cv::Mat want, got;
    cv::createCLAHE(clip, cv::Size(tx, ty))->apply(src, want);
    mi_cv::createCLAHE(clip, cv::Size(tx, ty))->apply(src, got);
    return same(want, got);
}
std::string json_escape(const std::string& s)
{
    std::string o;
    for (char ch : s) {
        if (ch == '"' || ch == '\\') { o += '\\'; o += ch; }
        else if (ch == '\n') o += "\\n";
        else if ((unsigned char)ch < 0x20) o += ' ';
        else o += ch;
    }
    return o;
}

}  // namespace

int main(int argc, char** argv)
{
    if (argc < 3) { fprintf(stderr, "usage: test_adapter_opencv <pin.json> <kat.txt>\n"); return 2; }
    const std::string build = cv::getBuildInformation();
    std::string fp_mode = "none";
    try {
        // ---- which arithmetic does this OpenCV's CLAHE use?  (decided on a frame where the two modes differ)
        {
            cv::Mat probe(1080, 1920, CV_8UC1);
            fill_plane(probe, 5, 1);
            set_fp_contract(0);
            if (clahe_agrees(probe, 2.0, 8, 8)) fp_mode = "separately rounded multiply and add (x86-64 baseline)";
            else {
                set_fp_contract(1);
                if (clahe_agrees(probe, 2.0, 8, 8)) fp_mode = "fused multiply-add contraction (aarch64 / -mfma builds)";
                else set_fp_contract(0);
            }
            record("CLAHE arithmetic mode of this OpenCV identified", fp_mode != "none", fp_mode);
        }
        // ---- BASELINE configs 1, 2 (equalizeHist on the Y plane of an NV12 frame, 1080p and 4K), 3 (4K CLAHE 8x8 clip 2.0), 4 (frames of
        // the 4K stream), by distribution
        const int shapes[2][2] = {{1920, 1080}, {3840, 2160}};
        for (const auto& wh : shapes) {
            const int W = wh[0], H = wh[1];
            std::vector<unsigned char> nv12((size_t)W * H * 3 / 2);
            for (int kind = 0; kind < 2; ++kind) {
                cv::Mat frame(H * 3 / 2, W, CV_8UC1, nv12.data());                // OpenCVequalHist.cpp:140
                fill_plane(frame, 100 + (uint64_t)W + (uint64_t)kind, kind);
                cv::Mat y_in = frame(cv::Rect(0, 0, W, H));                        // clahevideo.cpp:179: a VIEW
                cv::Mat want, got;
                cv::equalizeHist(y_in, want);
                mi_cv::equalizeHist(y_in, got);
                record("equalizeHist " + std::to_string(W) + "x" + std::to_string(H) + (kind ? " D2" : " D1") + " (BASELINE configs 1/2/4/5a)", same(want, got));
                record("CLAHE 8x8 clip 2.0 " + std::to_string(W) + "x" + std::to_string(H) + (kind ? " D2" : " D1") + " (BASELINE config 3)", clahe_agrees(y_in, 2.0, 8, 8));
            }
            // config 5, literal reading: NV12 -> BGR -> equalizeHist on B, G and R -> NV12 (cvtColor's 4:2:0 codes either side)
            {
                cv::Mat frame(H * 3 / 2, W, CV_8UC1, nv12.data());
                fill_plane(frame, 555 + (uint64_t)W, 0);
                cv::Mat bgr, i420;
                cv::cvtColor(frame, bgr, cv::COLOR_YUV2BGR_NV12);
                std::vector<cv::Mat> ch;
                cv::split(bgr, ch);
                for (auto& p : ch) cv::equalizeHist(p, p);
                cv::merge(ch, bgr);
                cv::cvtColor(bgr, i420, cv::COLOR_BGR2YUV_I420);
                std::vector<unsigned char> want((size_t)W * H * 3 / 2), got(want.size());
                memcpy(want.data(), i420.ptr<unsigned char>(0), (size_t)W * H);
                const unsigned char* u = i420.ptr<unsigned char>(0) + (size_t)W * H;
                const unsigned char* v = u + (size_t)W * H / 4;
                for (size_t k = 0; k < (size_t)W * H / 4; ++k) { want[(size_t)W * H + 2 * k] = u[k]; want[(size_t)W * H + 2 * k + 1] = v[k]; }
                micv::equalizeHistChannelsNV12(nv12.data(), got.data(), W, H);
                record("NV12 -> BGR -> equalizeHist(B, G, R) -> NV12 " + std::to_string(W) + "x" + std::to_string(H) + " (BASELINE config 5, literal)", want == got);
            }
        }
        // ---- hun.png's shape: both pads applied because one axis is indivisible
        {
            cv::Mat odd(1079, 1919, CV_8UC1);
            fill_plane(odd, 1919, 1);
            record("CLAHE 4x4 clip 3.0 on 1919x1079 (clahe1frame.cpp defaults, REFLECT_101 pad)", clahe_agrees(odd, 3.0, 4, 4));
            record("CLAHE 8x8 clip 2.0 on 1919x1079", clahe_agrees(odd, 2.0, 8, 8));
            cv::Mat want, got;
            cv::equalizeHist(odd, want); mi_cv::equalizeHist(odd, got);
            record("equalizeHist on 1919x1079", same(want, got));
        }
        // ---- ROI view with step > width (clahevideo.cpp:179), unaligned origin
        {
            cv::Mat big(360 + 8, 640 + 64, CV_8UC1);
            fill_plane(big, 179, 1);
            cv::Mat roi = big(cv::Rect(21, 3, 640, 360));
            cv::Mat want, got;
            cv::equalizeHist(roi, want); mi_cv::equalizeHist(roi, got);
            record("equalizeHist on a ROI view (step > width)", !roi.isContinuous() && same(want, got));
            record("CLAHE on a ROI view (step > width)", clahe_agrees(roi, 2.0, 8, 8));
        }
        // ---- caller-owned dst: written in place, never reallocated (nextimprovement.cpp:164-168)
        {
            const int W = 1280, H = 720;
            std::vector<unsigned char> in_buf((size_t)W * H * 3 / 2), out_buf((size_t)W * H * 3 / 2, 0xEE);
            cv::Mat y_in(H, W, CV_8UC1, in_buf.data()), y_out(H, W, CV_8UC1, out_buf.data());
            fill_plane(y_in, 164, 1);
            mi_cv::equalizeHist(y_in, y_out);
            cv::Mat want;
            cv::equalizeHist(y_in, want);
            record("caller-owned dst written in place, not reallocated", y_out.data == out_buf.data() && same(want, y_out) && out_buf[(size_t)W * H] == 0xEE);
            cv::Mat same_buf(H, W, CV_8UC1, in_buf.data());
            mi_cv::equalizeHist(same_buf, same_buf);                               // in place
            record("in place (src == dst)", same(want, same_buf));
        }
        // ---- wrong type: an exception that `catch (const std::exception&)` catches (OpenCVequalHist.cpp:189, clahevideo.cpp:273)
        {
            cv::Mat bgr(48, 64, CV_8UC3), out;
            bool caught = false, caught_clahe = false;
            try { mi_cv::equalizeHist(bgr, out); } catch (const std::exception&) { caught = true; }
            try { mi_cv::createCLAHE(2.0, cv::Size(8, 8))->apply(bgr, out); } catch (const std::exception&) { caught_clahe = true; }
            record("CV_8UC3 input throws a std::exception (equalizeHist)", caught);
            record("CV_8UC3 input throws a std::exception (CLAHE::apply)", caught_clahe);
            cv::Mat empty_src, untouched;
            mi_cv::equalizeHist(empty_src, untouched);
            record("empty src is a no-op", untouched.empty());
        }
        // ---- CLAHE object semantics: setters, reuse over frames of different sizes
        {
            cv::Ptr<cv::CLAHE> ours = mi_cv::createCLAHE(), theirs = cv::createCLAHE();
            bool ok = ours->getClipLimit() == theirs->getClipLimit() && ours->getTilesGridSize().width == theirs->getTilesGridSize().width;
            ours->setClipLimit(3.0); theirs->setClipLimit(3.0);
            ours->setTilesGridSize(cv::Size(4, 6)); theirs->setTilesGridSize(cv::Size(4, 6));
            for (int k = 0; k < 3 && ok; ++k) {
                cv::Mat f(200 + 37 * k, 320 + 51 * k, CV_8UC1), a, b;
                fill_plane(f, 900 + (uint64_t)k, k & 1);
                ours->apply(f, a); theirs->apply(f, b);
                ok = same(a, b);
            }
            ours->collectGarbage();
            record("CLAHE object: defaults, setters, reuse over frames of different sizes", ok);
        }
        // ---- CV_16UC1 CLAHE
        for (unsigned range : {4096u, 16384u, 65536u}) {
            cv::Mat f16(720, 1280, CV_16UC1);
            fill_plane16(f16, range, range);
            record("CV_16UC1 CLAHE 8x8 clip 2.0, values below " + std::to_string(range), clahe_agrees(f16, 2.0, 8, 8));
        }
        // ---- the known answers: OpenCV itself against SURVEY Appendix A's hand-derived values, and this library against both
        {
            std::ifstream kat(argv[2]);
            std::string line;
            int n = 0, opencv_ok = 0, ours_ok = 0;
            std::string opencv_bad, ours_bad;
            while (std::getline(kat, line)) {
                if (line.empty() || line[0] == '#') continue;
                std::istringstream is(line);
                std::string id, op; int rows, cols, tx, ty; double clip;
                is >> id >> op >> rows >> cols >> clip >> tx >> ty;
                std::vector<unsigned char> src, dst;
                std::string tok; bool after = false;
                while (is >> tok) { if (tok == "|") { after = true; continue; } (after ? dst : src).push_back((unsigned char)std::stoi(tok)); }
                if ((int)src.size() != rows * cols || dst.size() != src.size()) { record("kat.txt entry " + id + " is well-formed", false); continue; }
                ++n;
                cv::Mat s(rows, cols, CV_8UC1, src.data()), want(rows, cols, CV_8UC1, dst.data()), a, b;
                if (op == "equalize") { cv::equalizeHist(s, a); mi_cv::equalizeHist(s, b); }
                else { cv::createCLAHE(clip, cv::Size(tx, ty))->apply(s, a); mi_cv::createCLAHE(clip, cv::Size(tx, ty))->apply(s, b); }
                if (same(a, want)) ++opencv_ok; else opencv_bad += " " + id;
                if (same(b, a)) ++ours_ok; else ours_bad += " " + id;
            }
            // (known answers derived for the separately-rounded mode: an FMA build of OpenCV differs on the ones that guard that quirk)
            record("known answers of kat.json: OpenCV itself reproduces them (" + std::to_string(opencv_ok) + " of " + std::to_string(n) + ")",
                   n > 0 && (opencv_ok == n || fp_mode.find("fused") == 0), opencv_bad.empty() ? "" : "OpenCV differs on:" + opencv_bad);
            record("known answers of kat.json: this library equals OpenCV on every input (" + std::to_string(ours_ok) + " of " + std::to_string(n) + ")",
                   n > 0 && ours_ok == n, ours_bad.empty() ? "" : "differs on:" + ours_bad);
        }
    } catch (const std::exception& e) {
        record("no unexpected exception", false, e.what());
    }
    // ---- the record
    int failed = 0;
    for (const Check& c : g_checks) failed += c.ok ? 0 : 1;
    {
        std::ofstream js(argv[1]);
        js << "{\n  \"opencv_version\": \"" << CV_VERSION << "\",\n  \"library\": \"" << json_escape(mi_version()) << "\",\n";
        js << "  \"clahe_arithmetic_mode_matched\": \"" << json_escape(fp_mode) << "\",\n  \"build_information_cpu_lines\": [";
        std::istringstream bi(build);
        std::string ln; bool first = true;
        while (std::getline(bi, ln))
            if (ln.find("CPU/HW features") != std::string::npos || ln.find("Baseline:") != std::string::npos || ln.find("Dispatched code") != std::string::npos
                || ln.find("requested:") != std::string::npos || ln.find("C++ flags (Release)") != std::string::npos || ln.find("Parallel framework") != std::string::npos) {
                js << (first ? "" : ", ") << "\"" << json_escape(ln) << "\""; first = false;
            }
        js << "],\n  \"checks\": [\n";
        for (size_t k = 0; k < g_checks.size(); ++k)
            js << "    {\"name\": \"" << json_escape(g_checks[k].name) << "\", \"pass\": " << (g_checks[k].ok ? "true" : "false")
               << ", \"note\": \"" << json_escape(g_checks[k].note) << "\"}" << (k + 1 < g_checks.size() ? ",\n" : "\n");
        js << "  ],\n  \"failed\": " << failed << "\n}\n";
    }
    printf("%d check(s), %d failed; OpenCV %s; record: %s\n", (int)g_checks.size(), failed, CV_VERSION, argv[1]);
    return failed ? 1 : 0;
}
