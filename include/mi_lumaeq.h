/*
 * mi_lumaeq.h -- C ABI of the MI355X (gfx950) luma histogram-equalization library.
 *
 * This is the drop-in boundary for the ONE hot path of kimkimhun3/OpenCV-OpenCL: the call the
 * reference makes on the CV_8UC1 Y plane of every NV12 frame,
 *
 *     cv::equalizeHist(src, dst)                    OpenCVequalHist.cpp:145, nextimprovement.cpp:168,
 *                                                   AirplanMP4.cpp:90, 1frameMeasure.cpp:44
 *     cv::createCLAHE(clip, Size(t,t))->apply(..)   clahevideo.cpp:184-195/:497, clahe1frame.cpp:88-93,
 *                                                   CLAHECompare.cpp:144-150
 *
 * and for the accelerator backend the reference wires behind that call,
 *
 *     cl::Kernel "equalizeHist_accel"(in, ref, out, rows, cols)
 *                                                   OpenCLequalHist.cpp:346-365 (host sequence),
 *                                                   accel.cpp:36-61 (device kernel), 1frameMeasure.cpp:60-87
 *
 * Plain C: pointers, sizes, integer status codes.  No exceptions, no STL, no torch / OpenCV types.
 * The C++ adapter that presents the cv::Mat-in / cv::Mat-out surface on top of it is
 * opencv-opencl_amd/cxx/mi_cv.hpp; the binding a maintainer of the reference would add is shown
 * in INTEGRATION.md.
 *
 * Threading: every function is safe to call concurrently on DISTINCT contexts (the reference
 * runs 1..8 worker threads, OpenCVequalHist.cpp:274/:397-402 -> one context per worker).  A
 * context is internally locked, so sharing one between threads is safe but serialises.
 *
 * There is no CPU fallback: every entry point needs a HIP device and returns MI_ERR_NO_DEVICE /
 * MI_ERR_HIP otherwise.
 */
#ifndef MI_LUMAEQ_H_
#define MI_LUMAEQ_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI_LUMAEQ_VERSION_MAJOR 0
#define MI_LUMAEQ_VERSION_MINOR 2

typedef enum mi_status {
    MI_OK = 0,
    MI_ERR_BAD_ARG = 1,          /* null pointer, negative size, step < width, tiles <= 0 ...        */
    MI_ERR_UNSUPPORTED = 2,      /* not CV_8UC1-shaped work (the adapter maps this to cv::Exception) */
    MI_ERR_HIP = 3,              /* a HIP call failed; see mi_ctx_last_hip_error()                   */
    MI_ERR_OOM = 4,              /* host or device allocation failed                                 */
    MI_ERR_NO_DEVICE = 5,        /* no usable HIP device / device index out of range                 */
    MI_ERR_BUSY = 6              /* mi_pipe_submit: `depth` frames are in flight, call mi_pipe_wait; any compute entry point
                                  * while frames are pending in the context's pipe; a second mi_pipe_create on a context  */
} mi_status;

/* UV handling of whole-NV12-frame entry points:
 *   MI_UV_FILL128 : memset(out + W*H, 128, W*H/2)      OpenCVequalHist.cpp:160-162, clahevideo.cpp:200-201
 *   MI_UV_COPY    : memcpy(out + W*H, in + W*H, W*H/2) ColoropenCVCwqualHist.cpp:165, improvement.cpp:163,
 *                                                      nextimprovement.cpp:160 */
typedef enum mi_uv_mode { MI_UV_FILL128 = 0, MI_UV_COPY = 1 } mi_uv_mode;

typedef struct mi_ctx mi_ctx;    /* opaque: device id, streams, pinned staging, device scratch */

/* stream argument value selecting the context's own stream (see the device-resident forms) */
#define MI_STREAM_CTX ((void*)(uintptr_t)-1)

/* ---- context ------------------------------------------------------------------------------
 * Replaces the per-worker OpenCL objects of the reference (context/queue/kernel/3 buffers,
 * OpenCLequalHist.cpp:106-192): scratch is allocated lazily for the largest frame seen and
 * reused ("allocate once per size", OpenCLequalHist.cpp:175-186). */
mi_status   mi_ctx_create(int device, mi_ctx** out);
void        mi_ctx_destroy(mi_ctx* ctx);
int         mi_ctx_device(const mi_ctx* ctx);
int         mi_ctx_last_hip_error(const mi_ctx* ctx);      /* raw hipError_t of the last MI_ERR_HIP */
const char* mi_ctx_last_error_msg(const mi_ctx* ctx);      /* human readable, never NULL           */
const char* mi_status_str(mi_status s);
const char* mi_version(void);
int         mi_device_count(void);                         /* 0 when no HIP device is usable       */

/* ---- placement of a GPU's host worker -----------------------------------------------------------
 * The frame-sharded stream runs one host worker per GPU (reference: the worker pool of OpenCVequalHist.cpp:397-402, which
 * places nothing).  mi_thread_bind_near_device() binds the CALLING THREAD to the CPUs of the NUMA node the device's PCIe
 * root complex hangs off (device -> PCI address -> /sys/bus/pci/devices/<bdf>/numa_node -> that node's cpulist, intersected
 * with the CPUs the process may use; sched_setaffinity in-process).  Call it BEFORE mi_ctx_create / mi_pipe_create on that
 * thread: the pinned staging buffers they allocate are then first touched next to the GPU, and threads the library starts
 * later inherit the binding.  Never fatal: when the platform reports no node (-1) or none of its CPUs is available, the
 * thread stays where it is and `why` says so.  MI_LUMAEQ_NUMA_BIND=0 in the environment turns every call into a no-op. */
typedef struct mi_numa_binding {
    int  node;                   /* NUMA node of the device, -1 unknown                    */
    int  cpus;                   /* CPUs the thread is now bound to, 0 = left where it was */
    char why[192];               /* one line for a banner / log                            */
} mi_numa_binding;
mi_status mi_device_pci_bus_id(int device, char* buf, size_t buf_len);         /* "0000:c1:00.0" */
mi_status mi_thread_bind_near_device(int device, mi_numa_binding* out);        /* out may be NULL */

/* ---- host-pointer forms: the cv::Mat boundary -----------------------------------------------
 * Synchronous: on return dst is fully written in host memory (SURVEY 8b "Semantics to keep").
 * src/dst are CV_8UC1 planes with row pitch `*_step` >= width (ROI views: clahevideo.cpp:179);
 * dst may be the same memory as src (in place).  width==0 or height==0 is a no-op (MI_OK).
 * Replaces  cv::equalizeHist(y_in, y_out)  (OpenCVequalHist.cpp:145) and the whole blocking
 * write/write/task/read sequence of OpenCLequalHist.cpp:356-365.
 * Host memory: planes in PINNED memory (mi_host_register below, or hipHostMalloc / hipHostRegister by the caller) are DMA'd
 * as they are; anything else is packed through the context's own pinned staging buffers by the calling thread and one helper
 * thread of the context -- the library never hands the HIP runtime memory it did not pin itself.
 * Errors: whatever a host-pointer form returns, no copy on src / dst is in flight any more when it returns. */
mi_status mi_equalize_hist_u8(mi_ctx* ctx, const uint8_t* src, size_t src_step,
                              uint8_t* dst, size_t dst_step, int width, int height);

/* Replaces  clahe->apply(y_in, y_out)  with clahe = cv::createCLAHE(clip_limit, Size(tiles_x, tiles_y))
 * (clahevideo.cpp:184-195, clahe1frame.cpp:88-93). */
mi_status mi_clahe_u8(mi_ctx* ctx, const uint8_t* src, size_t src_step,
                      uint8_t* dst, size_t dst_step, int width, int height,
                      double clip_limit, int tiles_x, int tiles_y);

/* Whole tightly packed NV12 frame in host memory: Y op + UV fill/copy in one call, writing
 * straight into the caller's output frame (the zero-copy semantics of nextimprovement.cpp:159-168;
 * removes the Y clone + memcpy + memset of OpenCVequalHist.cpp:141/:160-162).
 * in/out hold width*height + width*height/2 bytes; in == out is allowed. */
mi_status mi_equalize_hist_nv12(mi_ctx* ctx, const uint8_t* in, uint8_t* out,
                                int width, int height, mi_uv_mode uv_mode);
mi_status mi_clahe_nv12(mi_ctx* ctx, const uint8_t* in, uint8_t* out, int width, int height,
                        mi_uv_mode uv_mode, double clip_limit, int tiles_x, int tiles_y);

/* ---- device-resident, batched, stream-ordered forms -------------------------------------------
 * Pointers are device pointers on the context's device.  `stream` is a hipStream_t passed as
 * void*, with HIP's own meaning: NULL is the device's default (null) stream -- what
 * torch.cuda.current_stream().cuda_stream is unless the caller switched streams.  Pass
 * MI_STREAM_CTX to use the context's private non-blocking stream (it does NOT order against the
 * null stream).  Asynchronous: the call returns after enqueueing.  A context owns one set of
 * scratch buffers: calls that share a context must be issued on one stream at a time (or be
 * ordered by the caller); use one context per concurrent stream.
 * Frame f of a batch lives at base + f * frame_stride.  These are what a per-GPU worker of the
 * frame-sharded pipeline (reference analogue: worker pool OpenCVequalHist.cpp:397-402)
 * calls, and what bench.py measures. */
mi_status mi_equalize_hist_u8_batch_dev(mi_ctx* ctx,
                                        const void* d_src, size_t src_step, size_t src_frame_stride,
                                        void* d_dst, size_t dst_step, size_t dst_frame_stride,
                                        int width, int height, int n_frames, void* stream);

mi_status mi_clahe_u8_batch_dev(mi_ctx* ctx,
                                const void* d_src, size_t src_step, size_t src_frame_stride,
                                void* d_dst, size_t dst_step, size_t dst_frame_stride,
                                int width, int height, int n_frames,
                                double clip_limit, int tiles_x, int tiles_y, void* stream);

/* n_frames tightly packed NV12 frames (frame pitch = W*H + W*H/2 bytes), Y op + UV fill/copy
 * fused into the same launches. d_in == d_out is allowed. */
mi_status mi_equalize_hist_nv12_batch_dev(mi_ctx* ctx, const void* d_in, void* d_out,
                                          int width, int height, int n_frames,
                                          mi_uv_mode uv_mode, void* stream);
mi_status mi_clahe_nv12_batch_dev(mi_ctx* ctx, const void* d_in, void* d_out,
                                  int width, int height, int n_frames, mi_uv_mode uv_mode,
                                  double clip_limit, int tiles_x, int tiles_y, void* stream);

/* ---- stage-level device entry points ---------------------------
 * The same kernels the fused forms launch, exposed one stage at a time so each can be checked
 * against the oracle and timed against its own roofline. */

/* A2: d_hist[f][256] (int32) = exact histogram of frame f. */
mi_status mi_hist_u8_batch_dev(mi_ctx* ctx, const void* d_src, size_t src_step, size_t src_frame_stride,
                               int width, int height, int n_frames, void* d_hist, void* stream);
/* A3: d_lut[f][256] (uint8) from d_hist[f][256]; total = width*height pixels per frame. */
mi_status mi_equalize_lut_batch_dev(mi_ctx* ctx, const void* d_hist, int64_t total, int n_frames,
                                    void* d_lut, void* stream);
/* A4: dst = lut_f[src] for every frame (the north-star roofline kernel). */
mi_status mi_lut_apply_u8_batch_dev(mi_ctx* ctx, const void* d_src, size_t src_step, size_t src_frame_stride,
                                    void* d_dst, size_t dst_step, size_t dst_frame_stride,
                                    int width, int height, int n_frames, const void* d_lut, void* stream);
/* A6 steps 1-4: d_luts[f][tiles_y*tiles_x][256] (uint8) per-tile clipped LUTs. */
mi_status mi_clahe_tile_luts_batch_dev(mi_ctx* ctx, const void* d_src, size_t src_step, size_t src_frame_stride,
                                       int width, int height, int n_frames,
                                       double clip_limit, int tiles_x, int tiles_y,
                                       void* d_luts, void* stream);

/* ---- the reference's own parity check as an operator (1frameMeasure.cpp:91-100) ------------------------
 * The only test the reference holds compares its accelerator's plane with cv::equalizeHist's:
 *     cv::absdiff(y_ocv, y_fpga, diff);  xf::cv::analyzeDiff(diff, 1, err_per);      pass iff err_per == 0
 * (Vitis Vision's analyzeDiff walks the difference image, reports the smallest and the largest difference and the
 * percentage of pixels whose difference EXCEEDS the threshold).  Both steps in one pass, on planes wherever they are:
 * per frame f, stats[f] = { pixels with |a - b| > threshold, largest |a - b|, smallest |a - b|, pixels compared };
 * err_per = 100.0 * above / total.  `b` may be NULL: `a` then already is a difference image (analyzeDiff on its own);
 * `diff` may be NULL: no difference image is written (otherwise diff = |a - b|, cv::absdiff; it may alias a or b).
 * This library's own tests demand bit-exactness (max_diff == 0); the operator exists so a caller can keep the
 * reference's +-1 check, and so full-size device batches can be compared without a download. */
typedef struct mi_diff_stats { uint32_t above, max_diff, min_diff, total; } mi_diff_stats;
/* host planes (step >= width), blocking */
mi_status mi_analyze_diff_u8(mi_ctx* ctx, const uint8_t* a, size_t a_step, const uint8_t* b, size_t b_step,
                             uint8_t* diff, size_t diff_step, int width, int height, int threshold, mi_diff_stats* out);
/* device planes, batched, stream-ordered; d_stats = n_frames mi_diff_stats in device memory */
mi_status mi_analyze_diff_u8_batch_dev(mi_ctx* ctx, const void* d_a, size_t a_step, size_t a_frame_stride,
                                       const void* d_b, size_t b_step, size_t b_frame_stride,
                                       void* d_diff, size_t diff_step, size_t diff_frame_stride,
                                       int width, int height, int n_frames, int threshold, mi_diff_stats* d_stats, void* stream);

/* ---- CLAHE on CV_16UC1 (OpenCV surface beyond what the reference uses) ------------------------
 * cv::createCLAHE(clip, Size(tx,ty))->apply on 16-bit single-channel images: 65 536 bins, ushort LUTs.
 * Steps / frame strides in BYTES (>= 2*width).  In place allowed.  Results never depend on the content, speed does: frames
 * that populate at most 4096 values -- 10/12-bit samples in the LOW bits, or in the HIGH bits of the word as P010 / P016
 * video stores them (every value a multiple of 1 << shift) -- take one pass of tile histograms and one interpolation from
 * a single table; wider content is walked in windows of the value range.
 * Speed (never bytes) also depends on the context's HISTORY: tiles that could go with several shifts -- a flat letterbox bar of a
 * P010 frame -- take the shift the frames of the context's PREVIOUS call settled on (a video stream keeps its format), so the first
 * call after a change of sample format may run its frames through the slower per-frame LUT pass once.  Every tile of one
 * LAUNCH sees the same hint (a call on many frames is cut into chunks, one launch sequence per chunk): it is read-only while the
 * chunk's tile kernels run and rolled over by the chunk's first interpolation launch.  The hint affects speed only, never bytes. */
mi_status mi_clahe_u16(mi_ctx* ctx, const uint16_t* src, size_t src_step, uint16_t* dst, size_t dst_step,
                       int width, int height, double clip_limit, int tiles_x, int tiles_y);
mi_status mi_clahe_u16_batch_dev(mi_ctx* ctx, const void* d_src, size_t src_step, size_t src_frame_stride,
                                 void* d_dst, size_t dst_step, size_t dst_frame_stride,
                                 int width, int height, int n_frames,
                                 double clip_limit, int tiles_x, int tiles_y, void* stream);

/* ---- optional: pin caller-owned host buffers ----------------------------------------------------------------
 * Video pipelines recycle a small pool of frame buffers (GstBufferPool; the reference maps such buffers at
 * OpenCVequalHist.cpp:115/:158).  Registering a pool's memory once lets the host-pointer forms DMA straight
 * from / into it instead of staging through the context's pinned buffers (contiguous planes only; anything
 * else still stages).  Process-wide, thread-safe; the memory must stay valid -- and must not be freed -- until
 * mi_host_unregister() has returned MI_OK.
 * mi_host_unregister returns MI_ERR_BUSY, and leaves the buffer registered, while a pipe still has a transfer queued on it
 * (between the mi_pipe_submit that took the frame and the mi_pipe_wait that returns it, or the pipe's destruction): unpinning
 * pages under the copy engine is a GPU access to an ordinary heap address, which ends the process.  The reference's accelerator
 * path has the same window and no guard (OpenCLequalHist.cpp:356-367).  MI_ERR_BAD_ARG: `ptr` is not the start of a registered range.
 * MI_ERR_HIP: the runtime refused to unpin although the pages are still pinned; the buffer is STILL registered -- ask again, do not
 * free it yet.  If the runtime refuses because it no longer knows the pages as pinned (the caller unpinned them itself with
 * hipHostUnregister, a runtime teardown did), the entry is dropped and MI_OK returned: nothing is left to undo.
 * mi_host_unregister may wait for the device; it does not hold up other threads' mi_pipe_submit / host-form calls meanwhile (the
 * range being unpinned is simply not treated as pinned any more).  A second thread unregistering the same buffer (same `ptr`) at that
 * moment gets BUSY; when it asks again after the first thread is through it gets MI_ERR_BAD_ARG, which then means "already
 * unregistered" -- only the thread that received MI_OK may free the memory on the strength of its own call.
 * Memory the caller pinned by other means (hipHostMalloc, hipHostRegister) is recognised as pinned when the whole plane lies in ONE
 * such allocation; releasing it while frames are pending is the caller's responsibility. */
mi_status mi_host_register(void* ptr, size_t bytes);
mi_status mi_host_unregister(void* ptr);

/* ---- asynchronous in-order frame pipeline: the per-GPU worker of the frame-sharded stream ---------------------
 * The reference's worker maps a frame, runs the op, rebuilds the NV12 frame and pushes it downstream, one frame at
 * a time (OpenCVequalHist.cpp:102-196); its accelerator variant blocks on each of write, write, task, read
 * (OpenCLequalHist.cpp:356-365).  A pipe is that worker's device side with up to `depth` frames in flight: the upload
 * of frame k+2, the kernels of frame k+1 and the download of frame k run concurrently (two upload lanes, one compute
 * stream, two download lanes per device, shared by all pipes of the process; both DMA directions busy), so ONE host
 * thread per GPU keeps the link full.  Frames are tightly packed NV12 in host memory
 * (W*H + W*H/2 bytes), caller-owned from mi_pipe_submit until the mi_pipe_wait that returns them; completion is in
 * submission order.  Register recycled frame buffers once with mi_host_register (a GstBufferPool's memory): their
 * copies are then fully asynchronous.  Unpinned (pageable) memory is accepted -- the calling thread copies it into / out of
 * pinned staging buffers of the pipe (in mi_pipe_submit / mi_pipe_wait), the DMA itself stays asynchronous.
 *   op         MI_OP_EQUALIZE (OpenCVequalHist.cpp:145), MI_OP_CLAHE (clahevideo.cpp:195), or MI_OP_CHANNELS
 *              (NV12 -> BGR -> equalizeHist on B, G, R -> NV12: mi_nv12_bgr_equalize; ignores uv_mode)
 *   uv_policy  MI_PIPE_UV_HOST (= AUTO for the Y-only ops): only the Y plane crosses PCIe; the UV half is filled with
 *              128 / copied by the calling thread inside mi_pipe_wait while the engines work (the reference's own
 *              memset / memcpy, OpenCVequalHist.cpp:160-162, ColoropenCVCwqualHist.cpp:165);
 *              MI_PIPE_UV_DEVICE: whole frames cross the bus and the kernels write the UV half (no host CPU work)
 *   depth      frames in flight, 2..16; 0 = by frame size: 3 for frames of 8 MiB and more (4K), 6 below (measured: a thread that
 *              submits and waits on 4K frames is fastest with three in flight, 1080p frames want six; a pool whose worker is fed
 *              by another thread runs 4K best with four -- cxx/mi_pool.hpp passes its own)
 * mi_pipe_submit returns MI_ERR_BUSY when `depth` frames are pending.  mi_pipe_wait blocks for the OLDEST pending
 * frame and returns its tag and output pointer.  A pipe uses its context's scratch and lock: ONE pipe per context (a
 * second mi_pipe_create answers MI_ERR_BUSY), destroy it before the context; the context's other compute entry points may
 * be used while no frame is pending and answer MI_ERR_BUSY otherwise.
 * Errors keep caller and pipe in step: a failed mi_pipe_submit occupies no slot and leaves no copy in flight on in / out;
 * a failed mi_pipe_wait has still retired the oldest frame (its tag is returned, its buffers are idle) -- one mi_pipe_wait
 * call, one frame gone, whatever the status.
 * mi_pipe_destroy with frames still pending (submitted, never waited for): the pipe's streams are drained first, so on return NO
 * transfer touches any `in` / `out` buffer any more and registered buffers may be unregistered -- but the CONTENT of those frames'
 * `out` buffers is UNDEFINED: the work mi_pipe_wait does for a frame never happened.  Concretely: under MI_PIPE_UV_HOST the UV half
 * was not written (stale bytes), a registered `out` may or may not hold the finished Y plane, an unpinned `out` received nothing
 * (its result sits in staging that is freed).  Ownership follows the reference's rule for a buffer that was pushed and then dropped
 * (OpenCVequalHist.cpp:183-187: ownership passes on push, a failed push is unref'd, never read): discard such outputs.  A caller
 * that wants every frame calls mi_pipe_wait until mi_pipe_pending() is 0 before destroying the pipe; micv::FramePool::finish()
 * and nv12_stream do exactly that and never rely on destroy to complete a frame. */
typedef struct mi_pipe mi_pipe;
enum { MI_OP_EQUALIZE = 0, MI_OP_CLAHE = 1, MI_OP_CHANNELS = 2 };
enum { MI_PIPE_UV_AUTO = 0, MI_PIPE_UV_HOST = 1, MI_PIPE_UV_DEVICE = 2 };
typedef struct mi_pipe_config {
    int width, height;
    int op;                      /* MI_OP_EQUALIZE | MI_OP_CLAHE | MI_OP_CHANNELS */
    mi_uv_mode uv_mode;
    double clip_limit;           /* MI_OP_CLAHE */
    int tiles_x, tiles_y;        /* MI_OP_CLAHE */
    int depth;
    int uv_policy;
} mi_pipe_config;
mi_status mi_pipe_create(mi_ctx* ctx, const mi_pipe_config* cfg, mi_pipe** out);
void      mi_pipe_destroy(mi_pipe* pipe);
mi_status mi_pipe_submit(mi_pipe* pipe, const uint8_t* in, uint8_t* out, uint64_t tag);
mi_status mi_pipe_wait(mi_pipe* pipe, uint64_t* tag, uint8_t** out_frame);
int       mi_pipe_pending(const mi_pipe* pipe);
int       mi_pipe_depth(const mi_pipe* pipe);

/* ---- colour-domain neighbours of the path (parity unpinned, see oracle/color_oracle.c) ------
 * CV_8UC3 interleaved images, row pitch >= 3*width.
 * mi_cvt_color_u8c3: cv::cvtColor(src, dst, code) for code = MI_COLOR_BGR2YUV (cv::COLOR_BGR2YUV = 82,
 *   singlecolor.cpp:39, clahe1frame.cpp:83) or MI_COLOR_YUV2BGR (cv::COLOR_YUV2BGR = 84, singlecolor.cpp:66,
 *   clahe1frame.cpp:102).  src == dst allowed.
 * mi_bgr_luma_op_u8c3: the whole image-bench sequence in one call -- cvtColor(BGR2YUV) -> split -> equalizeHist
 *   (op = MI_OP_EQUALIZE, singlecolor.cpp:39-66) or CLAHE (op = MI_OP_CLAHE, clahe1frame.cpp:83-102) on the Y
 *   plane -> merge -> cvtColor(YUV2BGR); split/merge are fused into the conversion kernels. */
enum { MI_COLOR_BGR2YUV = 82, MI_COLOR_YUV2BGR = 84 };
mi_status mi_cvt_color_u8c3(mi_ctx* ctx, const uint8_t* src, size_t src_step, uint8_t* dst, size_t dst_step,
                            int width, int height, int code);
mi_status mi_cvt_color_u8c3_batch_dev(mi_ctx* ctx, const void* d_src, size_t src_step, size_t src_frame_stride,
                                      void* d_dst, size_t dst_step, size_t dst_frame_stride,
                                      int width, int height, int n_frames, int code, void* stream);
mi_status mi_bgr_luma_op_u8c3(mi_ctx* ctx, const uint8_t* src, size_t src_step, uint8_t* dst, size_t dst_step,
                              int width, int height, int op, double clip_limit, int tiles_x, int tiles_y);
mi_status mi_bgr_luma_op_u8c3_batch_dev(mi_ctx* ctx, const void* d_src, size_t src_step, size_t src_frame_stride,
                                        void* d_dst, size_t dst_step, size_t dst_frame_stride,
                                        int width, int height, int n_frames, int op,
                                        double clip_limit, int tiles_x, int tiles_y, void* stream);

/* ---- BASELINE.json config 5 read literally (second half; parity unpinned) --------------------------
 * NV12 frame -> BGR -> cv::equalizeHist on each of B, G and R -> NV12.  No file of the reference does this
 * (ColoropenCVCwqualHist.cpp, which the config names, equalizes Y only and copies UV: :146, :165 -- that behaviour is
 * mi_equalize_hist_nv12(..., MI_UV_COPY)).  The entry points below stand for the OpenCV 4.4 sequence a maintainer
 * would write for the config's wording:
 *     cv::cvtColor(nv12, bgr, cv::COLOR_YUV2BGR_NV12); cv::split; cv::equalizeHist x3; cv::merge;
 *     cv::cvtColor(bgr, i420, cv::COLOR_BGR2YUV_I420); U and V interleaved back into an NV12 chroma plane
 * restated in oracle/color_oracle.c (orc_nv12_bgr_equalize).  Tight NV12 frames (W*H luma bytes, then H/2 rows of W
 * interleaved U,V bytes); width and height must be even (OpenCV asserts the same); in == out allowed. */
mi_status mi_nv12_bgr_equalize(mi_ctx* ctx, const uint8_t* nv12_in, uint8_t* nv12_out, int width, int height);
mi_status mi_nv12_bgr_equalize_batch_dev(mi_ctx* ctx, const void* d_in, size_t in_frame_stride,
                                         void* d_out, size_t out_frame_stride,
                                         int width, int height, int n_frames, void* stream);

/* cv::cvtColor's 4:2:0 codes on their own (same arithmetic as the pipeline above; parity unpinned):
 *   MI_COLOR_BGR2YUV_I420 (cv::COLOR_BGR2YUV_I420 = 128; 1frameMeasure.cpp:32 prepares its bench input with it):
 *     src CV_8UC3 W x H (pitch >= 3W) -> dst CV_8UC1 W x H*3/2 (pitch >= W): Y rows, then the U and the V plane packed
 *     the way OpenCV packs them (as if the W x H*3/2 matrix were tight, then laid out with dst_step);
 *   MI_COLOR_YUV2BGR_NV12 (cv::COLOR_YUV2BGR_NV12 = 93): src CV_8UC1 W x H*3/2 NV12 -> dst CV_8UC3 W x H.
 * width/height are the picture's (even).  The device form wants the planar side tight (step == width). */
enum { MI_COLOR_YUV2BGR_NV12 = 93, MI_COLOR_BGR2YUV_I420 = 128 };
mi_status mi_cvt_color_420_u8(mi_ctx* ctx, const uint8_t* src, size_t src_step, uint8_t* dst, size_t dst_step,
                              int width, int height, int code);
mi_status mi_cvt_color_420_u8_batch_dev(mi_ctx* ctx, const void* d_src, size_t src_step, size_t src_frame_stride,
                                        void* d_dst, size_t dst_step, size_t dst_frame_stride,
                                        int width, int height, int n_frames, int code, void* stream);

/* ---- stream completion, fail-soft behaviour of the fused kernel, statistics -------------------------------
 * The batched equalizeHist forms normally run as ONE fused launch whose workgroups hand data to each
 * other through bounded waits (they need a frame's slices co-resident on the GPU).  If such a wait
 * expires -- another tenant holds the compute units, a queue was preempted for longer than the bound --
 * the launch drains, and the small finish kernel that follows EVERY fused launch on the same stream
 * redoes exactly the parts that were not written, with no inter-workgroup dependency.  The caller's
 * stream therefore always carries correct output, whoever synchronises it and however; the event is
 * only counted: mi_ctx_get_stat("fused_fallbacks" | "fused_frames_repaired" | "fused_hard_errors" |
 * "fused_last_status").  (The reference's accelerator path ignores device errors altogether:
 * OpenCLequalHist.cpp:367 catches a type nobody throws.)
 * mi_ctx_synchronize() waits for `stream`; it returns MI_ERR_HIP only if a frame's state contradicted the
 * protocol and the repair refused to guess ("fused_hard_errors"; the host-pointer forms check the same).
 * hipGraph: a batched call issued inside a stream capture is recorded as it is -- all per-launch state of
 * the fused path lives in device memory -- and the graph can be replayed.  Size the scratch with one
 * eager call of the same shape first: scratch growth inside a capture returns MI_ERR_UNSUPPORTED, and
 * once a context has seen a capture it never frees scratch a graph node may reference.
 * A context that keeps being repaired gives the fused path up for a while ("demotion"): `fused_demote_after` repaired launches
 * within 32 fused launches route the following calls through the three-kernel path (no inter-workgroup dependency, nothing to
 * stall) for `fused_reprobe_ms`; then one fused launch probes again, and a repair during the probe doubles the period (up to
 * 64x).  mi_ctx_get_stat("fused_demotions" | "fused_demoted").  The bytes are the same on either path.
 * Options that change BEHAVIOUR (mi_ctx_set_option; the speed-only ones are in mi_lumaeq_tuning.h):
 *   "fused"              1/0, default 1: single-read fused kernel vs the three-kernel path for the batched equalizeHist forms
 *   "fused_timeout_ms"   bound of every inter-workgroup wait of the fused kernel, default 50
 *   "fused_demote_after" 0..32, default 3: repaired launches per window that demote the fused path (0 = never demote)
 *   "fused_reprobe_ms"   first demotion period in milliseconds, default 1000
 *   "clahe_fp_contract"  1/0, default 0: CLAHE interpolation arithmetic.  0 = every multiply and add rounded separately, what an
 *                        x86-64 baseline build of OpenCV computes; 1 = the fused multiply-adds GCC forms from clahe.cpp's
 *                        expressions on FMA targets under its default -ffp-contract=fast, i.e. OpenCV on aarch64 -- the
 *                        reference's own board: txf = fma(x, 1/tw, -0.5), res = fma(fma(l11, xa1, l12*xa), ya1,
 *                        fma(l21, xa1, l22*xa) * ya).  The two differ by 1 in about 0.03 % of the pixels.
 * Other statistics (mi_ctx_get_stat): "error_drains" (error returns that had to wait for a stream first),
 * "host_copies_shared" (staging copies of the host forms the context's helper thread took half of), "host_planes_staged" /
 * "host_planes_direct" (host planes -- inputs and outputs of the host forms and of pipe frames -- packed through the library's
 * pinned staging / DMA'd as the caller pinned them: the library never gives the runtime memory it did not find pinned),
 * "clahe16_mid_launches" (mi_clahe_u16* calls that launched the 16384-entry interpolation kernel for 14-bit content; see
 * MI_OPT_CLAHE16_WIDE in mi_lumaeq_tuning.h). */
mi_status mi_ctx_synchronize(mi_ctx* ctx, void* stream);
mi_status mi_ctx_set_option(mi_ctx* ctx, const char* name, int value);
mi_status mi_ctx_get_stat(mi_ctx* ctx, const char* name, uint64_t* out);

/* ---- timing of the library's own kernels -------------------------------------------------------
 * With profiling on, every kernel the library launches carries a start and a stop hipEvent stamped by
 * its own dispatch on the stream it is launched on (hipExtLaunchKernelGGL; the reference reads its kernel's
 * CL_PROFILING_COMMAND_START/END the same way, 1frameMeasure.cpp:77-85).  mi_ctx_profile_read() synchronises those events and accumulates. */
enum { MI_K_HIST = 0, MI_K_EQ_LUT = 1, MI_K_LUT_APPLY = 2, MI_K_TILE_HIST = 3, MI_K_TILE_LUT = 4,
       MI_K_CLAHE_INTERP = 5, MI_K_FUSED = 6, MI_K_COLOR = 7, MI_K_FUSED_FINISH = 8, MI_K_DIFF = 9, MI_K_COUNT = 10 };
typedef struct mi_profile {
    double   total_ms[MI_K_COUNT];   /* summed kernel durations since the last reset */
    uint64_t launches[MI_K_COUNT];
    double   min_ms[MI_K_COUNT], p10_ms[MI_K_COUNT], p50_ms[MI_K_COUNT], p90_ms[MI_K_COUNT], max_ms[MI_K_COUNT];
                                     /* distribution of the per-launch durations since the last reset (over at most the
                                      * 65 536 most recent launches of each kernel; 0 when there were none) */
} mi_profile;
mi_status   mi_ctx_set_profiling(mi_ctx* ctx, int enabled);   /* 0 off; 1 every kernel (what bench.py's timed region uses: its line
                                                                * lists the finish launches too); 2 every kernel except the
                                                                * few-microsecond housekeeping launch behind each fused kernel (two
                                                                * events fewer per batch) */
mi_status   mi_ctx_profile_read(mi_ctx* ctx, mi_profile* out, int reset);
const char* mi_kernel_name(int k);

#ifdef __cplusplus
}
#endif
#endif /* MI_LUMAEQ_H_ */
