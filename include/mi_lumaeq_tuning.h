/*
 * mi_lumaeq_tuning.h -- speed-only knobs of libmi_lumaeq (mi_ctx_set_option).
 *
 * Nothing in this file changes a single output byte: every option below only moves work between equivalent kernels or
 * changes a launch geometry.  They exist for the measurements under profiles/ and tools/; a caller of the drop-in boundary
 * (include/mi_lumaeq.h: the replacement of cv::equalizeHist / cv::CLAHE::apply at OpenCVequalHist.cpp:145,
 * clahevideo.cpp:195) never needs them.  Options that change behaviour are documented in mi_lumaeq.h.
 * This header declares no functions.
 */
#ifndef MI_LUMAEQ_TUNING_H_
#define MI_LUMAEQ_TUNING_H_

#include "mi_lumaeq.h"        /* mi_ctx_set_option */

/* fused single-read equalizeHist kernel */
#define MI_OPT_FUSED_WGS_PER_CU    "fused_wgs_per_cu"    /* 1..8, default 4: persistent workgroups per compute unit                   */
#define MI_OPT_FUSED_VPT           "fused_vpt"           /* 8 / 16 / 20 / 24 (0 = default 20): 16-byte vectors a thread keeps in
                                                           * registers between the histogram and the apply pass (slice = 256 x VPT x 16 B) */
#define MI_OPT_FUSED_ACQUIRE       "fused_acquire"       /* 1/0, default 1: agent-scope acquire before a consumer reads the LUT       */

/* equalizeHist on few frames */
#define MI_OPT_TWO_KERNEL_MAX      "two_kernel_max_frames" /* 0..64, default 8: calls of up to this many frames (twice as many when a frame is
                                                           * 1080p-sized or smaller) run as histogram + LUT in one launch (the last workgroup
                                                           * writes the LUT) followed by the apply kernel, instead of the fused pair; 0 = never */

/* CLAHE (8-bit) */
#define MI_OPT_CLAHE_XCD_MAP       "clahe_xcd_map"       /* 1/0, default 1: XCD-aware tile order of the tile-histogram pass           */
#define MI_OPT_CLAHE_HIST_THREADS  "clahe_hist_threads"  /* 256 / 512, default 512: threads per tile-histogram workgroup              */
#define MI_OPT_CLAHE_SEG_PAIRS     "clahe_seg_pairs"     /* 4..15, default 9: LUT pairs per LDS float table when a grid wider than 14
                                                           * tiles is cut into column segments                                      */
#define MI_OPT_CLAHE_TILES_PER_WG  "clahe_tiles_per_wg"  /* 0..8, default 0 = by tile size: tiles one tile-histogram workgroup walks
                                                           * in large batches of small tiles                                         */
#define MI_OPT_CLAHE_FLOAT_TABLES  "clahe_float_tables"  /* 1/0, default 1: f32 pair tables in LDS for the interpolation (<= 14 tiles
                                                           * across)                                                                 */

/* CLAHE (16-bit) */
#define MI_OPT_CLAHE16_FAST12      "clahe16_fast12"      /* 1/0, default 1: tile histograms bet on values < 4096 (4096 bins x 4 LDS copies,
                                                           * LUT folded in); a tile that loses the bet is redone with 16 384 counters   */
#define MI_OPT_CLAHE16_TRANSPOSED  "clahe16_transposed"  /* 1/0, default 0: value-major LUT layout for the 16-bit interpolation       */
#define MI_OPT_CLAHE16_WIDE        "clahe16_wide"        /* 0/1/2, default 1: rectangles whose range needs 8193..16384 table entries (14-bit
                                                           * sensors) are interpolated by a kernel with ONE 128-KiB table window instead of two
                                                           * windows of the 64-KiB table; 1 = launched while such content was seen in the
                                                           * context's last calls, 2 = always, 0 = never                                 */

/* colour neighbours */
#define MI_OPT_BGR_FUSED           "bgr_fused"           /* 1/0, default 1: mi_bgr_luma_op_u8c3 as two passes over the interleaved
                                                           * image instead of through Y/U/V planes                                   */

/* host-pointer forms */
#define MI_OPT_HOST_COPY_THREADS   "host_copy_threads"   /* 1 / 2, default 2: threads that pack unpinned planes through the pinned
                                                           * staging buffers (the caller alone, or the caller and the context's helper) */

#define MI_OPT_HOST_COPY_STREAMS   "host_copy_streams"   /* 1 / 2, default 2: streams the chunk DMAs of a staged plane alternate between (the
                                                           * copy engine idles ~10 us between dependent copies of one stream)          */
#define MI_OPT_PIPE_COPY_STREAMS   "pipe_copy_streams"   /* 1 / 2, default 2: copy streams per direction of a pipe created afterwards:
                                                           * consecutive frames alternate between them                                */
/* The same knobs from the environment, read by mi_ctx_create (for A/B runs of unmodified programs):
 * MI_LUMAEQ_FUSED, MI_LUMAEQ_FUSED_WGS_PER_CU, MI_LUMAEQ_FUSED_VPT, MI_LUMAEQ_FUSED_ACQUIRE, MI_LUMAEQ_HOST_COPY_THREADS,
 * MI_LUMAEQ_HOST_COPY_STREAMS, MI_LUMAEQ_PIPE_COPY_STREAMS. */

#endif /* MI_LUMAEQ_TUNING_H_ */
