// lumaeq_kernels.hip.h -- hand-written gfx950 (CDNA4, wave64) kernels for the luma-equalization path.
//
// What they compute is fixed by OpenCV 4.4's cv::equalizeHist / CLAHE::apply as the reference calls
// them (OpenCVequalHist.cpp:145, clahevideo.cpp:195); how they compute it is MI355X-first:
//   * everything is byte/LUT work bound by HBM (no MFMA): 16 B/lane coalesced loads and stores,
//     grid sized to ~8 workgroups/CU, a batch of frames per launch (grid.y / grid.z = frame);
//   * histograms are privatised in LDS as hist[bin][32]: the copy a lane uses is (lane & 31), so
//     its LDS bank is fixed by the lane and a ds_add_u32 wave-instruction is bank-conflict free
//     whatever the pixel values are (a constant frame costs the same as noise);
//   * the equalizeHist LUT is replicated the same way (lut[value][32]) so the per-pixel gather is
//     conflict free as well;
//   * float steps (LUT scale, CLAHE blend) must round every multiply and add separately.  What guarantees
//     it is the BUILD FLAG -ffp-contract=off (csrc/Makefile forces it with `override`): ROCm 7.2's
//     __fmul_rn/__fadd_rn are plain `*` / `+` unless OCML_BASIC_ROUNDED_OPERATIONS is defined, so on their
//     own they contract to v_fma under hipcc's default -ffp-contract=fast (checked in the ISA: 125 FMAs in
//     the interpolation kernel), and that mode also disregards `#pragma clang fp contract(off)`.  They are
//     kept as documentation of intent; a library built without the flag refuses to create a context
//     (contract_probe_kernel, run once per process by mi_ctx_create).
// The staged kernels (K1..K6) never synchronise between workgroups inside a launch: stage results
// cross kernel boundaries only (partials -> LUT -> apply).  The fused equalizeHist kernel (KF) is the
// one exception: its workgroups hand a frame's histogram/LUT to each other through a bounded,
// checksum-verified protocol (kernels/equalize_fused.hip.h).
//
//   kernels/common.hip.h          types, batch descriptors, bank-replicated LDS histogram, scans
//   kernels/equalize.hip.h        K1 hist partials, K2 CDF->LUT, K3 LUT apply (+UV)
//   kernels/equalize_fused.hip.h  KF fused single-read equalizeHist
//   kernels/clahe.hip.h           K4 tile hist, K5 clip/redistribute/LUT, K6 interpolation
//   kernels/clahe16.hip.h         CLAHE on CV_16UC1 (N4)
//   kernels/color.hip.h           cvtColor BGR2YUV / YUV2BGR + fused split/merge, 4:2:0 codes, NV12 per-channel equalize (N3)
//   kernels/color_clahe.hip.h     CLAHE on the luma of interleaved BGR in two passes (N3)
//   kernels/diff.hip.h            absdiff + analyzeDiff: the reference's own device-vs-CPU check (1frameMeasure.cpp:91-100)
#pragma once
#include "kernels/common.hip.h"
#include "kernels/equalize.hip.h"
#include "kernels/equalize_fused.hip.h"
#include "kernels/clahe.hip.h"
#include "kernels/clahe16.hip.h"
#include "kernels/color.hip.h"
#include "kernels/color_clahe.hip.h"
#include "kernels/diff.hip.h"
