"""Mutants of the COLOUR oracle's arithmetic (the weakest-pinned rows) -- TEST INFRASTRUCTURE for
tests/test_color_kill_matrix.py.  Same idea as tests/oracle_mutants.py: a switchable restatement of what oracle/color_oracle.c restates
from OpenCV 4.4's color_yuv.simd.hpp (unmutated = the oracle, bit for bit), each mutant getting ONE thing wrong the way a plausible
re-implementation would.  Reference call sites: cvtColor(COLOR_BGR2YUV) / (COLOR_YUV2BGR) singlecolor.cpp:39/:66, clahe1frame.cpp:83/:102;
COLOR_BGR2YUV_I420 1frameMeasure.cpp:32; the NV12 forms belong to BASELINE config 5 read literally (no reference file does that)."""
from __future__ import annotations

import numpy as np

MUTANTS = {
    # ---- COLOR_BGR2YUV, 8-bit: fixed point, shift 14
    "yuv_no_rounding": "CV_DESCALE without its rounding constant (x >> 14)",
    "yuv_crcb_order": "channels stored Y, Cr, Cb (the YCrCb order) instead of Y, U, V",
    "yuv_input_rgb": "input read as R, G, B instead of B, G, R",
    "yuv_delta_after_saturate": "128 added AFTER saturating the descaled difference",
    "yuv_coeff_r2y_truncated": "R2Y = 4898: 0.299 * 2^14 = 4898.8 truncated instead of rounded",
    "yuv_coeff_b2y_truncated": "B2Y = 1867: 0.114 * 2^14 = 1867.8 truncated",
    "yuv_coeff_b2ui_truncated": "B2UI = 8060: 0.492 * 2^14 = 8060.9 truncated",
    "yuv_coeff_r2vi_truncated": "R2VI = 14368: 0.877 * 2^14 = 14368.8 truncated",
    "yuv_chroma_from_exact_luma": "U and V from the UNROUNDED luma sum (one descale at the end) instead of from the stored Y",
    "yuv_bt709": "BT.709 luma weights (0.2126, 0.7152, 0.0722) instead of BT.601",
    # ---- COLOR_YUV2BGR
    "bgr_no_rounding": "CV_DESCALE without its rounding constant",
    "bgr_green_two_roundings": "green from two separately descaled products instead of one descale of their sum",
    "bgr_uv_swapped": "U and V read in the other order",
    "bgr_coeff_v2ri_truncated": "V2RI = 18677: 1.140 * 2^14 = 18677.8 truncated",
    "bgr_coeff_u2gi_truncated": "U2GI = -6471: -0.395 * 2^14 = -6471.7 truncated toward zero",
    "bgr_output_rgb": "output stored R, G, B",
    # ---- NV12 -> BGR (COLOR_YUV2BGR_NV12), shift 20
    "dec_full_range_luma": "no -16 on Y (full-range decode)",
    "dec_no_luma_clamp": "max(0, Y - 16) left out: Y < 16 gives negative luma",
    "dec_no_rounding": "no 2^19 rounding constant",
    "dec_nv21": "chroma read V, U (NV21)",
    "dec_chroma_of_next_pair": "odd columns take the chroma of the NEXT pair",
    # ---- BGR -> NV12 / I420 (COLOR_BGR2YUV_I420), shift 20
    "enc_chroma_averaged": "chroma from the average of the 2x2 block instead of its top-left pixel",
    "enc_no_rounding": "no 2^19 rounding constant",
    "enc_full_range_luma": "no +16 on Y",
    "enc_uv_swapped": "V stored before U",
    "enc_chroma_bottom_right": "chroma from the bottom-right pixel of the block",
}

S14, H14 = 14, 1 << 13
S20, H20 = 20, 1 << 19


def _sat(x):
    return np.clip(x, 0, 255).astype(np.uint8)


def bgr2yuv(px, mutant=None):
    a = np.asarray(px, np.uint8).astype(np.int64)
    b, g, r = (a[..., 2], a[..., 1], a[..., 0]) if mutant == "yuv_input_rgb" else (a[..., 0], a[..., 1], a[..., 2])
    r2y, g2y, b2y, b2ui, r2vi = 4899, 9617, 1868, 8061, 14369
    r2y -= mutant == "yuv_coeff_r2y_truncated"
    b2y -= mutant == "yuv_coeff_b2y_truncated"
    b2ui -= mutant == "yuv_coeff_b2ui_truncated"
    r2vi -= mutant == "yuv_coeff_r2vi_truncated"
    if mutant == "yuv_bt709":
        r2y, g2y, b2y = 3483, 11718, 1183
    half = 0 if mutant == "yuv_no_rounding" else H14
    ysum = b * b2y + g * g2y + r * r2y
    Y = (ysum + half) >> S14
    if mutant == "yuv_chroma_from_exact_luma":
        U = ((((b << S14) - ysum) * b2ui >> S14) + (128 << S14) + half) >> S14
        V = ((((r << S14) - ysum) * r2vi >> S14) + (128 << S14) + half) >> S14
    elif mutant == "yuv_delta_after_saturate":
        U = np.clip(((b - Y) * b2ui + half) >> S14, 0, 255) + 128
        V = np.clip(((r - Y) * r2vi + half) >> S14, 0, 255) + 128
    else:
        U = ((b - Y) * b2ui + (128 << S14) + half) >> S14
        V = ((r - Y) * r2vi + (128 << S14) + half) >> S14
    out = [Y, V, U] if mutant == "yuv_crcb_order" else [Y, U, V]
    return np.stack([_sat(c) for c in out], -1)


def yuv2bgr(px, mutant=None):
    a = np.asarray(px, np.uint8).astype(np.int64)
    y, u, v = a[..., 0], a[..., 1], a[..., 2]
    if mutant == "bgr_uv_swapped":
        u, v = v, u
    u2bi, u2gi, v2gi, v2ri = 33292, -6472, -9519, 18678
    v2ri -= mutant == "bgr_coeff_v2ri_truncated"
    u2gi += mutant == "bgr_coeff_u2gi_truncated"
    half = 0 if mutant == "bgr_no_rounding" else H14
    B = y + (((u - 128) * u2bi + half) >> S14)
    if mutant == "bgr_green_two_roundings":
        G = y + (((u - 128) * u2gi + half) >> S14) + (((v - 128) * v2gi + half) >> S14)
    else:
        G = y + (((u - 128) * u2gi + (v - 128) * v2gi + half) >> S14)
    R = y + (((v - 128) * v2ri + half) >> S14)
    out = [R, G, B] if mutant == "bgr_output_rgb" else [B, G, R]
    return np.stack([_sat(c) for c in out], -1)


def nv12_to_bgr(nv12, width, height, mutant=None):
    a = np.asarray(nv12, np.uint8).reshape(-1).astype(np.int64)
    Y = a[: width * height].reshape(height, width)
    uv = a[width * height:].reshape(height // 2, width // 2, 2)
    U = np.repeat(np.repeat(uv[..., 0], 2, 0), 2, 1) - 128
    V = np.repeat(np.repeat(uv[..., 1], 2, 0), 2, 1) - 128
    if mutant == "dec_nv21":
        U, V = V, U
    if mutant == "dec_chroma_of_next_pair":
        sh = lambda c: np.concatenate([c[:, 2:], c[:, -2:]], 1)
        Un, Vn = sh(U), sh(V)
        odd = (np.arange(width) & 1).astype(bool)[None, :]
        U, V = np.where(odd, Un, U), np.where(odd, Vn, V)
    half = 0 if mutant == "dec_no_rounding" else H20
    if mutant == "dec_full_range_luma":
        yy = Y * 1220542
    elif mutant == "dec_no_luma_clamp":
        yy = (Y - 16) * 1220542
    else:
        yy = np.maximum(Y - 16, 0) * 1220542
    R = (yy + half + 1673527 * V) >> S20
    G = (yy + half - 852492 * V - 409993 * U) >> S20
    B = (yy + half + 2116026 * U) >> S20
    return np.stack([_sat(B), _sat(G), _sat(R)], -1)


def bgr_to_nv12(bgr, mutant=None):
    a = np.asarray(bgr, np.uint8).astype(np.int64)
    h, w = a.shape[:2]
    B, G, R = a[..., 0], a[..., 1], a[..., 2]
    half = 0 if mutant == "enc_no_rounding" else H20
    yoff = 0 if mutant == "enc_full_range_luma" else (16 << S20)
    Y = _sat((269484 * R + 528482 * G + 102760 * B + half + yoff) >> S20)
    if mutant == "enc_chroma_averaged":
        f = lambda c: (c[0::2, 0::2] + c[0::2, 1::2] + c[1::2, 0::2] + c[1::2, 1::2] + 2) >> 2
        r0, g0, b0 = f(R), f(G), f(B)
    elif mutant == "enc_chroma_bottom_right":
        r0, g0, b0 = R[1::2, 1::2], G[1::2, 1::2], B[1::2, 1::2]
    else:
        r0, g0, b0 = R[0::2, 0::2], G[0::2, 0::2], B[0::2, 0::2]
    U = _sat((-155188 * r0 - 305135 * g0 + 460324 * b0 + half + (128 << S20)) >> S20)
    V = _sat((460324 * r0 - 385875 * g0 - 74448 * b0 + half + (128 << S20)) >> S20)
    uv = np.stack([V, U] if mutant == "enc_uv_swapped" else [U, V], -1)
    return np.concatenate([Y.reshape(-1), uv.reshape(-1)])
