"""Mutants of the oracle's arithmetic -- TEST INFRASTRUCTURE for the kill matrix (tests/test_kat_kill_matrix.py).

The oracle (oracle/lumaeq_oracle.c, oracle/np_oracle.py) restates OpenCV 4.4 and is PARITY UNPINNED: no OpenCV exists here to compare
with.  What pins it are hand-derived known answers (tests/golden/kat.json).  A known-answer set is only worth something if it
DISCRIMINATES: if a restatement that gets one of OpenCV's quirks wrong still reproduces every known answer, the set pins nothing about
that quirk.  This module is a third, switchable restatement of SURVEY.md Appendix A: with `mutant=None` it must equal the oracle bit
for bit (checked by the test), and each named mutant gets exactly ONE quirk wrong, the way a plausible re-implementation would.

The reference's own check (1frameMeasure.cpp:91-100) tolerates +-1 per pixel and would let most of these through.
"""
from __future__ import annotations

import numpy as np

F = np.float32

# name -> what a careless restatement would have done instead (the quirk of App. A it violates)
MUTANTS = {
    # ---- cv::equalizeHist (App. A.1) ----
    "eq_round_half_up": "LUT: floor(x + 0.5) instead of cvRound's round-half-to-even (A.1 step 6)",
    "eq_round_half_away": "LUT: C round() (half away from zero) instead of half-to-even",
    "eq_round_truncate": "LUT: (uchar)(sum * scale) truncation instead of cvRound",
    "eq_scale_total": "scale = 255 / total instead of 255 / (total - hist[first]) (A.1 step 5)",
    "eq_first_bin_counted": "lut[first] = round(hist[first] * scale): the first populated bin is not skipped (A.1 step 6)",
    "eq_scale_double": "scale and product in double, rounded once, instead of float32 division then float32 multiply",
    "eq_no_constant_shortcut": "no constant-image shortcut: a one-valued image goes through scale = 255 / 0 (A.1 step 4)",
    # ---- CLAHE LUT stage (App. A.2 steps 1-3) ----
    "cl_lut_round_half_up": "tile LUT: floor(x + 0.5) instead of round-half-to-even",
    "cl_lut_round_half_away": "tile LUT: half away from zero",
    "cl_pad_indivisible_axis_only": "pad only the axis that is not divisible (A.2 step 1 pads BOTH whenever EITHER is indivisible)",
    "cl_pad_reflect": "BORDER_REFLECT (edge pixel repeated) instead of BORDER_REFLECT_101",
    "cl_pad_to_multiple": "pad by (t - dim % t) % t: nothing added on a divisible axis",
    "cl_clip_float": "clip limit computed in float32 instead of double (A.2 step 2)",
    "cl_clip_round": "clip limit rounded to nearest instead of truncated",
    "cl_clip_no_floor_of_one": "max(clip, 1) left out: a clip limit that truncates to 0 means 'no clipping'",
    "cl_redistribute_until_stable": "excess redistributed repeatedly until no bin exceeds the clip (A.2 step 3 is ONE pass)",
    "cl_residual_first_bins": "residual spread over the first `resid` bins instead of every (256 / resid)-th bin",
    "cl_residual_step_ceil": "residual step rounded up instead of truncated",
    "cl_lut_scale_double": "lutScale in double (cumulative sum times 255.0 / area, rounded once)",
    # ---- CLAHE interpolation (App. A.2 steps 4-5) ----
    "cl_weights_after_clamp": "xa = txf - clamp(tx1): weights computed AFTER the tile index was clamped",
    "cl_coord_fma": "txf = fma(x, inv_tw, -0.5): one rounding instead of two (what -ffp-contract=fast does)",
    "cl_coord_divide": "txf = x / tile_w - 0.5 instead of x * (1 / tile_w) - 0.5",
    "cl_floor_truncates": "(int)txf instead of cvFloor(txf): -0.3 becomes 0, not -1",
    "cl_blend_fma": "blend contracted into fused multiply-adds (GCC on FMA targets)",
    "cl_blend_y_first": "blend associates the other way: (a*ya1 + c*ya)*xa1 + (b*ya1 + d*ya)*xa",
    "cl_blend_double": "blend in double, rounded once at the end",
    "cl_blend_round_half_up": "final saturate_cast rounds half up instead of half to even",
    "cl_tile_size_from_unpadded": "interpolation uses W / tiles for the tile size although the LUTs were built on the padded image",
}


def _fma32(a, b, c):
    """Correctly rounded binary32 fma(a, b, c) via binary64 with round-to-odd (same construction as oracle/np_oracle.py)."""
    a = np.asarray(a, np.float32).astype(np.float64)
    b = np.asarray(b, np.float32).astype(np.float64)
    c = np.asarray(c, np.float32).astype(np.float64)
    a, b, c = np.broadcast_arrays(a, b, c)
    p = a * b
    s = p + c
    bb = s - p
    err = (p - (s - bb)) + (c - bb)
    inexact = err != 0
    away = inexact & (np.sign(err) != np.sign(s)) & (s != 0)
    t = np.where(away, np.nextafter(s, 0.0), s)
    bits = np.ascontiguousarray(t).view(np.int64)
    bits = np.where(inexact, bits | 1, bits)
    return bits.astype(np.int64).view(np.float64).astype(np.float32)


def _round(x, mode):
    """x: float32 (or float64) array of non-negative values."""
    if mode == "even":
        return np.rint(x)
    if mode == "up":
        return np.floor(np.asarray(x, np.float64) + 0.5)
    if mode == "away":
        xd = np.asarray(x, np.float64)
        return np.sign(xd) * np.floor(np.abs(xd) + 0.5)
    if mode == "trunc":
        return np.trunc(x)
    raise ValueError(mode)


def equalize_hist(src, mutant=None):
    src = np.asarray(src)
    assert src.dtype == np.uint8 and src.ndim == 2
    if src.size == 0:
        return src.copy()
    h = np.bincount(src.reshape(-1), minlength=256).astype(np.int64)
    first = int(np.flatnonzero(h)[0])
    total = int(src.size)
    if h[first] == total and mutant != "eq_no_constant_shortcut":
        return np.full(src.shape, first, np.uint8)
    denom = total if mutant == "eq_scale_total" else total - int(h[first])
    csum = np.cumsum(h)
    if mutant != "eq_first_bin_counted":
        csum = csum - h[first]
    csum[:first] = 0
    mode = {"eq_round_half_up": "up", "eq_round_half_away": "away", "eq_round_truncate": "trunc"}.get(mutant, "even")
    with np.errstate(divide="ignore", invalid="ignore"):
        if mutant == "eq_scale_double":
            prod = csum.astype(np.float64) * (255.0 / denom)
        else:
            prod = csum.astype(F) * (F(255.0) / F(denom))
        lut = np.clip(np.nan_to_num(_round(prod, mode), nan=0.0, posinf=255.0), 0, 255).astype(np.uint8)
    if mutant != "eq_first_bin_counted":
        lut[:first + 1] = 0
    return lut[src]


def _border_index(n_ext, n, reflect101=True):
    idx = np.arange(n_ext)
    if n == 1:
        return np.zeros(n_ext, np.int64)
    if reflect101:
        period = 2 * (n - 1)
        m = idx % period
        return np.where(m < n, m, period - m)
    period = 2 * n                                        # BORDER_REFLECT: fedcba|abcdefgh|hgfedcb
    m = idx % period
    return np.where(m < n, m, period - 1 - m)


def clahe_geometry(W, H, clip_limit, tx, ty, mutant=None):
    if W % tx == 0 and H % ty == 0:
        ew, eh = W, H
    elif mutant == "cl_pad_indivisible_axis_only":
        ew = W + (tx - W % tx) if W % tx else W
        eh = H + (ty - H % ty) if H % ty else H
    elif mutant == "cl_pad_to_multiple":
        ew, eh = W + (tx - W % tx) % tx, H + (ty - H % ty) % ty
    else:
        ew, eh = W + (tx - W % tx), H + (ty - H % ty)
    tw, th = ew // tx, eh // ty
    area = tw * th
    clip = 0
    if clip_limit > 0.0:
        if mutant == "cl_clip_float":
            raw = int(F(clip_limit) * F(area) / F(256))
        elif mutant == "cl_clip_round":
            raw = int(np.rint(float(clip_limit) * area / 256))
        else:
            raw = int(float(clip_limit) * area / 256)
        clip = raw if mutant == "cl_clip_no_floor_of_one" else max(raw, 1)
    return ew, eh, tw, th, clip


def clahe(src, clip_limit=40.0, tiles_x=8, tiles_y=8, mutant=None):
    src = np.asarray(src)
    assert src.dtype == np.uint8 and src.ndim == 2
    H, W = src.shape
    if src.size == 0:
        return src.copy()
    ew, eh, tw, th, clip = clahe_geometry(W, H, clip_limit, tiles_x, tiles_y, mutant)
    r101 = mutant != "cl_pad_reflect"
    ext = src[_border_index(eh, H, r101)][:, _border_index(ew, W, r101)]
    area = tw * th
    lut_mode = {"cl_lut_round_half_up": "up", "cl_lut_round_half_away": "away"}.get(mutant, "even")
    luts = np.zeros((tiles_y, tiles_x, 256), np.uint8)
    for ty in range(tiles_y):
        for tx in range(tiles_x):
            tile = ext[ty * th:(ty + 1) * th, tx * tw:(tx + 1) * tw]
            h = np.bincount(tile.reshape(-1), minlength=256).astype(np.int64)
            if clip > 0:
                rounds = 64 if mutant == "cl_redistribute_until_stable" else 1
                for _ in range(rounds):
                    clipped = int(np.maximum(h - clip, 0).sum())
                    if clipped == 0:
                        break
                    h = np.minimum(h, clip)
                    batch, resid = divmod(clipped, 256)
                    h = h + batch
                    if resid:
                        if mutant == "cl_residual_first_bins":
                            h[:resid] += 1
                        else:
                            step = max(-(-256 // resid) if mutant == "cl_residual_step_ceil" else 256 // resid, 1)
                            h[np.arange(0, 256, step)[:resid]] += 1
            cs = np.cumsum(h)
            if mutant == "cl_lut_scale_double":
                prod = cs.astype(np.float64) * (255.0 / area)
            else:
                prod = cs.astype(F) * (F(255.0) / F(area))
            luts[ty, tx] = np.clip(_round(prod, lut_mode), 0, 255).astype(np.uint8)

    def axis(n, tile, ntiles):
        if mutant == "cl_tile_size_from_unpadded":
            tile = max(n // ntiles, 1)
        xs = np.arange(n).astype(F)
        if mutant == "cl_coord_divide":
            tf = xs / F(tile) - F(0.5)
        elif mutant == "cl_coord_fma":
            tf = _fma32(xs, F(1.0) / F(tile), F(-0.5))
        else:
            tf = xs * (F(1.0) / F(tile)) - F(0.5)
        t1 = (np.trunc(tf) if mutant == "cl_floor_truncates" else np.floor(tf)).astype(np.int64)
        t1c, t2c = np.minimum(np.maximum(t1, 0), ntiles - 1), np.minimum(t1 + 1, ntiles - 1)       # (upper clamp of t1: only a mutant's wrong tile size can reach it)
        a = tf - (t1c if mutant == "cl_weights_after_clamp" else t1).astype(F)
        return t1c, t2c, a.astype(F), (F(1.0) - a).astype(F)

    tx1, tx2, xa, xa1 = axis(W, tw, tiles_x)
    ty1, ty2, ya, ya1 = axis(H, th, tiles_y)
    v = src.astype(np.int64)
    A = luts[ty1[:, None], tx1[None, :], v].astype(F)
    B = luts[ty1[:, None], tx2[None, :], v].astype(F)
    C = luts[ty2[:, None], tx1[None, :], v].astype(F)
    D = luts[ty2[:, None], tx2[None, :], v].astype(F)
    X, X1, Y, Y1 = xa[None, :], xa1[None, :], ya[:, None], ya1[:, None]
    if mutant == "cl_blend_fma":
        res = _fma32(_fma32(A, X1, B * X), Y1, _fma32(C, X1, D * X) * Y)
    elif mutant == "cl_blend_y_first":
        res = (A * Y1 + C * Y) * X1 + (B * Y1 + D * Y) * X
    elif mutant == "cl_blend_double":
        d = np.float64
        res = (A.astype(d) * X1.astype(d) + B.astype(d) * X.astype(d)) * Y1.astype(d) + (C.astype(d) * X1.astype(d) + D.astype(d) * X.astype(d)) * Y.astype(d)
    else:
        res = (A * X1 + B * X) * Y1 + (C * X1 + D * X) * Y
    return np.clip(_round(res, "up" if mutant == "cl_blend_round_half_up" else "even"), 0, 255).astype(np.uint8)
