"""Second, independent restatement (numpy) of the same OpenCV 4.4 arithmetic -- TEST ONLY.

Written in a different style from ``lumaeq_oracle.c`` (vectorised, padded image materialised with
``np.pad(mode="reflect")`` instead of index reflection, bincount instead of loops) so that a
transcription slip in either one shows up as a disagreement.  PARITY UNPINNED (see
``lumaeq_oracle.c``): both follow SURVEY.md Appendix A, which restates
modules/imgproc/src/histogram.cpp (cv::equalizeHist) and modules/imgproc/src/clahe.cpp
(CLAHE_Impl::apply) of OpenCV 4.4 as called from OpenCVequalHist.cpp:145 and clahevideo.cpp:195.

numpy float32 scalars/arrays round every operation to binary32 (no FMA), np.rint rounds half to
even: the same semantics as the x86-64 baseline build.
"""
from __future__ import annotations

import numpy as np

F = np.float32


def _fma32(a, b, c):
    """Correctly rounded float32 fused multiply-add for float32 arrays, without an fma primitive: the product of two
    binary32 numbers is exact in binary64; the sum is formed in binary64 with ROUND-TO-ODD (the inexact bit is kept as a sticky
    LSB via the TwoSum residual), after which the final rounding to binary32 cannot double-round (53 >= 2*24 + 2)."""
    a = np.asarray(a, np.float32).astype(np.float64)
    b = np.asarray(b, np.float32).astype(np.float64)
    c = np.asarray(c, np.float32).astype(np.float64)
    p = a * b                                            # exact
    s = p + c                                            # round-to-nearest binary64
    bb = s - p
    err = (p - (s - bb)) + (c - bb)                      # exact residual (p + c) - s
    inexact = err != 0
    # truncate toward zero where round-to-nearest went away from zero, then force the last bit to 1
    away = inexact & (np.sign(err) != np.sign(s)) & (s != 0)
    t = np.where(away, np.nextafter(s, 0.0), s)
    bits = t.view(np.int64) if t.ndim else np.array(t).view(np.int64)
    bits = np.where(inexact, bits | 1, bits)
    return bits.astype(np.int64).view(np.float64).astype(np.float32)


def np_equalize_hist(src: np.ndarray) -> np.ndarray:
    src = np.asarray(src)
    assert src.dtype == np.uint8 and src.ndim == 2
    if src.size == 0:
        return src.copy()
    h = np.bincount(src.reshape(-1), minlength=256).astype(np.int64)
    first = int(np.flatnonzero(h)[0])
    total = src.size
    if h[first] == total:
        return np.full(src.shape, first, np.uint8)
    scale = F(255.0) / F(total - int(h[first]))
    csum = np.cumsum(h) - h[first]               # sum over bins first+1 .. j
    csum[:first + 1] = 0
    lut = np.clip(np.rint(csum.astype(F) * scale), 0, 255).astype(np.uint8)
    lut[:first + 1] = 0
    return lut[src]


def np_clahe_geometry(width, height, clip_limit, tiles_x, tiles_y):
    if width % tiles_x == 0 and height % tiles_y == 0:
        ew, eh = width, height
    else:
        ew = width + (tiles_x - width % tiles_x)
        eh = height + (tiles_y - height % tiles_y)
    tw, th = ew // tiles_x, eh // tiles_y
    area = tw * th
    clip = 0
    if clip_limit > 0.0:
        clip = max(int(float(clip_limit) * area / 256), 1)
    return dict(ext_w=ew, ext_h=eh, tile_w=tw, tile_h=th, clip=clip, lut_scale=F(255.0) / F(area))


def _reflect101_index(n_ext: int, n: int) -> np.ndarray:
    """borderInterpolate(p, n, BORDER_REFLECT_101) for p in [0, n_ext)."""
    idx = np.arange(n_ext)
    if n == 1:
        return np.zeros(n_ext, np.int64)
    period = 2 * (n - 1)
    m = idx % period
    return np.where(m < n, m, period - m)


def np_clahe(src: np.ndarray, clip_limit: float = 40.0, tiles_x: int = 8, tiles_y: int = 8, fp_contract: bool = False) -> np.ndarray:
    """8-bit (histSize 256) or 16-bit (histSize 65536) CLAHE.  fp_contract: the fused multiply-adds GCC forms on
    FMA targets (see lumaeq_oracle.c orc_set_fp_contract) instead of separately rounded operations."""
    src = np.asarray(src)
    assert src.dtype in (np.uint8, np.uint16) and src.ndim == 2
    HS = 256 if src.dtype == np.uint8 else 65536
    H, W = src.shape
    if src.size == 0:
        return src.copy()
    g = np_clahe_geometry(W, H, clip_limit, tiles_x, tiles_y)
    if HS != 256:
        area = g["tile_w"] * g["tile_h"]
        g["clip"] = max(int(float(clip_limit) * area / HS), 1) if clip_limit > 0.0 else 0
        g["lut_scale"] = F(HS - 1) / F(area)
    ext = src[_reflect101_index(g["ext_h"], H)][:, _reflect101_index(g["ext_w"], W)]
    tw, th, clip = g["tile_w"], g["tile_h"], g["clip"]
    luts = np.zeros((tiles_y, tiles_x, HS), src.dtype)
    for ty in range(tiles_y):
        for tx in range(tiles_x):
            tile = ext[ty * th:(ty + 1) * th, tx * tw:(tx + 1) * tw]
            h = np.bincount(tile.reshape(-1), minlength=HS).astype(np.int64)
            if clip > 0:
                clipped = int(np.maximum(h - clip, 0).sum())
                h = np.minimum(h, clip)
                batch, resid = divmod(clipped, HS)
                h = h + batch
                if resid:
                    step = max(HS // resid, 1)
                    bins = np.arange(0, HS, step)[:resid]
                    h[bins] += 1
            luts[ty, tx] = np.clip(np.rint(np.cumsum(h).astype(F) * g["lut_scale"]), 0, HS - 1).astype(src.dtype)

    def axis_tables(n, tile, ntiles):
        inv = F(1.0) / F(tile)
        tf = _fma32(np.arange(n).astype(F), inv, F(-0.5)) if fp_contract else np.arange(n).astype(F) * inv - F(0.5)
        t1 = np.floor(tf).astype(np.int64)
        a = tf - t1.astype(F)
        a1 = F(1.0) - a
        return np.maximum(t1, 0), np.minimum(t1 + 1, ntiles - 1), a.astype(F), a1.astype(F)

    tx1, tx2, xa, xa1 = axis_tables(W, tw, tiles_x)
    ty1, ty2, ya, ya1 = axis_tables(H, th, tiles_y)
    v = src.astype(np.int64)
    A = luts[ty1[:, None], tx1[None, :], v].astype(F)
    B = luts[ty1[:, None], tx2[None, :], v].astype(F)
    C = luts[ty2[:, None], tx1[None, :], v].astype(F)
    D = luts[ty2[:, None], tx2[None, :], v].astype(F)
    if fp_contract:                                     # fma(fma(A, xa1, B*xa), ya1, fma(C, xa1, D*xa) * ya)
        shp = A.shape
        bx = lambda v: np.broadcast_to(v, shp)
        top = _fma32(A, bx(xa1[None, :]), B * xa[None, :])
        bot = _fma32(C, bx(xa1[None, :]), D * xa[None, :]) * ya[:, None]
        res = _fma32(top, bx(ya1[:, None]), bot)
    else:
        top = (A * xa1[None, :] + B * xa[None, :]) * ya1[:, None]
        bot = (C * xa1[None, :] + D * xa[None, :]) * ya[:, None]
        res = top + bot
    assert res.dtype == np.float32
    return np.clip(np.rint(res), 0, HS - 1).astype(src.dtype)


def np_nv12_bgr_equalize(nv12: np.ndarray, width: int, height: int) -> np.ndarray:
    """Second restatement of the literal BASELINE config-5 pipeline (see color_oracle.c): COLOR_YUV2BGR_NV12 ->
    equalizeHist on B, G, R -> COLOR_BGR2YUV_I420 with U,V interleaved.  Vectorised, int64 arithmetic."""
    a = np.asarray(nv12, np.uint8).reshape(-1)
    assert width % 2 == 0 and height % 2 == 0 and a.size == width * height * 3 // 2
    if a.size == 0:
        return a.copy()
    Y = a[:width * height].reshape(height, width).astype(np.int64)
    uv = a[width * height:].reshape(height // 2, width // 2, 2).astype(np.int64) - 128
    U = np.repeat(np.repeat(uv[..., 0], 2, axis=0), 2, axis=1)
    V = np.repeat(np.repeat(uv[..., 1], 2, axis=0), 2, axis=1)
    half = 1 << 19
    yy = np.maximum(Y - 16, 0) * 1220542
    sat = lambda x: np.clip(x, 0, 255).astype(np.uint8)
    R = sat((yy + half + 1673527 * V) >> 20)
    G = sat((yy + half - 852492 * V - 409993 * U) >> 20)
    B = sat((yy + half + 2116026 * U) >> 20)
    B, G, R = (np_equalize_hist(c).astype(np.int64) for c in (B, G, R))
    Yo = sat((269484 * R + 528482 * G + 102760 * B + half + (16 << 20)) >> 20)
    r0, g0, b0 = R[::2, ::2], G[::2, ::2], B[::2, ::2]
    Uo = sat((-155188 * r0 - 305135 * g0 + 460324 * b0 + half + (128 << 20)) >> 20)
    Vo = sat((460324 * r0 - 385875 * g0 - 74448 * b0 + half + (128 << 20)) >> 20)
    return np.concatenate([Yo.reshape(-1), np.stack([Uo, Vo], axis=-1).reshape(-1)])


def np_analyze_diff(a: np.ndarray, b: np.ndarray | None = None, threshold: int = 1) -> dict:
    """The reference's own device-vs-CPU check (1frameMeasure.cpp:91-100): cv::absdiff(a, b, diff) followed by Vitis Vision's
    xf::cv::analyzeDiff(diff, threshold, err_per) -- smallest and largest difference, pixels whose difference EXCEEDS the threshold,
    err_per = 100 * count / (rows * cols).  b = None: `a` already is the difference image.  Vitis Vision is not in /root/reference
    (an include, accel.cpp:8, unversioned): this restates its published behaviour."""
    d = a.astype(np.int16) if b is None else np.abs(a.astype(np.int16) - b.astype(np.int16))
    total = int(d.size)
    above = int(np.count_nonzero(d > threshold))
    return {"above": above, "max_diff": int(d.max()) if total else 0, "min_diff": int(d.min()) if total else 0, "total": total,
            "err_per": 100.0 * above / total if total else 0.0, "diff": d.astype(np.uint8)}
