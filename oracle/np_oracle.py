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


def np_clahe(src: np.ndarray, clip_limit: float = 40.0, tiles_x: int = 8, tiles_y: int = 8) -> np.ndarray:
    """8-bit (histSize 256) or 16-bit (histSize 65536) CLAHE."""
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
        tf = np.arange(n).astype(F) * inv - F(0.5)
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
    top = (A * xa1[None, :] + B * xa[None, :]) * ya1[:, None]
    bot = (C * xa1[None, :] + D * xa[None, :]) * ya[:, None]
    res = top + bot
    assert res.dtype == np.float32
    return np.clip(np.rint(res), 0, HS - 1).astype(src.dtype)
