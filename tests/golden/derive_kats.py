"""Exact-arithmetic derivation of known answers from SURVEY.md Appendix A -- TEST INFRASTRUCTURE, shares no code with oracle/.

The known answers of tests/golden/kat.json are "hand-derived": worked out from Appendix A, not produced by OpenCV (there is none
here) and not by the oracle they are meant to pin.  For answers that hinge on ONE binary32 rounding (an FMA in the blend, the order
of the bilinear blend, `x * inv - 0.5f` rounded twice) working by hand means carrying 24-bit significands through nine operations;
this file does that bookkeeping with Python's exact rationals: every value is a `Fraction`, every float operation of Appendix A is
written out as ONE exact operation followed by ONE explicit `rn32()` (round to nearest binary32, ties to even), integers stay
integers, and `trace` records each intermediate so the deciding pixel of a KAT can be read step by step (the `why` field of the KAT
quotes it).  No numpy, no float32 type, no vectorisation: nothing here can inherit a mistake from oracle/np_oracle.py or
oracle/lumaeq_oracle.c except a misreading of Appendix A itself.

    python tests/golden/derive_kats.py            prints every derived KAT of kat.json with the trace of its deciding pixel

Doubles appear in exactly one place, as in OpenCV: the clip limit `(int)(clipLimit * area / 256)` (A.2 step 2), evaluated with
Python floats, which ARE IEEE binary64.
"""
from __future__ import annotations

import json
import math
from fractions import Fraction as Q
from pathlib import Path


def rn32(q: Q) -> Q:
    """Round an exact rational to the nearest binary32 (normal range), ties to even; the result is again an exact rational."""
    if q == 0:
        return Q(0)
    sign = -1 if q < 0 else 1
    a = abs(q)
    e = a.numerator.bit_length() - a.denominator.bit_length()
    while Q(2) ** e > a:
        e -= 1
    while Q(2) ** (e + 1) <= a:
        e += 1
    assert -126 <= e <= 127, "outside binary32's normal range"
    ulp = Q(2) ** (e - 23)
    n = a / ulp                                     # in [2^23, 2^24)
    fl = n.numerator // n.denominator
    rem = n - fl
    if rem > Q(1, 2) or (rem == Q(1, 2) and fl % 2 == 1):
        fl += 1
    return sign * fl * ulp


def cv_round(q: Q) -> int:
    """cvRound(float): nearest integer, ties to even (App. A preamble)."""
    fl = math.floor(q)
    rem = q - fl
    if rem > Q(1, 2) or (rem == Q(1, 2) and fl % 2 == 1):
        fl += 1
    return fl


def sat_u8(v: int) -> int:
    return 0 if v < 0 else (255 if v > 255 else v)


def equalize_hist(src, trace=None):
    """App. A.1 on a list of rows of ints."""
    flat = [v for row in src for v in row]
    total = len(flat)
    hist = [0] * 256
    for v in flat:
        hist[v] += 1
    i = next(b for b in range(256) if hist[b])
    if hist[i] == total:                                             # step 4
        return [[i] * len(row) for row in src]
    scale = rn32(Q(255) / Q(total - hist[i]))                         # step 5: ONE float division
    lut = [0] * 256
    s = 0
    for j in range(i + 1, 256):                                       # step 6
        s += hist[j]
        prod = rn32(Q(s) * scale)                                     # int -> float exact (s < 2^24), one float multiply
        lut[j] = sat_u8(cv_round(prod))
        if trace is not None and hist[j]:
            trace.append(f"lut[{j}]: sum={s}, scale=rn32(255/{total - hist[i]})={float(scale)!r}, rn32(sum*scale)={float(prod)!r} -> {lut[j]}")
    return [[lut[v] for v in row] for row in src]


def reflect101(p: int, n: int) -> int:
    """borderInterpolate(p, n, BORDER_REFLECT_101) for p >= 0: ...cb|abcdefgh|gfedcba...; a single row / column repeats."""
    if n == 1:
        return 0
    while p >= n:
        p = 2 * (n - 1) - p
        if p < 0:
            p = -p
    return p


def clahe(src, clip_limit: float, tiles_x: int, tiles_y: int, trace_px=None, trace=None):
    """App. A.2 on a list of rows of ints.  trace_px = (y, x): record that pixel's interpolation step by step into `trace`."""
    H, W = len(src), len(src[0])
    if W % tiles_x == 0 and H % tiles_y == 0:                         # step 1
        ew, eh = W, H
    else:
        ew, eh = W + (tiles_x - W % tiles_x), H + (tiles_y - H % tiles_y)     # BOTH pads whenever EITHER axis is indivisible
    tw, th = ew // tiles_x, eh // tiles_y
    ext = [[src[reflect101(y, H)][reflect101(x, W)] for x in range(ew)] for y in range(eh)]
    area = tw * th                                                    # step 2
    lut_scale = rn32(Q(255) / Q(area))
    clip = 0
    if clip_limit > 0.0:
        clip = max(int(float(clip_limit) * area / 256), 1)            # double math, truncation
    if trace is not None:
        trace.append(f"ext {ew}x{eh}, tile {tw}x{th}, area {area}, clip {clip}, lutScale rn32(255/{area})={float(lut_scale)!r}")
    luts = {}
    for ty in range(tiles_y):                                         # step 3
        for tx in range(tiles_x):
            h = [0] * 256
            for y in range(ty * th, (ty + 1) * th):
                for x in range(tx * tw, (tx + 1) * tw):
                    h[ext[y][x]] += 1
            if clip > 0:
                clipped = sum(max(c - clip, 0) for c in h)
                h = [min(c, clip) for c in h]
                batch = clipped // 256
                resid = clipped - batch * 256
                h = [c + batch for c in h]
                if resid:
                    step = max(256 // resid, 1)
                    b = 0
                    while b < 256 and resid > 0:                      # ONE pass
                        h[b] += 1
                        b += step
                        resid -= 1
            lut, s = [0] * 256, 0
            for b in range(256):
                s += h[b]
                lut[b] = sat_u8(cv_round(rn32(Q(s) * lut_scale)))
            luts[(ty, tx)] = lut
            if trace is not None:
                seen = sorted({ext[y][x] for y in range(ty * th, (ty + 1) * th) for x in range(tx * tw, (tx + 1) * tw)})
                trace.append(f"tile ({tx},{ty}): " + ", ".join(f"LUT[{v}]={lut[v]}" for v in seen))

    def axis(p, tile, ntiles):                                        # step 4
        inv = rn32(Q(1) / Q(tile))
        tf = rn32(rn32(Q(p) * inv) - Q(1, 2))                         # multiply rounded, THEN subtract rounded
        t1 = math.floor(tf)
        a = rn32(tf - Q(t1))                                          # weights BEFORE clamping
        a1 = rn32(Q(1) - a)
        return max(t1, 0), min(t1 + 1, ntiles - 1), a, a1, tf

    out = [[0] * W for _ in range(H)]
    for y in range(H):
        ty1, ty2, ya, ya1, tyf = axis(y, th, tiles_y)
        for x in range(W):
            tx1, tx2, xa, xa1, txf = axis(x, tw, tiles_x)
            v = src[y][x]
            a, b, c, d = (Q(luts[(ty1, tx1)][v]), Q(luts[(ty1, tx2)][v]), Q(luts[(ty2, tx1)][v]), Q(luts[(ty2, tx2)][v]))
            top = rn32(rn32(rn32(a * xa1) + rn32(b * xa)) * ya1)      # step 5: nine separately rounded operations
            bot = rn32(rn32(rn32(c * xa1) + rn32(d * xa)) * ya)
            res = rn32(top + bot)
            out[y][x] = sat_u8(cv_round(res))
            if trace is not None and trace_px == (y, x):
                f = lambda q: repr(float(q))
                trace.append(f"pixel (y={y}, x={x}) v={v}: txf={f(txf)} tx1,tx2={tx1},{tx2} xa={f(xa)} xa1={f(xa1)}; tyf={f(tyf)} ty1,ty2={ty1},{ty2} "
                             f"ya={f(ya)} ya1={f(ya1)}; a,b,c,d={int(a)},{int(b)},{int(c)},{int(d)}; top={f(top)} bot={f(bot)} res={f(res)} -> {out[y][x]}")
    return out


def derived_answer(k, trace=None):
    """The answer of one "derived" KAT entry of kat.json, recomputed from Appendix A in exact arithmetic."""
    h, w = k["shape"]
    src = [list(k["src"][r * w:(r + 1) * w]) for r in range(h)]
    if k["op"] == "equalize":
        return equalize_hist(src, trace)
    px = tuple(k["deciding_pixel"]) if "deciding_pixel" in k else None
    return clahe(src, k["clip"], k["tiles"][0], k["tiles"][1], px, trace)


if __name__ == "__main__":
    kat = json.loads((Path(__file__).parent / "kat.json").read_text())
    for k in kat.get("derived", []):
        tr = []
        out = derived_answer(k, tr)
        flat = [v for row in out for v in row]
        print(f"{k['id']}: {'matches kat.json' if flat == k['dst'] else 'DIFFERS FROM kat.json'}")
        for line in tr:
            print("    " + line)


# ---- colour neighbours: plain integers, one pixel at a time, constants as oracle/color_oracle.c's header states them ----
def _descale14(x: int) -> int:
    return (x + (1 << 13)) >> 14                    # CV_DESCALE(x, 14); Python's >> floors like C's arithmetic shift


def bgr2yuv_px(b: int, g: int, r: int):
    """cv::cvtColor(COLOR_BGR2YUV) on one 8-bit pixel: RGB2YCrCb_i<uchar>, yuv order."""
    y = _descale14(b * 1868 + g * 9617 + r * 4899)
    u = _descale14((b - y) * 8061 + (128 << 14))
    v = _descale14((r - y) * 14369 + (128 << 14))
    return [sat_u8(y), sat_u8(u), sat_u8(v)]


def yuv2bgr_px(y: int, u: int, v: int):
    """cv::cvtColor(COLOR_YUV2BGR) on one 8-bit pixel: YCrCb2RGB_i<uchar>, yuv order."""
    b = y + _descale14((u - 128) * 33292)
    g = y + _descale14((u - 128) * -6472 + (v - 128) * -9519)
    r = y + _descale14((v - 128) * 18678)
    return [sat_u8(b), sat_u8(g), sat_u8(r)]


def nv12_to_bgr(nv12, width: int, height: int):
    """COLOR_YUV2BGR_NV12 (ITUR_BT_601_SHIFT = 20): the four pixels of a 2x2 block share one (U, V)."""
    out = []
    for yy in range(height):
        for xx in range(width):
            base = width * height + (yy // 2) * width + (xx & ~1)
            u, v = nv12[base] - 128, nv12[base + 1] - 128
            luma = max(0, nv12[yy * width + xx] - 16) * 1220542
            out += [sat_u8((luma + (1 << 19) + 2116026 * u) >> 20),
                    sat_u8((luma + (1 << 19) - 852492 * v - 409993 * u) >> 20),
                    sat_u8((luma + (1 << 19) + 1673527 * v) >> 20)]
    return out


def bgr_to_nv12(bgr, width: int, height: int):
    """COLOR_BGR2YUV_I420 arithmetic with U, V interleaved: luma of every pixel, chroma of the TOP-LEFT pixel of each 2x2 block."""
    px = lambda yy, xx: bgr[3 * (yy * width + xx): 3 * (yy * width + xx) + 3]
    ys, uv = [], []
    for yy in range(height):
        for xx in range(width):
            b, g, r = px(yy, xx)
            ys.append(sat_u8((269484 * r + 528482 * g + 102760 * b + (1 << 19) + (16 << 20)) >> 20))
    for yy in range(0, height, 2):
        for xx in range(0, width, 2):
            b, g, r = px(yy, xx)
            uv += [sat_u8((-155188 * r - 305135 * g + 460324 * b + (1 << 19) + (128 << 20)) >> 20),
                   sat_u8((460324 * r - 385875 * g - 74448 * b + (1 << 19) + (128 << 20)) >> 20)]
    return ys + uv


def color_answer(k):
    """The answer of one entry of kat.json's "color" list."""
    if k["op"] == "bgr2yuv":
        return [c for p in k["src"] for c in bgr2yuv_px(*p)]
    if k["op"] == "yuv2bgr":
        return [c for p in k["src"] for c in yuv2bgr_px(*p)]
    if k["op"] == "nv12_to_bgr":
        return nv12_to_bgr(k["src"], *k["shape"])
    if k["op"] == "bgr_to_nv12":
        return bgr_to_nv12(k["src"], *k["shape"])
    raise KeyError(k["op"])
