"""The kill matrix of the COLOUR oracle's known answers (SURVEY 8f row N3 -- the weakest-pinned rows: cv::cvtColor BGR <-> YUV around the
luma op, singlecolor.cpp:39-66 / clahe1frame.cpp:83-102; COLOR_BGR2YUV_I420, 1frameMeasure.cpp:32; the NV12 pair of BASELINE config 5).
Same construction as tests/test_kat_kill_matrix.py: tests/color_mutants.py unmutated IS the oracle, each mutant gets one thing wrong
(a missing rounding constant, Cr / Cb order, a truncated fixed-point coefficient, green from two roundings, full-range luma, NV21,
averaged chroma ...), and every mutant must fail at least one entry of kat.json's "color" list.  The answers come from
tests/golden/derive_kats.py (plain integers, a pixel at a time).  Parity stays unpinned: this pins the oracle to the constants its header
states; tests/test_oracle_vs_opencv.py compares with a real cv2 wherever one exists."""
import json
import sys
from pathlib import Path

import numpy as np

import oracle

HERE = Path(__file__).parent
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(HERE / "golden"))
import color_mutants as C  # noqa: E402
import derive_kats  # noqa: E402

KATS = json.loads((HERE / "golden" / "kat.json").read_text())["color"]
FAMILY = {"bgr2yuv": "yuv_", "yuv2bgr": "bgr_", "nv12_to_bgr": "dec_", "bgr_to_nv12": "enc_"}


def _run(k, impl, mutant=None):
    """flat output of one known answer's op through `impl` ("oracle" = oracle/color_oracle.c, "mutants" = tests/color_mutants.py)"""
    if k["op"] in ("bgr2yuv", "yuv2bgr"):
        px = np.array(k["src"], np.uint8).reshape(1, -1, 3).copy()              # (a fresh array: a broadcast axis would have stride 0)
        if impl == "oracle":
            out = oracle.bgr2yuv(px) if k["op"] == "bgr2yuv" else oracle.yuv2bgr(px)
        else:
            out = C.bgr2yuv(px, mutant) if k["op"] == "bgr2yuv" else C.yuv2bgr(px, mutant)
        return out.reshape(-1).tolist()
    w, h = k["shape"]
    if k["op"] == "nv12_to_bgr":
        nv = np.array(k["src"], np.uint8)
        out = oracle.nv12_to_bgr(nv, w, h) if impl == "oracle" else C.nv12_to_bgr(nv, w, h, mutant)
        return np.asarray(out).reshape(-1).tolist()
    bgr = np.array(k["src"], np.uint8).reshape(h, w, 3)
    out = oracle.bgr_to_nv12(bgr) if impl == "oracle" else C.bgr_to_nv12(bgr, mutant)
    return np.asarray(out).reshape(-1).tolist()


def test_unmutated_colour_harness_is_the_oracle():
    rng = np.random.default_rng(20261006)
    for _ in range(25):
        h, w = 2 * int(rng.integers(1, 9)), 2 * int(rng.integers(1, 12))
        a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        assert np.array_equal(C.bgr2yuv(a), oracle.bgr2yuv(a)) and np.array_equal(C.yuv2bgr(a), oracle.yuv2bgr(a))
        nv = rng.integers(0, 256, w * h * 3 // 2, dtype=np.uint8)
        assert np.array_equal(C.nv12_to_bgr(nv, w, h), oracle.nv12_to_bgr(nv, w, h))
        assert np.array_equal(C.bgr_to_nv12(a), oracle.bgr_to_nv12(a))


def test_colour_known_answers_three_ways():
    """oracle/color_oracle.c, the switchable restatement and the integer derivation agree with kat.json; I420 is the NV12 encode with
    planar chroma (1frameMeasure.cpp:32)."""
    for k in KATS:
        assert _run(k, "oracle") == k["dst"], k["id"]
        assert _run(k, "mutants") == k["dst"], k["id"]
        assert derive_kats.color_answer(k) == k["dst"], k["id"]
        if k["op"] == "bgr_to_nv12":
            w, h = k["shape"]
            i420 = oracle.bgr_to_i420(np.array(k["src"], np.uint8).reshape(h, w, 3)).reshape(-1).tolist()
            n = w * h
            assert i420[:n] == k["dst"][:n] and i420[n:n + n // 4] == k["dst"][n::2] and i420[n + n // 4:] == k["dst"][n + 1::2]


def test_every_colour_mutant_is_killed(capsys):
    rows, survivors = [], []
    for name in C.MUTANTS:
        killed = [k["id"] for k in KATS if FAMILY[k["op"]] == name[:4] and _run(k, "mutants", name) != k["dst"]]
        worst = max([max(abs(a - b) for a, b in zip(_run(k, "mutants", name), k["dst"])) for k in KATS if FAMILY[k["op"]] == name[:4]] + [0])
        rows.append((name, killed, worst))
        if not killed:
            survivors.append(name)
    with capsys.disabled():
        print("\ncolour kill matrix: mutant of the restated color_yuv arithmetic -> known answers (kat.json \"color\") it FAILS")
        for name, killed, worst in rows:
            print(f"  {name:30s} {','.join(killed) or 'SURVIVES':24s} worst |diff| {worst:3d}   {C.MUTANTS[name]}")
        print(f"  {len(rows)} mutants, {len(survivors)} survive")
    assert survivors == []
