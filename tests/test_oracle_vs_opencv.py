"""Pins the oracle against a REAL OpenCV wherever one is importable (none is in the authoring image or
on the GPU box, so this normally skips -- which is why DESIGN.md says 'parity unpinned').  If cv2 is
present, any disagreement here means the restatement in oracle/ (and therefore the HIP kernels) is wrong."""
import numpy as np
import pytest

cv2 = pytest.importorskip("cv2")

import oracle  # noqa: E402
from mi_lumaeq import synth  # noqa: E402

SHAPES = [(1, 1), (3, 5), (47, 63), (48, 64), (15, 16), (270, 480), (1079, 1919), (1080, 1920)]


@pytest.mark.parametrize("shape", SHAPES, ids=str)
@pytest.mark.parametrize("dist", synth.DISTS)
def test_equalize_matches_cv2(shape, dist):
    h, w = shape
    src = synth.y_plane(w, h, dist, 31)
    assert np.array_equal(oracle.equalize_hist(src), cv2.equalizeHist(src))


def _cv2_fp_mode():
    """Which of the oracle's two CLAHE arithmetic modes this cv2 build computes: separately rounded (x86-64 baseline) or
    GCC's FMA contraction (aarch64 / -mfma builds).  Decided once on a frame where the two modes differ."""
    src = synth.y_plane(1920, 1080, "D2", 5)
    want = cv2.createCLAHE(clipLimit=2.0, tileGridSize=(8, 8)).apply(src)
    for mode in (False, True):
        old = oracle.set_fp_contract(mode)
        try:
            if np.array_equal(oracle.clahe(src, 2.0, 8, 8), want):
                return mode
        finally:
            oracle.set_fp_contract(old)
    pytest.fail(f"cv2 {cv2.__version__} CLAHE matches neither arithmetic mode of the oracle")


@pytest.mark.parametrize("shape", SHAPES, ids=str)
@pytest.mark.parametrize("cfg", [(2.0, 8, 8), (3.0, 4, 4), (40.0, 8, 8), (1.5, 1, 1), (2.0, 16, 2)], ids=str)
def test_clahe_matches_cv2(shape, cfg):
    h, w = shape
    clip, tx, ty = cfg
    old = oracle.set_fp_contract(_cv2_fp_mode())
    try:
        for dist in ("D1", "D2", "D3"):
            src = synth.y_plane(w, h, dist, 32)
            want = cv2.createCLAHE(clipLimit=clip, tileGridSize=(tx, ty)).apply(src)
            assert np.array_equal(oracle.clahe(src, clip, tx, ty), want), (dist, cv2.__version__)
    finally:
        oracle.set_fp_contract(old)


def test_color_conversions_match_cv2():
    rng = np.random.default_rng(8)
    a = rng.integers(0, 256, (97, 131, 3), dtype=np.uint8)
    yuv = cv2.cvtColor(a, cv2.COLOR_BGR2YUV)
    assert np.array_equal(oracle.bgr2yuv(a), yuv)
    assert np.array_equal(oracle.yuv2bgr(yuv), cv2.cvtColor(yuv, cv2.COLOR_YUV2BGR))
