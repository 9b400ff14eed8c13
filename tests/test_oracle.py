"""CPU tests of the oracle itself: SURVEY.md Appendix B known answers, structural properties, the
two independent restatements against each other, and the committed golden fixtures."""
import json
from pathlib import Path

import numpy as np
import pytest

import oracle
from mi_lumaeq import synth

GOLD = Path(__file__).parent / "golden"
KAT = json.loads((GOLD / "kat.json").read_text())


def _kat_src(k):
    h, w = k["shape"]
    if "src" in k:
        return np.array(k["src"], np.uint8).reshape(h, w)
    if "src_runs" in k:
        return np.concatenate([np.full(n, v, np.uint8) for v, n in k["src_runs"]]).reshape(h, w)
    if "src_const" in k:
        return np.full((h, w), k["src_const"], np.uint8)
    if "src_arange" in k:
        return np.arange(k["src_arange"], dtype=np.uint8).reshape(h, w)
    if "src_quadrants" in k:                 # four constant quadrants: top-left, top-right, bottom-left, bottom-right
        q = k["src_quadrants"]
        a = np.empty((h, w), np.uint8)
        a[: h // 2, : w // 2], a[: h // 2, w // 2:], a[h // 2:, : w // 2], a[h // 2:, w // 2:] = q
        return a
    raise KeyError(k["id"])


@pytest.mark.parametrize("k", KAT["equalize"], ids=lambda k: k["id"])
@pytest.mark.parametrize("impl", ["c", "numpy"])
def test_equalize_kat(k, impl):
    src = _kat_src(k)
    out = oracle.equalize_hist(src) if impl == "c" else oracle.np_equalize_hist(src)
    if "dst" in k:
        assert out.reshape(-1).tolist() == k["dst"]
    if "dst_const" in k:
        assert (out == k["dst_const"]).all()
    if "dst_arange" in k:
        assert out.reshape(-1).tolist() == list(range(k["dst_arange"]))
    if "lut" in k:                      # EQ-2: tie direction (round half to even)
        for v, want in k["lut"].items():
            assert (out[src == int(v)] == want).all(), (v, want)


@pytest.mark.parametrize("impl", ["c", "numpy"])
def test_clahe_kat_cl1(impl):
    k = KAT["clahe"][0]
    src = _kat_src(k)
    f = oracle.clahe if impl == "c" else oracle.np_clahe
    out = f(src, k["clip"], *k["tiles"])
    assert (out == k["dst_const"]).all()


@pytest.mark.parametrize("k", [k for k in KAT["clahe"] if k["id"] in ("CL-4", "CL-5")], ids=lambda k: k["id"])
@pytest.mark.parametrize("impl", ["c", "numpy"])
def test_clahe_kat_interpolation_weights_and_lut_tie(k, impl):
    """CL-4: the bilinear weights (computed before the tile indices are clamped) on an image whose tile LUTs are step functions;
    CL-5: round-half-to-even inside CLAHE's own LUT (cvRound of cum * lutScale)."""
    src = _kat_src(k)
    f = oracle.clahe if impl == "c" else oracle.np_clahe
    out = f(src, k["clip"], *k["tiles"])
    if "dst" in k:
        assert out.reshape(-1).tolist() == k["dst"]
    if "lut" in k:
        for v, want in k["lut"].items():
            assert (out[src == int(v)] == want).all(), (v, want)


def test_clahe_kat_cl3_pad_quirk():
    k = KAT["clahe"][1]
    h, w = k["shape"]
    g = oracle.np_clahe_geometry(w, h, 2.0, *k["tiles"])
    assert [g["ext_w"], g["ext_h"]] == k["ext"] and [g["tile_w"], g["tile_h"]] == k["tile"]
    # behavioural form: the LUT of the last tile column must be built from REFLECTED columns 14,13,...
    rng = np.random.default_rng(3)
    src = rng.integers(0, 256, (h, w), dtype=np.uint8)
    luts = oracle.clahe_tile_luts(src, 0.0, 8, 8).reshape(8, 8, 256)
    ext_cols = [16 - 2 - c for c in range(8)]            # index 16+c reads 16-2-c
    tile = src[0:2][:, [ext_cols[5], ext_cols[6], ext_cols[7]]]     # tile (ty=0, tx=7) = ext cols 21..23
    hist = np.bincount(tile.reshape(-1), minlength=256)
    want = np.clip(np.rint(np.cumsum(hist).astype(np.float32) * (np.float32(255) / np.float32(6))), 0, 255)
    assert np.array_equal(luts[0, 7], want.astype(np.uint8))


def test_clahe_grid_1x1_is_plain_lut():
    """CL-2: with one tile all four taps are the same LUT, so output == LUT0[src] exactly."""
    rng = np.random.default_rng(5)
    src = rng.integers(0, 256, (37, 53), dtype=np.uint8)
    lut = oracle.clahe_tile_luts(src, 4.0, 1, 1)[0]
    assert np.array_equal(oracle.clahe(src, 4.0, 1, 1), lut[src])


SHAPES = [(1, 1), (1, 17), (3, 4097), (47, 63), (48, 64), (15, 16), (270, 480), (135, 241)]


@pytest.mark.parametrize("shape", SHAPES, ids=str)
@pytest.mark.parametrize("dist", synth.DISTS)
def test_c_vs_numpy_equalize(shape, dist):
    h, w = shape
    src = synth.y_plane(w, h, dist, 11)
    assert np.array_equal(oracle.equalize_hist(src), oracle.np_equalize_hist(src))


@pytest.mark.parametrize("shape", SHAPES, ids=str)
@pytest.mark.parametrize("cfg", [(2.0, 8, 8), (3.0, 4, 4), (40.0, 8, 8), (0.0, 3, 5), (1.5, 1, 1), (2.0, 16, 2)], ids=str)
def test_c_vs_numpy_clahe(shape, cfg):
    h, w = shape
    clip, tx, ty = cfg
    for dist in ("D1", "D2", "D3"):
        src = synth.y_plane(w, h, dist, 12)
        assert np.array_equal(oracle.clahe(src, clip, tx, ty), oracle.np_clahe(src, clip, tx, ty)), dist


def test_equalize_properties():
    rng = np.random.default_rng(9)
    src = np.clip(rng.normal(100, 20, (120, 200)), 0, 255).astype(np.uint8)
    h = oracle.hist(src)
    assert h.sum() == src.size and np.array_equal(h, np.bincount(src.reshape(-1), minlength=256))
    lut, first = oracle.equalize_lut(h, src.size)
    nz = np.flatnonzero(h)
    assert first == nz[0] and lut[first] == 0 and lut[nz[-1]] == 255
    assert (np.diff(lut[first:].astype(int)) >= 0).all()             # monotone non-decreasing
    # permutation invariance of pixel order
    perm = rng.permutation(src.size)
    a = oracle.equalize_hist(src).reshape(-1)[perm]
    b = oracle.equalize_hist(src.reshape(-1)[perm].reshape(src.shape)).reshape(-1)
    assert np.array_equal(a, b)
    # idempotent LUT application through strided views and in place
    big = np.zeros((130, 260), np.uint8)
    view = big[5:125, 30:230]
    view[:] = src
    out = oracle.equalize_hist(view)
    assert np.array_equal(out, oracle.equalize_hist(src))
    oracle.equalize_hist(view, view)
    assert np.array_equal(view, out) and big[0].sum() == 0


def test_nv12_frame_uv_modes():
    w, h = 64, 36
    f = synth.nv12_frame(w, h, "D2", 3)
    fill = oracle.nv12_frame(f, w, h, uv_mode=0)
    copy = oracle.nv12_frame(f, w, h, uv_mode=1)
    y = oracle.equalize_hist(f[: w * h].reshape(h, w)).reshape(-1)
    assert np.array_equal(fill[: w * h], y) and np.array_equal(copy[: w * h], y)
    assert (fill[w * h:] == 128).all() and np.array_equal(copy[w * h:], f[w * h:])


def test_golden_fixtures():
    """Committed fixtures (tests/golden/make_golden.py): inputs from seeds, outputs from the oracle at
    the time of commit -- guards against silent drift of the restatement."""
    z = np.load(GOLD / "golden_small.npz")
    names = sorted({k.rsplit("__", 1)[0] for k in z.files})
    assert names
    for n in names:
        src = z[n + "__src"]
        if n.startswith("eq"):
            got = oracle.equalize_hist(src)
        else:
            clip, tx, ty = z[n + "__cfg"]
            got = oracle.clahe(src, float(clip), int(tx), int(ty))
        assert np.array_equal(got, z[n + "__dst"]), n


def test_color_conversion_kats():
    """SURVEY 8f N3 (parity unpinned): hand-computed BT.601 fixed-point values of COLOR_BGR2YUV / COLOR_YUV2BGR."""
    px = np.array([[[255, 255, 255], [255, 0, 0], [0, 0, 0], [128, 128, 128], [0, 0, 255]]], np.uint8)   # B,G,R
    yuv = oracle.bgr2yuv(px).reshape(-1, 3).tolist()
    assert yuv[0] == [255, 128, 128] and yuv[2] == [0, 128, 128] and yuv[3] == [128, 128, 128]
    assert yuv[1] == [29, 239, 103]              # pure blue: Y=(255*1868+8192)>>14, U=((226*8061)+(128<<14)+8192)>>14 ...
    assert yuv[4] == [76, 91, 255]               # pure red: V saturates
    back = oracle.yuv2bgr(oracle.bgr2yuv(px)).reshape(-1, 3).tolist()
    assert back[0] == [255, 255, 255] and back[1] == [255, 0, 0] and back[2] == [0, 0, 0] and back[3] == [128, 128, 128]
    # independent numpy restatement
    rng = np.random.default_rng(4)
    a = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    b, g, r = [a[..., i].astype(np.int64) for i in range(3)]
    Y = (b * 1868 + g * 9617 + r * 4899 + 8192) >> 14
    U = ((b - Y) * 8061 + (128 << 14) + 8192) >> 14
    V = ((r - Y) * 14369 + (128 << 14) + 8192) >> 14
    ref = np.stack([np.clip(Y, 0, 255), np.clip(U, 0, 255), np.clip(V, 0, 255)], -1).astype(np.uint8)
    assert np.array_equal(oracle.bgr2yuv(a), ref)
    y, u, v = [ref[..., i].astype(np.int64) for i in range(3)]
    B = y + (((u - 128) * 33292 + 8192) >> 14)
    G = y + (((u - 128) * -6472 + (v - 128) * -9519 + 8192) >> 14)
    R = y + (((v - 128) * 18678 + 8192) >> 14)
    ref2 = np.stack([np.clip(B, 0, 255), np.clip(G, 0, 255), np.clip(R, 0, 255)], -1).astype(np.uint8)
    assert np.array_equal(oracle.yuv2bgr(ref), ref2)
    # composite pipeline = the pieces
    yuv_img = oracle.bgr2yuv(a)
    yuv_img2 = yuv_img.copy()
    yuv_img2[..., 0] = oracle.equalize_hist(np.ascontiguousarray(yuv_img[..., 0]))
    assert np.array_equal(oracle.bgr_luma_op(a, 0), oracle.yuv2bgr(yuv_img2))


def test_nv12_bgr_channel_equalize_oracle():
    """BASELINE config 5 read literally (parity unpinned): 4:2:0 BT.601 KATs, C vs numpy, composite = pieces."""
    # decode KATs (hand-computed, shift 20): a grey 2x2 block (U=V=128) and a block carrying the chroma of pure red (U=90, V=240)
    nv = np.array([16, 235, 81, 0,
                   126, 81, 255, 16,
                   128, 128, 90, 240], np.uint8)
    bgr = oracle.nv12_to_bgr(nv, 4, 2)
    assert bgr[0, 0].tolist() == [0, 0, 0] and bgr[0, 1].tolist() == [255, 255, 255]       # video black / white
    assert bgr[1, 0].tolist() == [128, 128, 128]        # (110*1220542 + 2^19) >> 20
    assert bgr[1, 1].tolist() == [76, 76, 76]           # ( 65*1220542 + 2^19) >> 20
    assert bgr[0, 2].tolist() == [0, 0, 254]            # R = (65*1220542 + 2^19 + 1673527*112) >> 20; G, B go negative -> 0
    assert bgr[0, 3].tolist() == [0, 0, 179]            # Y < 16 clamps to 0: R = (2^19 + 1673527*112) >> 20
    assert bgr[1, 2].tolist() == [202, 202, 255]        # Y=255: B = (239*1220542 + 2^19 - 2116026*38) >> 20
    # encode KATs: luma of every pixel, chroma from the top-left pixel of the block only
    px = np.array([[[0, 0, 255], [255, 255, 255]], [[0, 0, 0], [128, 128, 128]]], np.uint8)  # B,G,R
    out = oracle.bgr_to_nv12(px)
    assert out.tolist() == [82, 235, 16, 126, 90, 240]
    assert (269484 * 255 + (1 << 19) + (16 << 20)) >> 20 == 82 and (460324 * 255 + (1 << 19) + (128 << 20)) >> 20 == 240
    # a grey frame on the video range survives decode + encode unchanged
    ramp = np.arange(16, 236, dtype=np.uint8)
    grey = np.concatenate([ramp, ramp[::-1], np.full(220, 128, np.uint8)])
    assert np.array_equal(oracle.bgr_to_nv12(oracle.nv12_to_bgr(grey, 220, 2)), grey)
    # C vs numpy on random and structured frames
    rng = np.random.default_rng(12)
    for (w, h) in [(2, 2), (16, 2), (18, 6), (64, 48), (322, 178)]:
        a = rng.integers(0, 256, w * h * 3 // 2, dtype=np.uint8)
        assert np.array_equal(oracle.nv12_bgr_equalize(a, w, h), oracle.np_nv12_bgr_equalize(a, w, h)), (w, h)
        lo = (a // 4 + 90).astype(np.uint8)                        # low-contrast frame
        assert np.array_equal(oracle.nv12_bgr_equalize(lo, w, h), oracle.np_nv12_bgr_equalize(lo, w, h)), (w, h)
    # composite = pieces
    w, h = 64, 48
    a = rng.integers(0, 256, w * h * 3 // 2, dtype=np.uint8)
    bgr = oracle.nv12_to_bgr(a, w, h)
    for c in range(3):
        bgr[..., c] = oracle.equalize_hist(np.ascontiguousarray(bgr[..., c]))
    assert np.array_equal(oracle.nv12_bgr_equalize(a, w, h), oracle.bgr_to_nv12(bgr))
    # COLOR_BGR2YUV_I420 (1frameMeasure.cpp:32): same values as the NV12 encode, planar chroma, (H*3/2, W) matrix
    i420 = oracle.bgr_to_i420(bgr)
    nvv = oracle.bgr_to_nv12(bgr)
    assert i420.shape == (h * 3 // 2, w)
    assert np.array_equal(i420.reshape(-1)[:w * h], nvv[:w * h])
    assert np.array_equal(i420.reshape(-1)[w * h:w * h * 5 // 4], nvv[w * h::2]) and np.array_equal(i420.reshape(-1)[w * h * 5 // 4:], nvv[w * h + 1::2])
    # odd sizes are rejected (OpenCV asserts even dimensions for 4:2:0)
    with pytest.raises(ValueError):
        oracle.nv12_bgr_equalize(np.zeros(3 * 2 * 3 // 2, np.uint8), 3, 2)
    assert oracle.nv12_bgr_equalize(np.zeros(0, np.uint8), 0, 0).size == 0


@pytest.mark.parametrize("shape", [(1, 1), (15, 16), (47, 63), (64, 48)], ids=str)
@pytest.mark.parametrize("cfg", [(2.0, 8, 8), (3.0, 4, 4), (0.0, 3, 5), (40.0, 1, 1)], ids=str)
def test_c_vs_numpy_clahe16(shape, cfg):
    """16-bit CLAHE (65 536 bins), the two restatements against each other."""
    h, w = shape
    clip, tx, ty = cfg
    rng = np.random.default_rng(h * 131 + w)
    for kind in range(3):
        if kind == 0:
            s = rng.integers(0, 65536, (h, w), dtype=np.uint16)
        elif kind == 1:
            s = rng.integers(1000, 1400, (h, w), dtype=np.uint16)
        else:
            s = np.full((h, w), 777, np.uint16)
        assert np.array_equal(oracle.clahe16(s, clip, tx, ty), oracle.np_clahe(s, clip, tx, ty)), kind


def test_hypothesis_c_vs_numpy():
    """Property test (hypothesis): the C and numpy restatements agree on arbitrary small images and CLAHE configs."""
    hyp = pytest.importorskip("hypothesis")
    from hypothesis import given, settings, strategies as st
    from hypothesis.extra import numpy as hnp

    img = hnp.arrays(np.uint8, st.tuples(st.integers(1, 40), st.integers(1, 40)), elements=st.integers(0, 255))

    @settings(max_examples=80, deadline=None)
    @given(img)
    def eq(a):
        assert np.array_equal(oracle.equalize_hist(a), oracle.np_equalize_hist(a))

    @settings(max_examples=80, deadline=None)
    @given(img, st.sampled_from([0.0, 0.3, 1.0, 2.0, 3.0, 40.0]), st.integers(1, 9), st.integers(1, 9))
    def cl(a, clip, tx, ty):
        assert np.array_equal(oracle.clahe(a, clip, tx, ty), oracle.np_clahe(a, clip, tx, ty))

    eq()
    cl()


def test_oracle_under_sanitizers(tmp_path):
    """The C restatement built with -fsanitize=address,undefined and run on ragged shapes/strides (CPU build only)."""
    import shutil
    import subprocess
    root = Path(__file__).resolve().parents[1]
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    exe = tmp_path / "oracle_sanitize"
    cmd = ["gcc", "-O1", "-g", "-std=c11", "-ffp-contract=off", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           str(root / "tests" / "cxx" / "oracle_sanitize.c"), str(root / "oracle" / "lumaeq_oracle.c"), str(root / "oracle" / "color_oracle.c"),
           "-o", str(exe), "-lm"]
    b = subprocess.run(cmd, capture_output=True, text=True)
    if b.returncode != 0 and "sanitize" in b.stderr:
        pytest.skip("sanitizer runtime not available")
    assert b.returncode == 0, b.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "clean" in r.stdout, r.stdout + r.stderr


def test_photo_like_scene():
    """Synthetic photo-like scene (mi_lumaeq.synth.photo_like; odd-sized, so CLAHE pads; tests/golden/make_photo_like_crc.py): the
    generator still produces the recorded bytes on this platform, the C oracle the recorded outputs, and the independent numpy
    restatement agrees on it.  (No image of the reference ships: SURVEY.md 2, `hun.png` row.)"""
    import json
    import zlib
    from mi_lumaeq import synth
    z = json.loads((Path(__file__).parent / "golden" / "photo_like_crc.json").read_text())
    y = synth.photo_like(z["width"], z["height"], z["seed"])
    crop = synth.photo_like(384, 256, z["seed"] + 1, channels=3)
    assert y.shape == (1079, 1919) and crop.shape == (256, 384, 3)
    crc = lambda a: zlib.crc32(np.ascontiguousarray(a).tobytes())
    assert crc(y) == z["crc_input_y"] and crc(crop) == z["crc_input_bgr_crop"]
    hist = np.bincount(y.reshape(-1), minlength=256)
    assert (hist > 0).sum() > 100 and hist.max() > y.size // 40           # many levels AND hot bins (flat regions)
    assert (y[:, 1:] == y[:, :-1]).mean() > 0.3                             # runs of equal neighbours
    eq = oracle.equalize_hist(y)
    assert crc(eq) == z["crc_equalize"]
    assert np.array_equal(eq, oracle.np_equalize_hist(y))
    c8 = oracle.clahe(y, 2.0, 8, 8)
    assert crc(c8) == z["crc_clahe_2_8x8"] and crc(oracle.clahe(y, 3.0, 4, 4)) == z["crc_clahe_3_4x4"]
    assert np.array_equal(c8, oracle.np_clahe(y, 2.0, 8, 8))
    old = oracle.set_fp_contract(True)                          # the FMA-contracted flavour of the blend has its own recorded bytes
    try:
        c8f = oracle.clahe(y, 2.0, 8, 8)
    finally:
        oracle.set_fp_contract(old)
    assert crc(c8f) == z["crc_clahe_2_8x8_fp_contract"] and not np.array_equal(c8f, c8)
    assert crc(oracle.bgr_luma_op(crop, 0)) == z["crc_bgr_luma_equalize_crop"]
    assert crc(oracle.bgr_luma_op(crop, 1, 3.0, 4, 4)) == z["crc_bgr_luma_clahe_crop"]


def test_fp_contract_mode_matches_what_gcc_does(tmp_path):
    """The oracle's second arithmetic mode (set_fp_contract: the FMAs GCC forms from clahe.cpp's expressions on FMA targets,
    i.e. a distribution OpenCV on the reference's aarch64 board) against the expressions compiled by GCC itself, once with
    -ffp-contract=off (= mode 0) and, where this CPU has FMA, once with -mfma -ffp-contract=fast (= mode 1)."""
    import ctypes
    import struct
    import subprocess
    src = Path(__file__).parent / "cxx" / "contract_probe.cpp"
    rng = np.random.default_rng(11)
    n = 4000
    xs = rng.integers(0, 4096, n)
    invs = (np.float32(1.0) / rng.choice(np.array([480, 270, 240, 135, 60, 7, 3], np.float32), n)).astype(np.float32)
    ls = rng.integers(0, 256, (n, 4))
    xa = rng.random(n, dtype=np.float32)
    ya = rng.random(n, dtype=np.float32)
    feed = "".join(f"{int(xs[i])} {float(invs[i]).hex()} {ls[i,0]} {ls[i,1]} {ls[i,2]} {ls[i,3]} {float(xa[i]).hex()} {float(ya[i]).hex()}\n"
                   for i in range(n))
    L = oracle.lib()
    L.orc_probe_tf.argtypes = [ctypes.c_int, ctypes.c_float]; L.orc_probe_tf.restype = ctypes.c_float
    L.orc_probe_blend.argtypes = [ctypes.c_int] * 4 + [ctypes.c_float] * 2; L.orc_probe_blend.restype = ctypes.c_float
    bits = lambda f: struct.unpack("<I", struct.pack("<f", f))[0]
    have_fma = " fma " in open("/proc/cpuinfo").read()
    builds = [(0, ["-ffp-contract=off"])] + ([(1, ["-mfma", "-ffp-contract=fast"])] if have_fma else [])
    differs = 0
    for mode, flags in builds:
        exe = tmp_path / f"probe{mode}"
        subprocess.run(["g++", "-O2", *flags, str(src), "-o", str(exe)], check=True)
        out = subprocess.run([str(exe)], input=feed, capture_output=True, text=True, check=True).stdout.split()
        assert len(out) == 2 * n
        old = oracle.set_fp_contract(bool(mode))
        try:
            for i in range(n):
                want_t, want_r = int(out[2 * i], 16), int(out[2 * i + 1], 16)
                assert bits(L.orc_probe_tf(int(xs[i]), float(invs[i]))) == want_t, (mode, i)
                got_r = bits(L.orc_probe_blend(int(ls[i, 0]), int(ls[i, 1]), int(ls[i, 2]), int(ls[i, 3]), float(xa[i]), float(ya[i])))
                assert got_r == want_r, (mode, i)
        finally:
            oracle.set_fp_contract(old)
    if have_fma:                                   # and the two modes really are different functions
        oracle.set_fp_contract(False)
        a = [bits(L.orc_probe_blend(int(ls[i, 0]), int(ls[i, 1]), int(ls[i, 2]), int(ls[i, 3]), float(xa[i]), float(ya[i]))) for i in range(n)]
        oracle.set_fp_contract(True)
        b = [bits(L.orc_probe_blend(int(ls[i, 0]), int(ls[i, 1]), int(ls[i, 2]), int(ls[i, 3]), float(xa[i]), float(ya[i]))) for i in range(n)]
        oracle.set_fp_contract(False)
        assert sum(x != y for x, y in zip(a, b)) > 0


def test_c_vs_numpy_clahe_fp_contract():
    """Mode 1 of the CLAHE blend (GCC FMA contraction) in the two independent restatements: explicit fmaf() in C against numpy with an
    emulated, correctly rounded float32 fma (round-to-odd in binary64) -- whole images, odd shapes included, 8- and 16-bit."""
    import ctypes
    from oracle.np_oracle import _fma32
    m = ctypes.CDLL("libm.so.6")
    m.fmaf.argtypes = [ctypes.c_float] * 3; m.fmaf.restype = ctypes.c_float
    rng = np.random.default_rng(3)
    a = (rng.random(5000, dtype=np.float32) * 255).astype(np.float32)
    b = rng.random(5000, dtype=np.float32)
    c = (rng.random(5000, dtype=np.float32) * 255 * rng.choice([-1, 1], 5000)).astype(np.float32)
    want = np.array([m.fmaf(float(x), float(y), float(z)) for x, y, z in zip(a, b, c)], np.float32)
    assert np.array_equal(_fma32(a, b, c).view(np.int32), want.view(np.int32))
    old = oracle.set_fp_contract(True)
    try:
        diff = 0
        for (w, h), cfg in [((64, 48), (2.0, 8, 8)), ((63, 47), (3.0, 4, 4)), ((640, 360), (2.0, 8, 8)), ((481, 271), (40.0, 16, 2))]:
            for dist in ("D1", "D2"):
                y = synth.y_plane(w, h, dist, 21)
                got = oracle.clahe(y, *cfg)
                assert np.array_equal(got, oracle.np_clahe(y, *cfg, fp_contract=True)), ((w, h), cfg, dist)
                diff += int((got != oracle.np_clahe(y, *cfg)).sum())
        assert diff > 0                                   # the modes are not the same function
        s16 = rng.integers(0, 65536, (90, 160), dtype=np.uint16)
        assert np.array_equal(oracle.clahe16(s16, 2.0, 8, 8), oracle.np_clahe(s16, 2.0, 8, 8, fp_contract=True))
    finally:
        oracle.set_fp_contract(old)


def test_np_analyze_diff_matches_the_reference_check():
    """1frameMeasure.cpp:91-100: absdiff, then analyzeDiff(diff, 1, err_per) -- pixels whose difference EXCEEDS 1, as a percentage."""
    a = np.array([[10, 20, 30, 40], [50, 60, 70, 80]], np.uint8)
    b = np.array([[10, 21, 28, 40], [55, 60, 70, 0]], np.uint8)
    r = oracle.np_analyze_diff(a, b, 1)
    assert r["diff"].tolist() == [[0, 1, 2, 0], [5, 0, 0, 80]]
    assert (r["above"], r["max_diff"], r["min_diff"], r["total"], r["err_per"]) == (3, 80, 0, 8, 37.5)
    assert oracle.np_analyze_diff(a, a, 0)["err_per"] == 0.0
    assert oracle.np_analyze_diff(r["diff"], None, 1)["above"] == 3
