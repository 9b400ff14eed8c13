"""GPU parity tests (run with -m gpu on an MI355X): HIP path through the C ABI vs the CPU oracle,
bit-exact (integer/byte work; the float steps are reproduced operation by operation)."""
import numpy as np
import pytest

import mi_lumaeq
import oracle
from mi_lumaeq import synth, xfer

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

SMALL = [(1, 1), (1, 17), (3, 4097), (47, 63), (48, 64), (15, 16), (135, 241), (270, 480), (360, 640)]


def dev(a):
    """host array -> device tensor WITHOUT handing pageable memory to the runtime (mi_lumaeq.xfer says why)"""
    return xfer.to_device(a)


host = xfer.to_host                                     # device tensor -> numpy, likewise through pinned staging


def hooks_ctx():
    """A context of libmi_lumaeq_test.so: the product sources built with -DMI_TEST_HOOKS (fault injection into the fused kernel's
    hand-off, microsecond wait bounds, forced failures of checked HIP calls).  The product library knows none of these options."""
    c = mi_lumaeq.Context(0, lib=mi_lumaeq.test_lib())
    c.set_option("two_kernel_max_frames", 0)      # these tests are about the fused kernel's hand-off: one- and two-frame calls must take it too
    return c


@pytest.mark.parametrize("shape", SMALL, ids=str)
@pytest.mark.parametrize("dist", synth.DISTS)
def test_equalize_host_form(ctx, shape, dist):
    h, w = shape
    src = synth.y_plane(w, h, dist, 21)
    assert np.array_equal(ctx.equalize_hist(src), oracle.equalize_hist(src))


def test_equalize_kats(ctx):
    assert ctx.equalize_hist(np.array([[50, 50, 100, 200]], np.uint8)).tolist() == [[0, 0, 128, 255]]
    a = np.array([0] + [1] * 253 + [2] * 2 + [3] * 255, np.uint8)[None, :]
    o = ctx.equalize_hist(a)
    assert (o[0, 0], o[0, 1], o[0, 254], o[0, 256]) == (0, 126, 128, 255)      # ties to even
    assert (ctx.equalize_hist(np.full((9, 13), 77, np.uint8)) == 77).all()
    assert ctx.equalize_hist(np.array([[10, 10, 10, 20]], np.uint8)).tolist() == [[0, 0, 0, 255]]
    assert ctx.equalize_hist(np.arange(256, dtype=np.uint8)[None, :]).reshape(-1).tolist() == list(range(256))


def test_equalize_empty_and_errors(ctx):
    e = np.empty((0, 0), np.uint8)
    assert ctx.equalize_hist(e).shape == (0, 0)
    with pytest.raises(mi_lumaeq.MiError):
        ctx.equalize_hist(np.zeros((4, 4), np.float32))
    with pytest.raises(mi_lumaeq.MiError):
        ctx.clahe(np.zeros((4, 4), np.uint8), 2.0, 0, 8)


def test_equalize_strided_roi_and_inplace(ctx):
    big = synth.y_plane(300, 200, "D2", 5)
    view = big[10:190, 37:291]                      # step 300 > width 254, unaligned start
    want = oracle.equalize_hist(view)
    assert np.array_equal(ctx.equalize_hist(view), want)
    out = np.zeros((200, 320), np.uint8)
    oview = out[3:183, 5:259]
    ctx.equalize_hist(view, oview)                 # caller-owned dst written in place (nextimprovement.cpp:164-168)
    assert np.array_equal(oview, want) and out[:3].sum() == 0 and out[:, :5].sum() == 0
    buf = view.copy()
    ctx.equalize_hist(buf, buf)
    assert np.array_equal(buf, want)


@pytest.mark.parametrize("pitch", [320, 336, 300, 4096], ids=str)
def test_equalize_roi_views_by_pitch(ctx, pitch):
    """ROI views by row pitch: a pitch that is a multiple of 16 takes the band walk of the three-kernel path (all rows share one alignment
    phase: vectors over (row, slot) items, head / tail bytes apart), any other pitch the row-by-row walk; origins at every alignment,
    widths shorter than a vector, in place, and a device batch with separate source / destination pitches."""
    rng = np.random.default_rng(pitch)
    rows = 97
    big = rng.integers(0, 256, (rows, pitch), dtype=np.uint8)
    big[:, : pitch // 2] //= 2
    for x0, wv in ((0, min(pitch, 256)), (1, 254), (15, 33), (16, 17), (21, min(pitch - 21, 277)), (5, 9), (pitch - 20, 20)):
        view = big[3:rows - 2, x0:x0 + wv]
        want = oracle.equalize_hist(view)
        assert np.array_equal(ctx.equalize_hist(view), want), (pitch, x0, wv)
        buf = big.copy()
        bview = buf[3:rows - 2, x0:x0 + wv]
        ctx.equalize_hist(bview, bview)                                  # in place, strided
        assert np.array_equal(bview, want), (pitch, x0, wv)
        untouched = np.ones_like(buf, dtype=bool)
        untouched[3:rows - 2, x0:x0 + wv] = False
        assert np.array_equal(buf[untouched], big[untouched]), (pitch, x0, wv)
    # device batch: source pitch `pitch`, destination pitch pitch + 16 (same phase class) and pitch + 5 (not)
    n, w, h = 5, min(pitch - 29, 501), 61
    src = rng.integers(0, 200, (n, h + 6, pitch), dtype=np.uint8)
    d_src = dev(src)
    for dpitch in (pitch + 16, pitch + 5):
        d_dst = torch.zeros((n, h + 6, dpitch), dtype=torch.uint8, device="cuda:0")
        so, do = 2 * pitch + 29, 3 * dpitch + 7
        ctx.equalize_hist_batch_dev(d_src.data_ptr() + so, d_dst.data_ptr() + do, w, h, n, src_step=pitch, src_frame=(h + 6) * pitch,
                                    dst_step=dpitch, dst_frame=(h + 6) * dpitch)
        ctx.synchronize()
        out = host(d_dst)
        for k in range(n):
            want = oracle.equalize_hist(src[k, 2:2 + h, 29:29 + w])
            assert np.array_equal(out[k, 3:3 + h, 7:7 + w], want), (pitch, dpitch, k)
            assert out[k, :3].sum() == 0 and out[k, 3:3 + h, :7].sum() == 0 and out[k, 3:3 + h, 7 + w:].sum() == 0


@pytest.mark.parametrize("wh", [(1920, 1080), (3840, 2160)], ids=str)
@pytest.mark.parametrize("dist", synth.DISTS)
def test_equalize_full_size_frames(ctx, wh, dist):
    """BASELINE.json configs[0]/[1]: one 1080p / 4K frame, bit-exact vs the oracle."""
    w, h = wh
    src = synth.y_plane(w, h, dist, 1)
    assert np.array_equal(ctx.equalize_hist(src), oracle.equalize_hist(src))


# ---- the reference's own check as an operator: cv::absdiff + xf::cv::analyzeDiff (1frameMeasure.cpp:91-100) -------------------

@pytest.mark.parametrize("shape", SMALL + [(1080, 1920)], ids=str)
@pytest.mark.parametrize("threshold", [0, 1, 7])
def test_analyze_diff_host_form(ctx, shape, threshold):
    h, w = shape
    a = synth.y_plane(w, h, "D1", 61)
    b = a.copy()
    rng = np.random.default_rng(w * 131 + h)
    k = max(1, a.size // 9)
    idx = rng.choice(a.size, size=k, replace=False)
    b.reshape(-1)[idx] = (b.reshape(-1)[idx].astype(np.int16) + rng.integers(-9, 10, size=k)).clip(0, 255).astype(np.uint8)
    want = oracle.np_analyze_diff(a, b, threshold)
    got = ctx.analyze_diff(a, b, threshold, want_diff=True)
    assert np.array_equal(got.pop("diff"), want.pop("diff"))
    assert got == want
    # analyzeDiff on its own: the difference image in, same statistics out
    alone = ctx.analyze_diff(oracle.np_analyze_diff(a, b)["diff"], None, threshold)
    assert alone == {k_: v for k_, v in want.items()}
    same = ctx.analyze_diff(a, a, threshold)
    assert (same["above"], same["max_diff"], same["min_diff"], same["err_per"]) == (0, 0, 0, 0.0)


def test_analyze_diff_strided_views_and_errors(ctx):
    big_a, big_b = synth.y_plane(300, 200, "D2", 5), synth.y_plane(320, 210, "D1", 6)
    va, vb = big_a[10:190, 37:291], big_b[7:187, 3:257]                 # different steps, unaligned starts
    want = oracle.np_analyze_diff(va, vb, 3)
    out = np.zeros((200, 400), np.uint8)
    got = ctx.analyze_diff(va, vb, 3, want_diff=True)
    assert np.array_equal(got.pop("diff"), want.pop("diff")) and got == want
    assert ctx.analyze_diff(np.empty((0, 0), np.uint8), np.empty((0, 0), np.uint8))["total"] == 0
    with pytest.raises(mi_lumaeq.MiError):
        ctx.analyze_diff(va, vb, 256)
    with pytest.raises(mi_lumaeq.MiError):
        ctx.analyze_diff(va.astype(np.float32), vb)
    assert out.sum() == 0


def test_analyze_diff_batch_dev_is_how_full_size_batches_are_compared(ctx):
    """The reference's check at BASELINE size, on the device: 16 x 4K frames through the fused path and through the three-kernel path
    (two independent implementations) differ nowhere -- and a single planted +-1 and +-2 are found where they were put."""
    w, h, n = 3840, 2160, 16
    d_in = synth.nv12_batch_torch(w, h, n, "D2", "cuda:0", seed=77)
    fused, staged = torch.empty_like(d_in), torch.empty_like(d_in)
    ctx.equalize_hist_nv12_batch_dev(d_in, fused, w, h, n, mi_lumaeq.UV_FILL128)
    other = mi_lumaeq.Context(0)
    other.set_option("fused", 0)
    other.equalize_hist_nv12_batch_dev(d_in, staged, w, h, n, mi_lumaeq.UV_FILL128)
    other.synchronize(); ctx.synchronize()
    fb = w * h * 3 // 2
    stats = torch.full((n, 4), 0xFFFFFFFF, dtype=torch.int64, device="cuda:0").to(torch.int32)
    ctx.analyze_diff_batch_dev(fused, staged, w, h * 3 // 2, n, stats, threshold=1, a_frame=fb, b_frame=fb)
    torch.cuda.synchronize()
    s = host(stats).astype(np.int64) & 0xFFFFFFFF
    assert (s[:, 0] == 0).all() and (s[:, 1] == 0).all() and (s[:, 2] == 0).all() and (s[:, 3] == fb).all()
    staged[3, 12345] = fused[3, 12345] ^ 1                                  # within the reference's tolerance: seen by max_diff, not by `above`
    v = int(fused[9, 777])
    staged[9, 777] = v + 2 if v < 254 else v - 2                            # outside it
    diff = torch.empty_like(d_in)
    ctx.analyze_diff_batch_dev(fused, staged, w, h * 3 // 2, n, stats, threshold=1, diff=diff, a_frame=fb, b_frame=fb, diff_frame=fb)
    torch.cuda.synchronize()
    s = host(stats).astype(np.int64) & 0xFFFFFFFF
    assert s[3].tolist() == [0, 1, 0, fb] and s[9].tolist() == [1, 2, 0, fb]
    assert s[[i for i in range(n) if i not in (3, 9)], :3].sum() == 0
    assert int(diff[3, 12345]) == 1 and int(diff[9, 777]) == 2 and int(diff.sum()) == 3
    other.close()


def test_stage_apis(ctx):
    w, h, n = 640, 360, 3
    ys = np.stack([synth.y_plane(w, h, d, 30 + i) for i, d in enumerate(("D1", "D2", "D3"))])
    d_src = dev(ys)
    d_hist = torch.zeros((n, 256), dtype=torch.int32, device="cuda:0")
    ctx.hist_batch_dev(d_src, w, h, n, d_hist)
    d_lut = torch.zeros((n, 256), dtype=torch.uint8, device="cuda:0")
    ctx.equalize_lut_batch_dev(d_hist, w * h, n, d_lut)
    d_dst = torch.empty_like(d_src)
    ctx.lut_apply_batch_dev(d_src, d_dst, w, h, n, d_lut)
    torch.cuda.synchronize()
    hist, lut, dst = host(d_hist), host(d_lut), host(d_dst)
    for k in range(n):
        oh = oracle.hist(ys[k])
        assert np.array_equal(hist[k], oh)
        ol, first = oracle.equalize_lut(oh, w * h)
        assert np.array_equal(lut[k][first:], ol[first:]) or (oh[first] == w * h and (lut[k] == first).all())
        assert np.array_equal(dst[k], oracle.equalize_hist(ys[k]))


@pytest.mark.parametrize("uv_mode", [0, 1])
def test_nv12_batch_dev(ctx, uv_mode):
    w, h, n = 640, 360, 5
    frames = np.stack([synth.nv12_frame(w, h, synth.DISTS[k % 5], 40 + k) for k in range(n)])
    d_in = dev(frames)
    d_out = torch.empty_like(d_in)
    ctx.equalize_hist_nv12_batch_dev(d_in, d_out, w, h, n, uv_mode, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    out = host(d_out)
    for k in range(n):
        assert np.array_equal(out[k], oracle.nv12_frame(frames[k], w, h, uv_mode=uv_mode, op=0)), k
    # in place
    ctx.equalize_hist_nv12_batch_dev(d_in, d_in, w, h, n, uv_mode, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(host(d_in), out)


def test_nv12_host_form(ctx):
    w, h = 322, 178                                 # odd-ish, unaligned UV start
    f = synth.nv12_frame(w, h, "D2", 77)
    for uv_mode in (0, 1):
        assert np.array_equal(ctx.equalize_hist_nv12(f, w, h, uv_mode), oracle.nv12_frame(f, w, h, uv_mode=uv_mode, op=0))
        assert np.array_equal(ctx.clahe_nv12(f, w, h, uv_mode, 2.0, 8, 8),
                              oracle.nv12_frame(f, w, h, uv_mode=uv_mode, op=1, clip_limit=2.0, tiles_x=8, tiles_y=8))


CLAHE_CFG = [(2.0, 8, 8), (3.0, 4, 4), (40.0, 8, 8), (0.0, 3, 5), (1.5, 1, 1), (2.0, 16, 2), (2.0, 70, 3)]


@pytest.mark.parametrize("shape", SMALL, ids=str)
@pytest.mark.parametrize("cfg", CLAHE_CFG, ids=str)
def test_clahe_host_form(ctx, shape, cfg):
    h, w = shape
    clip, tx, ty = cfg
    for dist in ("D1", "D2", "D3"):
        src = synth.y_plane(w, h, dist, 22)
        assert np.array_equal(ctx.clahe(src, clip, tx, ty), oracle.clahe(src, clip, tx, ty)), dist


@pytest.mark.parametrize("case", [(3840, 2160, 16, 16), (3840, 2160, 32, 8), (3840, 2160, 60, 4), (1920, 1080, 24, 5), (1919, 1079, 20, 7),
                                  (1919, 1079, 16, 7), (4097, 64, 48, 2), (4097, 64, 20, 2), (1280, 720, 63, 3)], ids=str)
def test_clahe_wide_tile_grids(ctx, case):
    """Grids of more than 14 tiles across: the float pair tables are staged per COLUMN SEGMENT (each segment only the pairs its
    columns use), down to tiles of ~20 pixels; narrower tiles take the uchar-quad tables.  Host form and batch form, both table
    kinds, against the oracle."""
    w, h, tx, ty = case
    src = synth.y_plane(w, h, "D2", 41)
    want = oracle.clahe(src, 2.0, tx, ty)
    try:
        for ft, pairs in ((1, 9), (1, 4), (1, 15), (0, 9)):
            ctx.set_option("clahe_float_tables", ft)
            ctx.set_option("clahe_seg_pairs", pairs)               # pairs per segment table (speed only)
            assert np.array_equal(ctx.clahe(src, 2.0, tx, ty), want), (case, ft, pairs)
        ctx.set_option("clahe_float_tables", 1)
        ctx.set_option("clahe_seg_pairs", 9)
        if w % 2 == 0 and h % 2 == 0:
            frames = np.stack([synth.nv12_frame(w, h, synth.DISTS[k], 300 + k) for k in range(3)])
            d_in = dev(frames)
            d_out = torch.zeros_like(d_in)
            ctx.clahe_nv12_batch_dev(d_in, d_out, w, h, 3, mi_lumaeq.UV_COPY, 3.0, tx, ty)
            ctx.synchronize()
            out = host(d_out)
            for k in range(3):
                assert np.array_equal(out[k], oracle.nv12_frame(frames[k], w, h, uv_mode=1, op=1, clip_limit=3.0, tiles_x=tx, tiles_y=ty)), (case, k)
    finally:
        ctx.set_option("clahe_float_tables", 1)
        ctx.set_option("clahe_seg_pairs", 9)
    with pytest.raises(mi_lumaeq.MiError):
        ctx.set_option("clahe_seg_pairs", 16)


def test_clahe_kats(ctx):
    assert (ctx.clahe(np.full((4, 4), 7, np.uint8), 2.0, 1, 1) == 32).all()             # CL-1
    rng = np.random.default_rng(5)
    src = rng.integers(0, 256, (37, 53), dtype=np.uint8)
    lut = oracle.clahe_tile_luts(src, 4.0, 1, 1)[0]
    assert np.array_equal(ctx.clahe(src, 4.0, 1, 1), lut[src])                           # CL-2
    import json
    from pathlib import Path
    kats = {k["id"]: k for k in json.loads((Path(__file__).parent / "golden" / "kat.json").read_text())["clahe"]}
    k = kats["CL-4"]                                                                     # interpolation weights on step-function LUTs
    q = np.empty((8, 8), np.uint8)
    q[:4, :4], q[:4, 4:], q[4:, :4], q[4:, 4:] = k["src_quadrants"]
    assert ctx.clahe(q, k["clip"], *k["tiles"]).reshape(-1).tolist() == k["dst"]
    k = kats["CL-5"]                                                                     # ties to even inside CLAHE's LUT
    a = np.concatenate([np.full(n, v, np.uint8) for v, n in k["src_runs"]])[None, :]
    o = ctx.clahe(a, k["clip"], *k["tiles"])
    assert (o[a == 1] == 126).all() and (o[a == 2] == 255).all()


def test_derived_kats_one_per_quirk(ctx):
    """The round-5 known answers (kat.json "derived": one per quirk of SURVEY App. A, answers derived in exact arithmetic by
    tests/golden/derive_kats.py, each killing a named mutant in tests/test_kat_kill_matrix.py) through the HIP kernels: host form and
    a device-resident NV12-style batch of the same plane.  With clahe_fp_contract = 1 (GCC's FMA contraction, an OPTION) the two
    FMA known answers must FAIL -- the kernels' default arithmetic is the uncontracted one."""
    import json
    from pathlib import Path
    kats = json.loads((Path(__file__).parent / "golden" / "kat.json").read_text())["derived"]
    assert len(kats) >= 15
    for k in kats:
        h, w = k["shape"]
        src = np.array(k["src"], np.uint8).reshape(h, w)
        want = np.array(k["dst"], np.uint8).reshape(h, w)
        got = ctx.equalize_hist(src) if k["op"] == "equalize" else ctx.clahe(src, k["clip"], *k["tiles"])
        assert np.array_equal(got, want), (k["id"], got.reshape(-1).tolist(), k["dst"])
        # the batched device form: three copies of the plane, every one must come out the same
        d_in = dev(np.stack([src] * 3))
        d_out = torch.zeros_like(d_in)
        if k["op"] == "equalize":
            ctx.equalize_hist_batch_dev(d_in, d_out, w, h, 3)
        else:
            ctx.clahe_batch_dev(d_in, d_out, w, h, 3, k["clip"], *k["tiles"])
        torch.cuda.synchronize()
        assert all(np.array_equal(o, want) for o in host(d_out)), k["id"]
    ctx.set_option("clahe_fp_contract", 1)
    try:
        flipped = []
        for k in kats:
            if k["op"] == "clahe":
                h, w = k["shape"]
                src = np.array(k["src"], np.uint8).reshape(h, w)
                if ctx.clahe(src, k["clip"], *k["tiles"]).reshape(-1).tolist() != k["dst"]:
                    flipped.append(k["guards"][0])
        assert "cl_blend_fma" in flipped and "cl_coord_fma" in flipped, flipped
    finally:
        ctx.set_option("clahe_fp_contract", 0)


def test_clahe_tile_luts_stage(ctx):
    for (w, h, tx, ty, clip) in [(640, 360, 8, 8, 2.0), (16, 15, 8, 8, 2.0), (241, 135, 4, 4, 3.0), (1919, 1079, 4, 4, 3.0)]:
        src = synth.y_plane(w, h, "D2", 50)
        d_luts = torch.zeros((tx * ty, 256), dtype=torch.uint8, device="cuda:0")
        ctx.clahe_tile_luts_batch_dev(dev(src), w, h, 1, clip, tx, ty, d_luts)
        torch.cuda.synchronize()
        assert np.array_equal(host(d_luts), oracle.clahe_tile_luts(src, clip, tx, ty)), (w, h, tx, ty)


@pytest.mark.parametrize("wh,cfg", [((3840, 2160), (2.0, 8, 8)), ((1920, 1080), (2.0, 8, 8)), ((1919, 1079), (3.0, 4, 4)),
                                    ((1280, 720), (2.0, 8, 8))], ids=str)
def test_clahe_full_size(ctx, wh, cfg):
    """BASELINE.json configs[2] (4K, 8x8, clip 2.0) and the odd hun.png-shaped case of clahe1frame.cpp."""
    w, h = wh
    clip, tx, ty = cfg
    for dist in ("D1", "D2"):
        src = synth.y_plane(w, h, dist, 2)
        assert np.array_equal(ctx.clahe(src, clip, tx, ty), oracle.clahe(src, clip, tx, ty)), dist


def test_clahe_strided_and_batch(ctx):
    big = synth.y_plane(400, 300, "D2", 6)
    view = big[7:287, 21:389]
    want = oracle.clahe(view, 2.0, 8, 8)
    assert np.array_equal(ctx.clahe(view, 2.0, 8, 8), want)
    w, h, n = 480, 270, 4
    frames = np.stack([synth.nv12_frame(w, h, synth.DISTS[k % 3], 60 + k) for k in range(n)])
    d_in = dev(frames)
    d_out = torch.empty_like(d_in)
    ctx.clahe_nv12_batch_dev(d_in, d_out, w, h, n, 1, 2.0, 8, 8, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    out = host(d_out)
    for k in range(n):
        assert np.array_equal(out[k], oracle.nv12_frame(frames[k], w, h, uv_mode=1, op=1, clip_limit=2.0, tiles_x=8, tiles_y=8)), k


def test_full_size_batch_properties(ctx):
    """At BASELINE size (4K batch) check size-independent properties instead of the oracle on every
    frame: a histogram of the output is the LUT-image of the input histogram; constant frames are
    fixed points; the UV plane is untouched / 128; one frame is compared against the oracle."""
    w, h, n = 3840, 2160, 8
    d_in = synth.nv12_batch_torch(w, h, n, "D2", "cuda:0", seed=123)
    d_in[3, : w * h] = 128
    d_out = torch.empty_like(d_in)
    # frames were generated asynchronously on torch's stream: launch on that stream (the context's own
    # stream is non-blocking and does not order against it)
    ctx.equalize_hist_nv12_batch_dev(d_in, d_out, w, h, n, mi_lumaeq.UV_COPY, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(d_in[:, w * h:], d_out[:, w * h:])
    assert (d_out[3, : w * h] == 128).all()
    for k in (0, 5):
        yi, yo = d_in[k, : w * h], d_out[k, : w * h]
        hi = torch.bincount(yi.to(torch.int64), minlength=256)
        first = int(torch.nonzero(hi)[0])
        assert int(yo.min()) == 0 and int(yo.max()) == 255
        # monotone: sorting pixels by input value sorts them by output value
        lut = torch.zeros(256, dtype=torch.int64, device="cuda:0")
        lut[yi.to(torch.int64)] = yo.to(torch.int64)
        nzv = torch.nonzero(hi).view(-1)
        assert (torch.diff(lut[nzv]) >= 0).all() and int(lut[first]) == 0
    k = 5
    assert np.array_equal(host(d_out[k]), oracle.nv12_frame(host(d_in[k]), w, h, uv_mode=1, op=0))


def test_profiling_counters(ctx):
    w, h, n = 640, 360, 2
    d_in = dev(np.stack([synth.nv12_frame(w, h, "D1", k) for k in range(n)]))
    d_out = torch.empty_like(d_in)
    try:
        ctx.set_option("two_kernel_max_frames", 0)                  # (a two-frame call would otherwise take the two-kernel path, checked last)
        for fused, names in ((1, ["equalize_fused_kernel", "fused_finish_kernel"]), (0, ["hist_partial_kernel", "equalize_lut_kernel", "lut_apply_kernel"])):
            ctx.set_option("fused", fused)
            ctx.profile_read(reset=True)
            ctx.set_profiling(True)
            ctx.equalize_hist_nv12_batch_dev(d_in, d_out, w, h, n, 0)
            p = ctx.profile_read(reset=True)
            ctx.set_profiling(False)
            for k, v in p.items():
                assert v["launches"] == (1 if k in names else 0), (fused, k)
                assert (v["total_ms"] > 0) == (k in names)
        # distribution of per-launch durations: 12 launches -> ordered percentiles that bracket the mean
        ctx.set_option("fused", 1)
        ctx.set_profiling(True)
        for _ in range(12):
            ctx.equalize_hist_nv12_batch_dev(d_in, d_out, w, h, n, 0)
        v = ctx.profile_read(reset=True)["equalize_fused_kernel"]
        assert v["launches"] == 12
        assert 0 < v["min_ms"] <= v["p10_ms"] <= v["p50_ms"] <= v["p90_ms"] <= v["max_ms"]
        assert v["min_ms"] <= v["total_ms"] / 12 <= v["max_ms"]
        assert ctx.profile_read(reset=True)["equalize_fused_kernel"]["max_ms"] == 0          # reset clears the samples
        # default routing of a one- or two-frame call: histogram + LUT in ONE launch (its last workgroup writes the LUT), then the apply kernel
        ctx.set_option("two_kernel_max_frames", 8)
        ctx.set_profiling(True)
        ctx.equalize_hist_nv12_batch_dev(d_in, d_out, w, h, n, 0)
        p = ctx.profile_read(reset=True)
        ctx.set_profiling(False)
        assert {k for k, v in p.items() if v["launches"]} == {"hist_partial_kernel", "lut_apply_kernel"}
        assert p["hist_partial_kernel"]["launches"] == 1 and p["lut_apply_kernel"]["launches"] == 1
        src, out = host(d_in), host(d_out)
        for k in range(n):
            assert np.array_equal(out[k], oracle.nv12_frame(src[k], w, h, uv_mode=0, op=0))
    finally:
        ctx.set_option("fused", 1)
        ctx.set_option("two_kernel_max_frames", 8)
        ctx.set_profiling(False)


@pytest.mark.parametrize("opts", [dict(fused=0), dict(fused_vpt=8), dict(fused_vpt=16), dict(fused_vpt=20, fused_wgs_per_cu=2),
                                  dict(fused_vpt=24, fused_acquire=0), dict(fused_wgs_per_cu=1), dict(two_kernel_max_frames=64),
                                  dict(two_kernel_max_frames=0), dict(two_kernel_max_frames=0, fused=0)], ids=str)
def test_equalize_paths_agree(ctx, opts):
    """Three-kernel path and every fused-kernel configuration give the oracle's bytes (4K, 1080p, tiny, UV modes)."""
    try:
        if "two_kernel_max_frames" not in opts:
            ctx.set_option("two_kernel_max_frames", 0)              # these configurations are about the fused / three-kernel paths: keep small batches on them
        for k, v in opts.items():
            ctx.set_option(k, v)
        for (w, h, n) in [(3840, 2160, 3), (1920, 1080, 5), (64, 36, 7), (16, 1, 2)]:
            frames = np.stack([synth.nv12_frame(w, h, synth.DISTS[k % 5], 70 + k) for k in range(n)])
            d_in = dev(frames)
            for uv_mode in (0, 1):
                d_out = torch.zeros_like(d_in)
                ctx.equalize_hist_nv12_batch_dev(d_in, d_out, w, h, n, uv_mode)
                ctx.synchronize()
                out = host(d_out)
                for k in range(n):
                    assert np.array_equal(out[k], oracle.nv12_frame(frames[k], w, h, uv_mode=uv_mode, op=0)), (w, h, k, uv_mode)
    finally:
        for k, v in dict(fused=1, fused_vpt=20, fused_wgs_per_cu=4, fused_acquire=1, two_kernel_max_frames=8).items():
            ctx.set_option(k, v)


def test_fused_stress_many_frames(ctx):
    """Hand-off protocol under load: 256 1080p frames in one launch, twice, checked frame by frame on the GPU
    against the three-kernel path (itself oracle-checked above)."""
    w, h, n = 1920, 1080, 256
    d_in = synth.nv12_batch_torch(w, h, n, "D2", "cuda:0", seed=77)
    d_a, d_b = torch.empty_like(d_in), torch.empty_like(d_in)
    try:
        ctx.set_option("fused", 0)
        ctx.equalize_hist_nv12_batch_dev(d_in, d_a, w, h, n, 1)
        ctx.set_option("fused", 1)
        for _ in range(3):
            d_b.zero_()
            ctx.equalize_hist_nv12_batch_dev(d_in, d_b, w, h, n, 1)
            ctx.synchronize()
            assert torch.equal(d_a, d_b)
    finally:
        ctx.set_option("fused", 1)


def test_large_and_degenerate_shapes(ctx):
    """Maximum / ragged sizes: an 8K plane (more slices than CUs -> three-kernel fallback), a frame taller than
    65535 rows with a stride, width 1, height 1, non-multiple-of-16 sizes, zero frames."""
    rng = np.random.default_rng(17)
    big = rng.integers(0, 256, (4320, 7680), dtype=np.uint8)
    big[1000:3000] = np.clip(big[1000:3000] // 4 + 90, 0, 255)
    assert np.array_equal(ctx.equalize_hist(big), oracle.equalize_hist(big))
    tall = rng.integers(0, 256, (70000, 24), dtype=np.uint8)
    view = tall[:, 3:20]                                         # 70000 strided rows of 17 bytes
    assert np.array_equal(ctx.equalize_hist(view), oracle.equalize_hist(view))
    assert np.array_equal(ctx.clahe(view, 2.0, 2, 64), oracle.clahe(view, 2.0, 2, 64))
    for shape in [(1, 70001), (70001, 1), (2, 8), (1, 16), (17, 31)]:
        a = rng.integers(0, 256, shape, dtype=np.uint8)
        assert np.array_equal(ctx.equalize_hist(a), oracle.equalize_hist(a)), shape
        assert np.array_equal(ctx.clahe(a, 2.0, 8, 8), oracle.clahe(a, 2.0, 8, 8)), shape
    d = torch.zeros(64, dtype=torch.uint8, device="cuda:0")
    ctx.equalize_hist_nv12_batch_dev(d, d, 8, 4, 0, 0)           # zero frames: no-op
    ctx.equalize_hist_nv12_batch_dev(d, d, 0, 4, 1, 0)           # empty frame: no-op
    ctx.synchronize()


def test_maximum_size_frame(ctx):
    """The largest frame the interface admits: W*H just below 2^31 (OpenCV's `int total`), 2 GiB of pixels.  Every index in the host
    form and the three-kernel path has to survive it, the counts reach 2^23 per bin and the LUT scale is 255 / (2^31 - ...) -- bit-exact
    against the oracle, checked by CRC and on both ends of the plane."""
    import zlib
    side = 46336                                                    # 46336^2 = 2 147 024 896 < 2^31, a multiple of 16
    rng = np.random.default_rng(99)
    row = rng.integers(0, 256, (1024, side), dtype=np.uint8)
    src = np.empty((side, side), np.uint8)
    for y0 in range(0, side, 1024):                                 # 2 GiB of varied content without 2 GiB of random numbers
        n = min(1024, side - y0)
        np.add(row[:n], np.uint8((y0 // 1024) * 5), out=src[y0:y0 + n])      # wraps modulo 256
    src[: side // 3] //= 3                                          # a dark third: the histogram is far from flat
    want = oracle.equalize_hist(src)
    got = ctx.equalize_hist(src)
    assert np.array_equal(got[:64], want[:64]) and np.array_equal(got[-64:], want[-64:])
    assert zlib.crc32(got) == zlib.crc32(want)
    del got, want
    # CLAHE 8 x 8 on the same plane: tiles of 5792 x 5792 pixels (33.5 M each: counts and clip limit far above 2^16)
    want = oracle.clahe(src, 2.0, 8, 8)
    got = ctx.clahe(src, 2.0, 8, 8)
    assert np.array_equal(got[:64], want[:64]) and np.array_equal(got[-64:], want[-64:])
    assert zlib.crc32(got) == zlib.crc32(want)
    del got, want, src                                              # (one pixel more per side is refused: test_argument_errors)


def test_argument_errors(ctx):
    d = torch.zeros(1024, dtype=torch.uint8, device="cuda:0")
    with pytest.raises(mi_lumaeq.MiError) as e:
        ctx.equalize_hist_nv12_batch_dev(d, d, 8, 4, 1, 7)       # bad uv_mode
    assert e.value.status == 1
    with pytest.raises(mi_lumaeq.MiError) as e:
        ctx.equalize_hist_batch_dev(d, d, 16, 4, 1, src_step=8)  # step < width
    assert e.value.status == 1
    with pytest.raises(mi_lumaeq.MiError) as e:
        ctx.equalize_hist_batch_dev(d, d, 1 << 16, 1 << 16, 1)   # W*H >= 2^31 (OpenCV: int total)
    assert e.value.status == 2
    with pytest.raises(mi_lumaeq.MiError) as e:
        ctx.clahe_batch_dev(d, d, 16, 4, 1, 2.0, 0, 8)           # tile grid 0
    assert e.value.status == 1
    with pytest.raises(mi_lumaeq.MiError):
        ctx.set_option("no_such_option", 1)
    with pytest.raises(mi_lumaeq.MiError) as e:
        mi_lumaeq.Context(99)                                    # device out of range
    assert e.value.status == 5


def test_unaligned_device_batches(ctx):
    """Device batches whose planes are not 16-B aligned / not multiples of 16 take the generic path."""
    w, h, n = 333, 77, 4
    frames = np.stack([synth.nv12_frame(w, h, synth.DISTS[k % 5], 90 + k) for k in range(n)])
    fb = frames.shape[1]
    pad = torch.zeros(n * fb + 5, dtype=torch.uint8, device="cuda:0")
    d_in = pad[3:3 + n * fb]                                      # misaligned base
    d_in.copy_(dev(frames.reshape(-1)))
    d_out = torch.zeros(n * fb + 7, dtype=torch.uint8, device="cuda:0")[7:]
    for uv_mode in (0, 1):
        ctx.equalize_hist_nv12_batch_dev(d_in, d_out, w, h, n, uv_mode)
        ctx.synchronize()
        out = host(d_out).reshape(n, fb)
        for k in range(n):
            assert np.array_equal(out[k], oracle.nv12_frame(frames[k], w, h, uv_mode=uv_mode, op=0)), (k, uv_mode)
        ctx.clahe_nv12_batch_dev(d_in, d_out, w, h, n, uv_mode, 2.0, 8, 8)
        ctx.synchronize()
        out = host(d_out).reshape(n, fb)
        for k in range(n):
            assert np.array_equal(out[k], oracle.nv12_frame(frames[k], w, h, uv_mode=uv_mode, op=1, clip_limit=2.0, tiles_x=8, tiles_y=8)), (k, uv_mode)


def _bgr(w, h, seed):
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    img[: h // 2] = (img[: h // 2] // 3 + 60).astype(np.uint8)      # a low-contrast half
    return img


@pytest.mark.parametrize("shape", [(1, 1), (3, 5), (47, 63), (48, 64), (270, 480), (1079, 1919), (1080, 1920)], ids=str)
def test_color_neighbours(ctx, shape):
    """cvtColor BGR2YUV / YUV2BGR and the whole singlecolor.cpp / clahe1frame.cpp sequence."""
    h, w = shape
    a = _bgr(w, h, 5)
    yuv = ctx.cvt_color(a, mi_lumaeq.COLOR_BGR2YUV)
    assert np.array_equal(yuv, oracle.bgr2yuv(a))
    assert np.array_equal(ctx.cvt_color(yuv, mi_lumaeq.COLOR_YUV2BGR), oracle.yuv2bgr(yuv))
    assert np.array_equal(ctx.bgr_luma_op(a, mi_lumaeq.OP_EQUALIZE), oracle.bgr_luma_op(a, 0))
    assert np.array_equal(ctx.bgr_luma_op(a, mi_lumaeq.OP_CLAHE, 3.0, 4, 4), oracle.bgr_luma_op(a, 1, 3.0, 4, 4))
    inplace = a.copy()
    ctx.cvt_color(inplace, mi_lumaeq.COLOR_BGR2YUV, inplace)
    assert np.array_equal(inplace, yuv)


def test_color_strided_and_batch(ctx):
    big = _bgr(400, 300, 9)
    view = big[11:289, 7:391]                                    # row pitch 1200 > 3*384, unaligned start
    assert np.array_equal(ctx.cvt_color(view, mi_lumaeq.COLOR_BGR2YUV), oracle.bgr2yuv(view))
    assert np.array_equal(ctx.bgr_luma_op(view, mi_lumaeq.OP_EQUALIZE), oracle.bgr_luma_op(np.ascontiguousarray(view), 0))
    w, h, n = 640, 360, 3
    frames = np.stack([_bgr(w, h, 20 + k) for k in range(n)])
    d_in = dev(frames)
    d_out = torch.empty_like(d_in)
    ctx.bgr_luma_op_batch_dev(d_in, d_out, w, h, n, mi_lumaeq.OP_EQUALIZE)
    ctx.synchronize()
    out = host(d_out)
    for k in range(n):
        assert np.array_equal(out[k], oracle.bgr_luma_op(frames[k], 0)), k
    ctx.cvt_color_batch_dev(d_in, d_out, w, h, n, mi_lumaeq.COLOR_BGR2YUV)
    ctx.synchronize()
    out = host(d_out)
    for k in range(n):
        assert np.array_equal(out[k], oracle.bgr2yuv(frames[k])), k
    with pytest.raises(mi_lumaeq.MiError):
        ctx.cvt_color(frames[0], 4)                              # unsupported conversion code


def test_randomized_differential(ctx):
    """Seeded random differential test, HIP vs oracle: random shapes (1..260), row pitches, ROI offsets, in-place,
    tile grids (1..20), clip limits, distributions -- 160 cases through the host forms, plus random device batches."""
    rng = np.random.default_rng(20260101)
    for case in range(160):
        h = int(rng.integers(1, 261)); w = int(rng.integers(1, 261))
        pad_l = int(rng.integers(0, 20)); pad_r = int(rng.integers(0, 20)); pad_t = int(rng.integers(0, 3))
        kind = int(rng.integers(0, 4))
        big = np.zeros((h + pad_t + 2, w + pad_l + pad_r), np.uint8)
        view = big[pad_t:pad_t + h, pad_l:pad_l + w]
        if kind == 0:
            view[:] = rng.integers(0, 256, (h, w), dtype=np.uint8)
        elif kind == 1:
            lo = int(rng.integers(0, 200)); hi = lo + int(rng.integers(1, 56))
            view[:] = rng.integers(lo, hi + 1, (h, w), dtype=np.uint8)
        elif kind == 2:
            view[:] = int(rng.integers(0, 256))
        else:
            view[:] = (np.add.outer(np.arange(h), np.arange(w)) * int(rng.integers(1, 5)) % 256).astype(np.uint8)
        want = oracle.equalize_hist(view)
        got = ctx.equalize_hist(view)
        assert np.array_equal(got, want), ("equalize", case, h, w, kind)
        tx = int(rng.integers(1, 21)); ty = int(rng.integers(1, 21))
        clip = float(rng.choice([0.0, 0.5, 1.0, 2.0, 3.0, 7.5, 40.0]))
        wantc = oracle.clahe(view, clip, tx, ty)
        assert np.array_equal(ctx.clahe(view, clip, tx, ty), wantc), ("clahe", case, h, w, kind, clip, tx, ty)
        if case % 4 == 0:                                       # in place through a strided view
            buf = big.copy(); v2 = buf[pad_t:pad_t + h, pad_l:pad_l + w]
            ctx.equalize_hist(v2, v2)
            assert np.array_equal(v2, want) and buf[:pad_t].sum() == 0, ("inplace", case)
    for case in range(12):                                      # device NV12 batches of random even sizes
        w = int(rng.integers(1, 80)) * 2; h = int(rng.integers(1, 60)) * 2; n = int(rng.integers(1, 9))
        frames = np.stack([synth.nv12_frame(w, h, synth.DISTS[int(rng.integers(0, 5))], 300 + case * 10 + k) for k in range(n)])
        d_in = dev(frames)
        d_out = torch.zeros_like(d_in)
        uv_mode = int(rng.integers(0, 2))
        ctx.equalize_hist_nv12_batch_dev(d_in, d_out, w, h, n, uv_mode)
        ctx.synchronize()
        out = host(d_out)
        for k in range(n):
            assert np.array_equal(out[k], oracle.nv12_frame(frames[k], w, h, uv_mode=uv_mode, op=0)), ("nv12", case, w, h, k)
        tx = int(rng.integers(1, 12)); ty = int(rng.integers(1, 12))
        ctx.clahe_nv12_batch_dev(d_in, d_out, w, h, n, uv_mode, 2.0, tx, ty)
        ctx.synchronize()
        out = host(d_out)
        for k in range(n):
            assert np.array_equal(out[k], oracle.nv12_frame(frames[k], w, h, uv_mode=uv_mode, op=1, clip_limit=2.0, tiles_x=tx, tiles_y=ty)), ("nv12-clahe", case, w, h, k)


def test_many_contexts_share_one_gpu():
    """More contexts than fused slots on one device: the extra ones take the three-kernel path; all agree."""
    w, h, n = 1920, 1080, 4
    frames = np.stack([synth.nv12_frame(w, h, "D2", 500 + k) for k in range(n)])
    d_in = dev(frames)
    want = [oracle.nv12_frame(frames[k], w, h, uv_mode=0, op=0) for k in range(n)]
    ctxs = [mi_lumaeq.Context(0) for _ in range(7)]
    try:
        for c in ctxs:
            c.set_option("two_kernel_max_frames", 0)               # four 1080p frames would take the two-kernel path: this test is about the fused slots
        outs = [torch.zeros_like(d_in) for _ in ctxs]
        for c, o in zip(ctxs, outs):
            c.equalize_hist_nv12_batch_dev(d_in, o, w, h, n, 0, stream=mi_lumaeq.STREAM_CTX)   # 7 private streams, concurrently
        for c in ctxs:
            c.synchronize(mi_lumaeq.STREAM_CTX)
        for o in outs:
            got = host(o)
            for k in range(n):
                assert np.array_equal(got[k], want[k])
    finally:
        for c in ctxs:
            c.close()


def test_registered_host_buffers(ctx):
    """mi_host_register: caller-pinned contiguous buffers are DMA'd directly; results identical, views still stage."""
    w, h = 1280, 720
    frame = synth.nv12_frame(w, h, "D2", 123).copy()
    out = np.zeros_like(frame)
    mi_lumaeq.host_register(frame)
    mi_lumaeq.host_register(out)
    try:
        for uv_mode in (0, 1):
            ctx.equalize_hist_nv12(frame, w, h, uv_mode, out=out)
            assert np.array_equal(out, oracle.nv12_frame(frame, w, h, uv_mode=uv_mode, op=0))
            ctx.clahe_nv12(frame, w, h, uv_mode, 2.0, 8, 8, out=out)
            assert np.array_equal(out, oracle.nv12_frame(frame, w, h, uv_mode=uv_mode, op=1, clip_limit=2.0, tiles_x=8, tiles_y=8))
        y = frame[: w * h].reshape(h, w)
        want = oracle.equalize_hist(y)
        assert np.array_equal(ctx.equalize_hist(y), want)                       # pinned src, staged dst
        roi = y[10:700, 16:1200]
        assert np.array_equal(ctx.equalize_hist(roi), oracle.equalize_hist(roi))   # strided view of pinned memory: staged
        ctx.equalize_hist(y, y)                                                  # in place in pinned memory
        assert np.array_equal(y, want)
    finally:
        mi_lumaeq.host_unregister(frame)
        mi_lumaeq.host_unregister(out)
    with pytest.raises(mi_lumaeq.MiError):
        mi_lumaeq.host_unregister(out)                                           # not registered any more


def test_hip_graph_capture_and_replay():
    """The batched device form can be captured into a HIP graph (after a warm-up call sized the scratch) and the
    graph replayed on new data: every per-launch datum of the fused path (ticket counter, epoch) lives in device memory."""
    w, h, n = 1920, 1080, 6
    c = mi_lumaeq.Context(0)
    try:
        c.set_option("two_kernel_max_frames", 0)                    # capture the FUSED pair (six 1080p frames would otherwise be routed to the two-kernel path, captured below)
        d_in = synth.nv12_batch_torch(w, h, n, "D2", "cuda:0", seed=11)
        d_out = torch.zeros_like(d_in)
        c.equalize_hist_nv12_batch_dev(d_in, d_out, w, h, n, 1)            # warm-up: allocations happen here, not in the capture
        c.clahe_nv12_batch_dev(d_in, d_out, w, h, n, 1, 2.0, 8, 8)
        c.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            c.equalize_hist_nv12_batch_dev(d_in, d_out, w, h, n, 1, stream=torch.cuda.current_stream().cuda_stream)
        for rep in range(3):
            d_in.copy_(synth.nv12_batch_torch(w, h, n, synth.DISTS[rep], "cuda:0", seed=100 + rep))
            d_out.zero_()
            g.replay()
            torch.cuda.synchronize()
            src, out = host(d_in), host(d_out)
            for k in range(n):
                assert np.array_equal(out[k], oracle.nv12_frame(src[k], w, h, uv_mode=1, op=0)), (rep, k)
        # eager calls on the same context keep working after replays
        c.equalize_hist_nv12_batch_dev(d_in, d_out, w, h, n, 0)
        c.synchronize()
        assert np.array_equal(host(d_out[0]), oracle.nv12_frame(host(d_in[0]), w, h, uv_mode=0, op=0))
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2):
            c.clahe_nv12_batch_dev(d_in, d_out, w, h, n, 0, 2.0, 8, 8, stream=torch.cuda.current_stream().cuda_stream)
        g2.replay()
        torch.cuda.synchronize()
        assert np.array_equal(host(d_out[1]), oracle.nv12_frame(host(d_in[1]), w, h, uv_mode=0, op=1, clip_limit=2.0, tiles_x=8, tiles_y=8))
        # the default routing of few frames (histogram + LUT in one launch, then apply) is capturable as well
        c.set_option("two_kernel_max_frames", 8)
        c.equalize_hist_nv12_batch_dev(d_in, d_out, w, h, 2, 1)            # sizes its scratch eagerly
        c.synchronize()
        g4 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g4):
            c.equalize_hist_nv12_batch_dev(d_in, d_out, w, h, 2, 1, stream=torch.cuda.current_stream().cuda_stream)
        for rep in range(2):
            d_in.copy_(synth.nv12_batch_torch(w, h, n, synth.DISTS[3 + rep], "cuda:0", seed=900 + rep))
            d_out.zero_()
            g4.replay()
            torch.cuda.synchronize()
            for k in range(2):
                assert np.array_equal(host(d_out[k]), oracle.nv12_frame(host(d_in[k]), w, h, uv_mode=1, op=0)), (rep, k)
        # 16-bit CLAHE: its per-frame arrival words live in the context and are left zero by every launch, so a captured sequence
        # replays on new content -- frames whose 12-bit bet holds and one that loses it
        w16, h16, n16 = 640, 368, 4
        rng = np.random.default_rng(61)
        s16 = dev(rng.integers(0, 4096, (n16, h16, w16), dtype=np.uint16).view(np.int16))
        o16 = torch.zeros_like(s16)
        c.clahe16_batch_dev(s16, o16, w16, h16, n16, 2.0, 8, 8)            # sizes the scratch eagerly
        c.synchronize()
        g5 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g5):
            c.clahe16_batch_dev(s16, o16, w16, h16, n16, 2.0, 8, 8, stream=torch.cuda.current_stream().cuda_stream)
        for rep in range(3):
            fr = rng.integers(0, 4096 if rep != 1 else 1024, (n16, h16, w16), dtype=np.uint16)
            if rep == 2: fr[2, 100, 100] = 50000
            s16.copy_(dev(fr.view(np.int16)))
            o16.zero_()
            g5.replay()
            torch.cuda.synchronize()
            out16 = host(o16).view(np.uint16)
            for k in range(n16):
                assert np.array_equal(out16[k], oracle.clahe16(fr[k], 2.0, 8, 8)), ("clahe16 replay", rep, k)
        # destroy the graphs while the context (whose scratch their kernel nodes point at) is still alive
        del g, g2, g4, g5
        torch.cuda.synchronize()
    finally:
        c.close()


def _stats(c):
    return {k: c.get_stat("fused_" + k) for k in ("fallbacks", "frames_repaired", "hard_errors", "last_status")}


@pytest.mark.parametrize("mode", [1, 2, 3], ids=["lost_producer", "partial_frame", "bad_checksum"])
@pytest.mark.parametrize("in_place", [False, True], ids=["out_of_place", "in_place"])
def test_fused_bounded_wait_expiry_is_repaired_on_device(mode, in_place):
    """Fail-soft path of the fused kernel, in the session process, with the default 50 ms bound.  A test hook breaks the
    inter-workgroup hand-off of one frame (1: the last arriver never publishes its LUT; 2: two workgroups leave a frame partly
    written after its LUT was published; 3: the published LUT never passes its checksum).  The waits expire, the grid drains,
    and the finish kernel that follows every fused launch redoes the missing tickets: the OUTPUT BYTES must be the oracle's,
    no error is raised, and the event shows up in the statistics -- for the device form on a caller stream (nothing polled on
    the host), in place and out of place, and for the host-pointer form with its copies queued behind the stalled kernel."""
    import time
    w, h, n = 1920, 1080, 3
    frames = np.stack([synth.nv12_frame(w, h, "D2", 700 + k) for k in range(n)])
    want = [oracle.nv12_frame(frames[k], w, h, uv_mode=0, op=0) for k in range(n)]
    c = hooks_ctx()
    try:
        d_in = dev(frames)
        d_out = d_in if in_place else torch.zeros_like(d_in)
        s0 = _stats(c)
        assert s0["fallbacks"] == 0 and s0["hard_errors"] == 0
        c.set_option("fused_fault_inject", mode)
        t0 = time.perf_counter()
        c.equalize_hist_nv12_batch_dev(d_in, d_out, w, h, n, 0, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()                                   # the caller's own synchronisation: no library call needed
        assert time.perf_counter() - t0 < 5.0                      # bounded: the grid drained
        out = host(d_out)
        for k in range(n):
            assert np.array_equal(out[k], want[k]), (mode, in_place, k)
        c.synchronize()                                            # ... and the library reports no error either
        s1 = _stats(c)
        assert s1["fallbacks"] == 1 and s1["frames_repaired"] >= 1 and s1["hard_errors"] == 0, s1
        assert s1["last_status"] == (2 if mode == 3 else 1), s1
        # the next launch on the same context is a normal one (hand-off block left clean, epoch advanced)
        c.set_option("fused_fault_inject", 0)
        d_in2 = dev(frames)
        d_out2 = torch.zeros_like(d_in2)
        c.equalize_hist_nv12_batch_dev(d_in2, d_out2, w, h, n, 1)
        c.synchronize()
        out = host(d_out2)
        for k in range(n):
            assert np.array_equal(out[k], oracle.nv12_frame(frames[k], w, h, uv_mode=1, op=0)), k
        assert _stats(c)["fallbacks"] == 1                         # sticky, and not re-triggered
        # host-pointer form: H2D, the stalled kernel, the finish kernel and the D2H all queue on the context's stream
        y = frames[0][: w * h].reshape(h, w)
        c.set_option("fused_fault_inject", mode if mode != 2 else 1)     # hook 2 needs >= 2 frames
        assert np.array_equal(c.equalize_hist(y), oracle.equalize_hist(y))
        assert _stats(c)["fallbacks"] == 2
        c.set_option("fused_fault_inject", 0)
        assert np.array_equal(c.equalize_hist(y), oracle.equalize_hist(y))
        s2 = _stats(c)
        assert s2["fallbacks"] == 2 and s2["hard_errors"] == 0, s2
    finally:
        c.close()


@pytest.mark.parametrize("in_place", [False, True], ids=["out_of_place", "in_place"])
def test_fused_repair_under_naturally_expiring_waits(in_place):
    """No injected fault: the bound of the inter-workgroup waits is set to a few MICROseconds, so ordinary hand-offs expire at
    whatever point the timing of that launch puts them -- consumers give up while their frame's LUT is being published, frames are
    left partly written, tickets are never drawn.  Whatever state a launch ends in, the finish kernel must turn it into the
    oracle's bytes (in place included), launch after launch, and the next launch must start from a clean hand-off block."""
    w, h, n = 1920, 1080, 6
    frames = np.stack([synth.nv12_frame(w, h, synth.DISTS[k % 5], 1500 + k) for k in range(n)])
    want = [[oracle.nv12_frame(frames[k], w, h, uv_mode=uv, op=0) for k in range(n)] for uv in (0, 1)]
    c = hooks_ctx()
    try:
        c.set_option("fused_demote_after", 0)                       # this test wants every launch on the fused path, however often it is repaired
        total_fallbacks = 0
        for us in (1, 2, 4, 8, 16, 40):
            c.set_option("fused_timeout_us", us)
            for rep in range(6):
                uv = rep & 1
                d_in = dev(frames)
                d_out = d_in if in_place else torch.zeros_like(d_in)
                c.equalize_hist_nv12_batch_dev(d_in, d_out, w, h, n, uv)
                c.synchronize()
                out = host(d_out)
                for k in range(n):
                    assert np.array_equal(out[k], want[uv][k]), (us, rep, k)
            fb = c.get_stat("fused_fallbacks")
            assert c.get_stat("fused_hard_errors") == 0
            total_fallbacks = fb
        assert total_fallbacks > 0                                  # the scenario did exercise the repair path
        c.set_option("fused_timeout_ms", 50)                        # back to the default bound: fast path, no new fallback
        d_in = dev(frames); d_out = torch.zeros_like(d_in)
        c.equalize_hist_nv12_batch_dev(d_in, d_out, w, h, n, 0)
        c.synchronize()
        assert c.get_stat("fused_fallbacks") == total_fallbacks
        assert np.array_equal(host(d_out[n - 1]), want[0][n - 1])
    finally:
        c.close()


def test_fused_failure_statistics_survive_later_launches_and_block_growth():
    """A failure in launch N must still be visible after launch N+1 ... N+k (the statistics words are never cleared by the
    per-launch housekeeping) and after the hand-off block was re-allocated for a larger batch."""
    w, h = 640, 368
    c = hooks_ctx()
    try:
        f4 = np.stack([synth.nv12_frame(w, h, "D1", 40 + k) for k in range(4)])
        d_in, d_out = dev(f4), torch.zeros(f4.shape, dtype=torch.uint8, device="cuda:0")
        c.set_option("fused_fault_inject", 1)
        c.equalize_hist_nv12_batch_dev(d_in, d_out, w, h, 4, 0)
        c.set_option("fused_fault_inject", 0)
        for _ in range(3):
            c.equalize_hist_nv12_batch_dev(d_in, d_out, w, h, 4, 0)
        c.synchronize()
        assert _stats(c)["fallbacks"] == 1
        out = host(d_out)
        for k in range(4):
            assert np.array_equal(out[k], oracle.nv12_frame(f4[k], w, h, uv_mode=0, op=0)), k
        n_big = 130                                                # > the block's initial capacity of 64 frames
        big = np.stack([synth.nv12_frame(w, h, "D2", 90 + k) for k in range(n_big)])
        d_in, d_out = dev(big), torch.zeros(big.shape, dtype=torch.uint8, device="cuda:0")
        c.equalize_hist_nv12_batch_dev(d_in, d_out, w, h, n_big, 1)
        c.synchronize()
        assert _stats(c)["fallbacks"] == 1 and _stats(c)["hard_errors"] == 0
        out = host(d_out)
        for k in (0, 63, 64, 129):
            assert np.array_equal(out[k], oracle.nv12_frame(big[k], w, h, uv_mode=1, op=0)), k
    finally:
        c.close()


def test_fused_failure_inside_a_replayed_graph():
    """A captured fused launch carries its finish kernel with it: a hand-off failure inside a REPLAY is repaired like an eager one,
    replay after replay (all per-launch state lives in the hand-off block, nothing on the host)."""
    w, h, n = 1280, 720, 3
    c = hooks_ctx()
    try:
        d_in = synth.nv12_batch_torch(w, h, n, "D2", "cuda:0", seed=5)
        d_out = torch.zeros_like(d_in)
        c.set_option("fused_demote_after", 0)                             # keep the eager calls below on the fused path
        c.equalize_hist_nv12_batch_dev(d_in, d_out, w, h, n, 0)           # sizes the scratch (allocations are not capturable)
        c.synchronize()
        c.set_option("fused_fault_inject", 1)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            c.equalize_hist_nv12_batch_dev(d_in, d_out, w, h, n, 0, stream=torch.cuda.current_stream().cuda_stream)
        c.set_option("fused_fault_inject", 0)
        for rep in range(3):
            d_in.copy_(synth.nv12_batch_torch(w, h, n, synth.DISTS[rep], "cuda:0", seed=300 + rep))
            d_out.zero_()
            g.replay()
            torch.cuda.synchronize()
            src, out = host(d_in), host(d_out)
            for k in range(n):
                assert np.array_equal(out[k], oracle.nv12_frame(src[k], w, h, uv_mode=0, op=0)), (rep, k)
        assert _stats(c)["fallbacks"] == 3
        # a batch that needs more scratch than the captured one: the old scratch stays alive for the graph, which still replays
        n2 = 80
        e_in = synth.nv12_batch_torch(w, h, n2, "D1", "cuda:0", seed=9)
        e_out = torch.zeros_like(e_in)
        c.equalize_hist_nv12_batch_dev(e_in, e_out, w, h, n2, 1)
        c.synchronize()
        assert np.array_equal(host(e_out[n2 - 1]), oracle.nv12_frame(host(e_in[n2 - 1]), w, h, uv_mode=1, op=0))
        d_out.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert np.array_equal(host(d_out[1]), oracle.nv12_frame(host(d_in[1]), w, h, uv_mode=0, op=0))
        # growth INSIDE a capture is refused, loudly, instead of corrupting the capture
        n3 = 200
        f_in = synth.nv12_batch_torch(w, h, n3, "D1", "cuda:0", seed=10)
        f_out = torch.zeros_like(f_in)
        g3 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g3):
            f_out.zero_()                                          # keeps the graph non-empty
            with pytest.raises(mi_lumaeq.MiError) as e:
                c.equalize_hist_nv12_batch_dev(f_in, f_out, w, h, n3, 1, stream=torch.cuda.current_stream().cuda_stream)
        assert e.value.status == 2 and "capture" in str(e.value)
        del g, g3
        torch.cuda.synchronize()
    finally:
        c.close()


@pytest.mark.parametrize("shape", [(1, 1), (15, 16), (47, 63), (270, 480), (360, 640), (1079, 1919)], ids=str)   # (360, 640): vector path
@pytest.mark.parametrize("cfg", [(2.0, 8, 8), (3.0, 4, 4), (0.0, 3, 5), (40.0, 1, 1)], ids=str)
def test_clahe16(ctx, shape, cfg):
    """CLAHE on CV_16UC1 (65 536 bins) vs the oracle, incl. flat and narrow-range images and strides."""
    h, w = shape
    clip, tx, ty = cfg
    rng = np.random.default_rng(h * 7 + w)
    for kind in range(4):
        if kind == 0:
            s = rng.integers(0, 65536, (h, w), dtype=np.uint16)
        elif kind == 1:
            s = rng.integers(1000, 1400, (h, w), dtype=np.uint16)
        elif kind == 2:
            s = np.full((h, w), 40000, np.uint16)
        else:
            big = rng.integers(0, 4096, (h + 2, w + 5), dtype=np.uint16)
            s = big[1:h + 1, 3:w + 3]                              # strided view
        assert np.array_equal(ctx.clahe16(s, clip, tx, ty), oracle.clahe16(s, clip, tx, ty)), kind


@pytest.mark.parametrize("cfg", [(2.0, 8, 8), (40.0, 3, 5), (0.0, 4, 4)], ids=str)
def test_clahe16_value_ranges(ctx, cfg):
    """The 16-bit kernels only touch the bins a frame populates: ranges below / above / across 32768 (one or two histogram
    passes), ranges that fit the LDS pair table and ranges that do not (8192 values is the edge), MSB-aligned 10-bit video
    (sparse values over the whole 16-bit range), shapes that need REFLECT_101 padding, and a batch whose frames differ."""
    clip, tx, ty = cfg
    rng = np.random.default_rng(int(clip) + tx)
    h, w = 272, 480
    ranges = [(0, 4096), (100, 101), (30000, 34000), (32768, 36864), (61000, 65536), (5000, 5000 + 8192), (5001, 5001 + 8193),
              (3, 3 + 8191), (0, 65536), (20000, 45000)]
    frames = [rng.integers(lo, hi, (h, w), dtype=np.uint16) for lo, hi in ranges]
    frames.append((rng.integers(0, 1024, (h, w), dtype=np.uint16) << 6).astype(np.uint16))         # P010-style MSB-aligned samples
    for k, s in enumerate(frames):
        assert np.array_equal(ctx.clahe16(s, clip, tx, ty), oracle.clahe16(s, clip, tx, ty)), (cfg, k)
    odd = rng.integers(700, 3000, (271, 479), dtype=np.uint16)                                        # padded tiles, unaligned rows
    assert np.array_equal(ctx.clahe16(odd, clip, tx, ty), oracle.clahe16(odd, clip, tx, ty))
    batch = np.stack(frames)
    d_in = dev(batch.view(np.int16))
    d_out = torch.empty_like(d_in)
    ctx.clahe16_batch_dev(d_in, d_out, w, h, len(frames), clip, tx, ty)
    ctx.synchronize()
    out = host(d_out).view(np.uint16)
    for k, s in enumerate(frames):
        assert np.array_equal(out[k], oracle.clahe16(s, clip, tx, ty)), (cfg, "batch", k)
    # in place
    d = dev(batch.view(np.int16))
    ctx.clahe16_batch_dev(d, d, w, h, len(frames), clip, tx, ty)
    ctx.synchronize()
    assert np.array_equal(host(d).view(np.uint16), out)


def test_clahe16_twelve_bit_bet_mixed_outcomes(ctx):
    """The 12-bit fast path of the 16-bit tile histograms (4096 bins x 4 LDS copies, LUT folded in) bets per TILE that every value is
    < 4096.  Frames where the bet holds everywhere, where ONE tile loses it late (a single bright pixel in its last row), where one
    loses it at once (bright pixels in the first vectors), where a whole frame is wide -- in one batch, so that tiles whose LUT was
    written by the fast path sit in frames whose range exceeds 4096 and must be redone by the LUT kernel -- and with the option off."""
    w, h = 640, 368                                                # 8x8 tiles of 80 x 46: vector geometry (80 % 8 == 0, no padding)
    rng = np.random.default_rng(12)
    base = rng.integers(0, 4096, (h, w), dtype=np.uint16)
    f_all12 = base.copy()
    f_late = base.copy(); f_late[45, 79] = 60000                  # last pixel of tile (0, 0)
    f_early = base.copy(); f_early[46:48, 80:160] = 9000          # first rows of tile (1, 1)
    f_one_high = base.copy(); f_one_high[200, 300] = 4096         # the smallest value that loses the bet
    f_max12 = base.copy(); f_max12[100, 100] = 4095; f_max12[0, 0] = 0
    f_wide = rng.integers(0, 65536, (h, w), dtype=np.uint16)
    f_const = np.full((h, w), 4095, np.uint16)
    frames = [f_all12, f_late, f_early, f_one_high, f_max12, f_wide, f_const]
    for cfg in ((2.0, 8, 8), (0.0, 8, 8), (40.0, 4, 8)):
        want = [oracle.clahe16(f, *cfg) for f in frames]
        try:
            for fast in (1, 0):
                ctx.set_option("clahe16_fast12", fast)
                for k, f in enumerate(frames):
                    assert np.array_equal(ctx.clahe16(f, *cfg), want[k]), (cfg, fast, "single", k)
                d_in = dev(np.stack(frames).view(np.int16))
                d_out = torch.empty_like(d_in)
                ctx.clahe16_batch_dev(d_in, d_out, w, h, len(frames), *cfg)
                ctx.synchronize()
                out = host(d_out).view(np.uint16)
                for k in range(len(frames)):
                    assert np.array_equal(out[k], want[k]), (cfg, fast, "batch", k)
                ctx.clahe16_batch_dev(d_in, d_in, w, h, len(frames), *cfg)            # in place
                ctx.synchronize()
                assert np.array_equal(host(d_in).view(np.uint16), out), (cfg, fast, "in place")
        finally:
            ctx.set_option("clahe16_fast12", 1)


def test_clahe16_msb_aligned_content(ctx):
    """10- and 12-bit samples stored in the HIGH bits of the 16-bit word (P010 / P016 video): every value is a multiple of 1 << shift,
    the frame spans the whole 16-bit range but populates 1024 / 4096 values.  The LUT kernel and the interpolation then work at
    value >> shift.  Shifts 1..8, black bars (tiles of zeros: any shift), a constant tile, one odd pixel that voids the shift, frames of
    different shifts in one batch, vector and padded geometry, in place -- every frame against the oracle."""
    rng = np.random.default_rng(2026)
    def msb(bits, shift, shape): return (rng.integers(0, 1 << bits, shape, dtype=np.uint32) << shift).astype(np.uint16)
    for (w, h, tx, ty) in [(640, 368, 8, 8), (323, 201, 4, 3), (1280, 96, 8, 2)]:
        f10 = msb(10, 6, (h, w))
        f12 = msb(12, 4, (h, w))
        f8 = msb(8, 8, (h, w))
        f15 = msb(15, 1, (h, w))                                      # 32 768 populated values: several table windows at shift 1
        bars = f10.copy(); bars[: h // 4] = 0; bars[-h // 5:] = 64 << 6   # black bars: tiles whose own shift is larger than the frame's
        flat = f12.copy(); flat[h // 3: h // 2, w // 4: w // 2] = 0x8000
        odd = f10.copy(); odd[h // 2, w // 2] |= 1                        # one odd value: the frame has no shift any more
        lsb = rng.integers(0, 4096, (h, w), dtype=np.uint16)            # ordinary 12-bit content beside them
        frames = [f10, f12, f8, f15, bars, flat, odd, lsb]
        for cfg in ((2.0, tx, ty), (0.0, tx, ty), (40.0, tx, ty)):
            want = [oracle.clahe16(f, *cfg) for f in frames]
            for k, f in enumerate(frames):
                assert np.array_equal(ctx.clahe16(f, *cfg), want[k]), (w, h, cfg, "single", k)
            d_in = dev(np.stack(frames).view(np.int16))
            d_out = torch.zeros_like(d_in)
            ctx.clahe16_batch_dev(d_in, d_out, w, h, len(frames), *cfg)
            ctx.synchronize()
            out = host(d_out).view(np.uint16)
            for k in range(len(frames)):
                assert np.array_equal(out[k], want[k]), (w, h, cfg, "batch", k)
            ctx.clahe16_batch_dev(d_in, d_in, w, h, len(frames), *cfg)    # in place
            ctx.synchronize()
            assert np.array_equal(host(d_in).view(np.uint16), out), (w, h, cfg, "in place")
    # letterboxed frames of alternating formats, call after call: the bars' tiles may take any shift their value allows, and take the one
    # the context's previous frame ran with -- a choice that must never show in the result
    w, h = 640, 368
    def letterbox(bits, shift, bar):
        f = msb(bits, shift, (h, w)); f[: h // 8] = bar; f[-(h // 8):] = bar; return f
    seq = [letterbox(10, 6, 64 << 6), letterbox(12, 4, 256 << 4), letterbox(10, 6, 64 << 6), letterbox(12, 0, 256), letterbox(10, 6, 0),
           letterbox(8, 8, 16 << 8), letterbox(10, 6, 64 << 6)]
    # ... and after a P010 frame (hint: shift 6) a nearly black frame with sparse ODD speckles: most tiles see nothing but zeros in
    # what they sample, go along with shift 6, and lose that bet on the pixels that follow
    speckle = np.zeros((h, w), np.uint16)
    idx = rng.choice(h * w, size=h * w // 4000, replace=False)
    speckle.reshape(-1)[idx] = rng.integers(1, 4096, idx.size, dtype=np.uint16) | 1
    seq += [speckle, letterbox(10, 6, 64 << 6), (speckle.astype(np.uint32) << 3).astype(np.uint16)]
    for k, f in enumerate(seq):
        assert np.array_equal(ctx.clahe16(f, 2.0, 8, 8), oracle.clahe16(f, 2.0, 8, 8)), ("letterbox sequence", k)
    try:                                                                  # the value-major LUT layout follows the shift too
        ctx.set_option("clahe16_transposed", 1)
        f = msb(10, 6, (180, 320))
        assert np.array_equal(ctx.clahe16(f, 2.0, 8, 8), oracle.clahe16(f, 2.0, 8, 8))
    finally:
        ctx.set_option("clahe16_transposed", 0)


def test_clahe16_tables_follow_the_local_range(ctx):
    """The LUT kernel writes a tile's LUT over the range of the tile and its eight neighbours, and an interpolation workgroup stages
    its table over the range of the four tiles it blends -- not over the frame's.  Frames whose tiles differ wildly: one hot pixel in a
    corner / on an edge / in the middle of a 12-bit frame, a full-range half beside a 12-bit half, a bright corner tile, a dark frame
    with one full-range tile; out of place and in place (in-place frames with a wide range take the gathering kernel, whole)."""
    rng = np.random.default_rng(404)
    for (w, h, tx, ty) in [(640, 368, 8, 8), (640, 368, 4, 2), (323, 201, 5, 3)]:
        base = rng.integers(0, 4096, (h, w), dtype=np.uint16)
        hot_corner = base.copy(); hot_corner[0, 0] = 65535
        hot_edge = base.copy(); hot_edge[h - 1, w // 2] = 40000
        hot_mid = base.copy(); hot_mid[h // 2, w // 2] = 65535; hot_mid[h // 2 + 1, w // 2] = 5000
        halves = base.copy(); halves[:, w // 2:] = rng.integers(0, 65536, (h, w - w // 2), dtype=np.uint16)
        corner = base.copy(); corner[: h // ty, : w // tx] = rng.integers(30000, 65536, (h // ty, w // tx), dtype=np.uint16)
        dark = rng.integers(0, 64, (h, w), dtype=np.uint16)
        dark[h // ty: 2 * (h // ty), w // tx: 2 * (w // tx)] = rng.integers(0, 65536, (h // ty, w // tx), dtype=np.uint16)
        frames = [hot_corner, hot_edge, hot_mid, halves, corner, dark, base]
        for cfg in ((2.0, tx, ty), (40.0, tx, ty)):
            want = [oracle.clahe16(f, *cfg) for f in frames]
            d_in = dev(np.stack(frames).view(np.int16))
            d_out = torch.zeros_like(d_in)
            ctx.clahe16_batch_dev(d_in, d_out, w, h, len(frames), *cfg)
            ctx.synchronize()
            out = host(d_out).view(np.uint16)
            for k in range(len(frames)):
                assert np.array_equal(out[k], want[k]), (w, h, cfg, "batch", k)
            ctx.clahe16_batch_dev(d_in, d_in, w, h, len(frames), *cfg)    # in place
            ctx.synchronize()
            got = host(d_in).view(np.uint16)
            for k in range(len(frames)):
                assert np.array_equal(got[k], want[k]), (w, h, cfg, "in place", k)


def test_clahe16_wide_content_full_size(ctx):
    """Content wider than 8192 values on full-size 4K frames, 8x8 tiles of 129 600 pixels (round 6).  Rectangles whose range needs
    8193..16384 table entries -- every rectangle of a 14-bit frame -- are interpolated by clahe_interp16_mid_kernel (ONE window of a
    128-KiB table, persistent, launched while such content was seen lately: option clahe16_wide 1 = that hint, 2 = always, 0 = never);
    wider ones keep the 64-KiB table's windows.  Full-range noise, 14-bit, 15-bit, a smooth full-range ramp (rectangles of every
    width), a 12-bit frame with one hot pixel, full-range noise with 70 000 pixels of ONE value in tile (3, 2), a half 12-bit / half
    full-range frame, and 14-bit samples in the high bits of the word (16384 entries in the compressed domain).  Out of place and IN
    PLACE (where frames wider than 16384 values keep the gathering kernel), every option value, the hint turning on over consecutive calls: every frame
    bit-exact against the oracle."""
    w, h = 3840, 2160
    rng = np.random.default_rng(606)
    def noise(lo, hi): return rng.integers(lo, hi, (h, w), dtype=np.uint16)
    full = noise(0, 65536)
    f14 = noise(0, 16384)
    f15 = noise(0, 32768)
    ramp = ((np.arange(h, dtype=np.uint32)[:, None] * 12 + np.arange(w, dtype=np.uint32)[None, :] * 10
             + rng.integers(0, 512, (h, w), dtype=np.uint32)) % 65536).astype(np.uint16)
    hot = noise(0, 4096); hot[1000, 2000] = 65535
    wrap = full.copy()
    ty0, tx0 = 2 * 270, 3 * 480
    blk = wrap[ty0: ty0 + 270, tx0: tx0 + 480].reshape(-1).copy()
    blk[rng.choice(blk.size, 70000, replace=False)] = 31337       # 70 000 > 65 535 pixels of one value, scattered (no flat vectors to speak of)
    wrap[ty0: ty0 + 270, tx0: tx0 + 480] = blk.reshape(270, 480)
    halves = noise(0, 4096); halves[:, w // 2:] = rng.integers(0, 65536, (h, w - w // 2), dtype=np.uint16)
    msb14 = (rng.integers(0, 16384, (h, w), dtype=np.uint32) << 2).astype(np.uint16)
    frames = [full, f14, f15, ramp, hot, wrap, halves, msb14]
    names = ["full", "14-bit", "15-bit", "ramp", "hot pixel", "counter wrap", "halves", "14-bit << 2"]
    for cfg in ((2.0, 8, 8), (40.0, 8, 8)):
        want = [oracle.clahe16(f, *cfg) for f in frames]
        try:
            # 2: the mid kernel is always launched; 0: never; 1 (the default): when the context's last calls met a 14-bit rectangle -- the
            # first such call runs without it, the pinned hint word turns, the following ones launch it
            for wide in (2, 0, 1, 1, 1):
                ctx.set_option("clahe16_wide", wide)
                d_in = dev(np.stack(frames).view(np.int16))
                d_out = torch.zeros_like(d_in)
                ctx.clahe16_batch_dev(d_in, d_out, w, h, len(frames), *cfg)
                ctx.synchronize()
                out = host(d_out).view(np.uint16)
                bad = [names[k] for k in range(len(frames)) if not np.array_equal(out[k], want[k])]
                assert not bad, (cfg, wide, "batch", bad)
                del d_out
                ctx.clahe16_batch_dev(d_in, d_in, w, h, len(frames), *cfg)            # in place
                ctx.synchronize()
                got = host(d_in).view(np.uint16)
                bad = [names[k] for k in range(len(frames)) if not np.array_equal(got[k], want[k])]
                assert not bad, (cfg, wide, "in place", bad)
                del d_in
        finally:
            ctx.set_option("clahe16_wide", 1)
    # other tile grids: small tiles, a grid with more pairs than a workgroup has row phases, one tile, and one whose pair edges do not
    # fall on multiples of eight pixels (1000 / 5 = 200-pixel tiles: edges at 100 + 200 k: 8-pixel groups cut by an edge)
    ctx.set_option("clahe16_wide", 2)
    # ... and 64 x 64 tiles: 65 x 65 rectangles, more than the in-place gathering kernel's ownership table holds (it then asks per pixel)
    for (cw, chh, tx, ty) in [(640, 368, 8, 8), (1280, 96, 16, 2), (512, 512, 1, 1), (1024, 64, 2, 4), (1000, 120, 5, 3), (1024, 1024, 64, 64)]:
        fs = [rng.integers(0, 65536, (chh, cw), dtype=np.uint16), rng.integers(0, 16384, (chh, cw), dtype=np.uint16),
              rng.integers(20000, 45000, (chh, cw), dtype=np.uint16)]
        fs.append(fs[0].copy()); fs[-1][: chh // 2] = 4242                                 # flat half: wave-uniform vectors
        for clip in (2.0, 0.0):
            d_in = dev(np.stack(fs).view(np.int16))
            d_out = torch.zeros_like(d_in)
            ctx.clahe16_batch_dev(d_in, d_out, cw, chh, len(fs), clip, tx, ty)
            ctx.synchronize()
            out = host(d_out).view(np.uint16)
            for k, f in enumerate(fs):
                assert np.array_equal(out[k], oracle.clahe16(f, clip, tx, ty)), (cw, chh, tx, ty, clip, k)
            ctx.clahe16_batch_dev(d_in, d_in, cw, chh, len(fs), clip, tx, ty)
            ctx.synchronize()
            assert np.array_equal(host(d_in).view(np.uint16), out), (cw, chh, tx, ty, clip, "in place")
    ctx.set_option("clahe16_wide", 1)


def test_clahe16_frame_done_flags_over_many_frames(ctx):
    """The last tile workgroup of each FRAME settles the frame's range and whether every tile wrote its LUT in the histogram kernel
    (the LUT kernel then leaves on one scalar load); the per-frame arrival words must come back to zero after every launch.  150
    small frames whose outcomes alternate -- 12-bit, 10-bit, narrow, one pixel >= 4096, full range -- in calls of 150, 40 and 150
    frames, out of place and in place: every frame against the oracle."""
    w, h = 128, 64                                                 # 8x8 tiles of 16 x 8 pixels: vector geometry
    rng = np.random.default_rng(77)
    def frame(k):
        kind = k % 5
        if kind == 0: return rng.integers(0, 4096, (h, w), dtype=np.uint16)
        if kind == 1: return rng.integers(0, 1024, (h, w), dtype=np.uint16)
        if kind == 2: return rng.integers(1000 + k, 1400 + k, (h, w), dtype=np.uint16)
        if kind == 3:
            f = rng.integers(0, 4096, (h, w), dtype=np.uint16); f[(7 * k) % h, (13 * k) % w] = 4096 + 100 * k; return f
        return rng.integers(0, 65536, (h, w), dtype=np.uint16)
    frames = np.stack([frame(k) for k in range(150)])
    want = np.stack([oracle.clahe16(f, 2.0, 8, 8) for f in frames])
    for n in (150, 40, 150):
        d_in = dev(frames[:n].view(np.int16))
        d_out = torch.zeros_like(d_in)
        ctx.clahe16_batch_dev(d_in, d_out, w, h, n, 2.0, 8, 8)
        ctx.synchronize()
        out = host(d_out).view(np.uint16)
        bad = [k for k in range(n) if not np.array_equal(out[k], want[k])]
        assert not bad, (n, bad[:10])
        ctx.clahe16_batch_dev(d_in, d_in, w, h, n, 2.0, 8, 8)
        ctx.synchronize()
        assert np.array_equal(host(d_in).view(np.uint16), want[:n]), (n, "in place")


def test_clahe16_mid_kernel_follows_the_content_it_sees():
    """The hint that launches clahe_interp16_mid_kernel (pinned host words stamped by the kernels, counted in EXECUTED calls): a fresh
    context on 12-bit content never launches it; on 14-bit content the first call runs without it and, once that call has executed,
    the following ones launch it -- also when twenty of them are enqueued without waiting; back on 12-bit content it is launched for at
    most eight more executed calls (plus what was enqueued meanwhile) and then not again; option 0 / 2 = never / always; in place
    as well (a frame of up to 16384 values is one window per rectangle).  Results are compared with the oracle at every change of regime (the hint must never show in the bytes)."""
    w, h, n = 640, 368, 2
    rng = np.random.default_rng(1416)
    f12 = rng.integers(0, 4096, (n, h, w), dtype=np.uint16)
    f14 = rng.integers(0, 16384, (n, h, w), dtype=np.uint16)
    want12 = np.stack([oracle.clahe16(f, 2.0, 8, 8) for f in f12])
    want14 = np.stack([oracle.clahe16(f, 2.0, 8, 8) for f in f14])
    with mi_lumaeq.Context(0) as c:
        d12, d14 = dev(f12.view(np.int16)), dev(f14.view(np.int16))
        out = torch.zeros_like(d12)
        def run(d, sync=True):
            c.clahe16_batch_dev(d, out, w, h, n, 2.0, 8, 8)
            if sync:
                c.synchronize()
        def launched(): return c.get_stat("clahe16_mid_launches")
        for _ in range(5):
            run(d12)
        assert launched() == 0                                           # narrow content: never
        assert np.array_equal(host(out).view(np.uint16), want12)
        run(d14)
        assert launched() == 0                                           # the first 14-bit call: nothing was known yet
        assert np.array_equal(host(out).view(np.uint16), want14)
        run(d14)
        assert launched() == 1                                           # ... now it is
        assert np.array_equal(host(out).view(np.uint16), want14)
        for _ in range(20):
            run(d14, sync=False)                                         # a caller that enqueues far ahead of the device
        c.synchronize()
        assert launched() == 21 and np.array_equal(host(out).view(np.uint16), want14)
        for _ in range(12):
            run(d12)                                                     # narrow again: eight more executed calls, then no more
        k = launched()
        assert 21 + 8 <= k <= 21 + 9, k
        assert np.array_equal(host(out).view(np.uint16), want12)
        for _ in range(5):
            run(d12)
        assert launched() == k
        c.set_option("clahe16_wide", 2)
        run(d12)
        assert launched() == k + 1                                       # always
        c.clahe16_batch_dev(d14, d14, w, h, n, 2.0, 8, 8)               # in place too: a 14-bit frame is one window per rectangle
        c.synchronize()
        assert launched() == k + 2 and np.array_equal(host(d14).view(np.uint16), want14)
        c.set_option("clahe16_wide", 0)
        d14 = dev(f14.view(np.int16))
        run(d14); run(d14)
        assert launched() == k + 2 and np.array_equal(host(out).view(np.uint16), want14)      # never


def test_clahe16_suite_again_with_the_mid_kernel_always_launched(ctx):
    """With the option at its default clahe_interp16_mid_kernel is launched only after a 14-bit rectangle was seen (a hint in pinned
    memory), so which kernels the tests above ran depends on their order.  Here every one of them runs again with the kernel ALWAYS
    launched: small frames, mixed batches, MSB-aligned content, hot pixels, 150-frame calls."""
    try:
        ctx.set_option("clahe16_wide", 2)
        for cfg in [(2.0, 8, 8), (40.0, 3, 5), (0.0, 4, 4)]:
            test_clahe16_value_ranges(ctx, cfg)
        test_clahe16_twelve_bit_bet_mixed_outcomes(ctx)
        test_clahe16_msb_aligned_content(ctx)
        test_clahe16_tables_follow_the_local_range(ctx)
        test_clahe16_frame_done_flags_over_many_frames(ctx)
        for shape in [(270, 480), (360, 640)]:
            test_clahe16(ctx, shape, (2.0, 8, 8))
    finally:
        ctx.set_option("clahe16_wide", 1)


def test_clahe16_batch_and_errors(ctx):
    w, h, n = 320, 180, 3
    rng = np.random.default_rng(3)
    frames = rng.integers(0, 65536, (n, h, w), dtype=np.uint16)
    d_in = dev(frames.view(np.int16))
    d_out = torch.empty_like(d_in)
    ctx.clahe16_batch_dev(d_in, d_out, w, h, n, 2.0, 8, 8)
    ctx.synchronize()
    out = host(d_out).view(np.uint16)
    for k in range(n):
        assert np.array_equal(out[k], oracle.clahe16(frames[k], 2.0, 8, 8)), k
    # both LUT layouts of the interpolation (value-major is an option for <= 64 tiles) and a grid too large for it
    try:
        for layout in (0, 1):
            ctx.set_option("clahe16_transposed", layout)
            for cfg in [(2.0, 8, 8), (3.0, 4, 4), (2.0, 9, 8), (40.0, 1, 1)]:
                assert np.array_equal(ctx.clahe16(frames[1], *cfg), oracle.clahe16(frames[1], *cfg)), (layout, cfg)
    finally:
        ctx.set_option("clahe16_transposed", 0)
    # full-size frame through the vectorised tile histogram (tile rows of 480 pixels, no padding), 12-bit content
    big = np.random.default_rng(5).integers(0, 4096, (2160, 3840), dtype=np.uint16)
    assert np.array_equal(ctx.clahe16(big, 2.0, 8, 8), oracle.clahe16(big, 2.0, 8, 8))
    with pytest.raises(mi_lumaeq.MiError):
        ctx.clahe16(frames[0].astype(np.uint8))
    with pytest.raises(mi_lumaeq.MiError):
        ctx.clahe16(frames[0], 2.0, 0, 8)


def test_more_frames_than_grid_limit(ctx):
    """70 000 tiny frames in one call: the fused path's ticket space and the three-kernel path's 65 535-frame chunks."""
    w, h, n = 4, 4, 70000
    rng = np.random.default_rng(2)
    frames = rng.integers(0, 256, (n, w * h + w * h // 2), dtype=np.uint8)
    d_in = dev(frames)
    ys = frames[:, : w * h].reshape(n, h, w)
    check = [0, 1, 65534, 65535, 65536, 69999]
    try:
        for fused in (1, 0):
            ctx.set_option("fused", fused)
            d_out = torch.zeros_like(d_in)
            ctx.equalize_hist_nv12_batch_dev(d_in, d_out, w, h, n, 1)
            ctx.synchronize()
            out = host(d_out)
            for k in check:
                assert np.array_equal(out[k], oracle.nv12_frame(frames[k], w, h, uv_mode=1, op=0)), (fused, k)
            # every frame: compare against a vectorised numpy equalize of all 70k frames
            srt = np.sort(ys.reshape(n, -1), axis=1)
            assert np.array_equal(out[:, w * h:], frames[:, w * h:])
            assert (out[:, : w * h].min(axis=1) == np.where(srt[:, 0] == srt[:, -1], srt[:, 0], 0)).all()
        d_out = torch.zeros_like(d_in)
        ctx.clahe_nv12_batch_dev(d_in, d_out, w, h, n, 0, 2.0, 2, 2)
        ctx.synchronize()
        out = host(d_out)
        for k in check:
            assert np.array_equal(out[k], oracle.nv12_frame(frames[k], w, h, uv_mode=0, op=1, clip_limit=2.0, tiles_x=2, tiles_y=2)), k
    finally:
        ctx.set_option("fused", 1)


def test_bgr_luma_paths_agree(ctx):
    """The 9 B/px two-pass BGR path and the planar cvtColor+split -> op -> merge+cvtColor path give the oracle's bytes."""
    try:
        for fused in (1, 0):
            ctx.set_option("bgr_fused", fused)
            for (h, w) in [(1, 1), (5, 3), (47, 63), (360, 640), (1080, 1920)]:
                a = _bgr(w, h, 31)
                assert np.array_equal(ctx.bgr_luma_op(a, mi_lumaeq.OP_EQUALIZE), oracle.bgr_luma_op(a, 0)), (fused, h, w)
            big = _bgr(400, 300, 9)
            view = big[11:289, 7:391]
            assert np.array_equal(ctx.bgr_luma_op(view, mi_lumaeq.OP_EQUALIZE), oracle.bgr_luma_op(np.ascontiguousarray(view), 0)), fused
            flat = np.full((64, 64, 3), (10, 200, 30), np.uint8)                      # constant luma: LUT shortcut
            assert np.array_equal(ctx.bgr_luma_op(flat, mi_lumaeq.OP_EQUALIZE), oracle.bgr_luma_op(flat, 0)), fused
            # CLAHE: the two-pass kernels take the unpadded, 16-pixel-aligned shapes, everything else goes through planes
            for (h, w), (clip, tx, ty) in [((1080, 1920), (2.0, 8, 8)), ((360, 640), (3.0, 4, 4)), ((96, 448), (40.0, 14, 3)),
                                           ((64, 256), (0.0, 1, 1)), ((360, 640), (2.0, 8, 8)), ((47, 63), (3.0, 4, 4)), ((128, 480), (2.0, 15, 2))]:
                a = _bgr(w, h, 41)
                assert np.array_equal(ctx.bgr_luma_op(a, mi_lumaeq.OP_CLAHE, clip, tx, ty), oracle.bgr_luma_op(a, 1, clip, tx, ty)), (fused, h, w, tx, ty)
            inplace = _bgr(640, 360, 42)
            want = oracle.bgr_luma_op(inplace, 1, 2.0, 4, 4)
            d = dev(np.stack([inplace, _bgr(640, 360, 43)]))
            ctx.bgr_luma_op_batch_dev(d, d, 640, 360, 2, mi_lumaeq.OP_CLAHE, 2.0, 4, 4)       # batch, in place on the device
            torch.cuda.synchronize()
            assert np.array_equal(host(d[0]), want), fused
            assert np.array_equal(host(d[1]), oracle.bgr_luma_op(_bgr(640, 360, 43), 1, 2.0, 4, 4)), fused
    finally:
        ctx.set_option("bgr_fused", 1)


def _nv12_frames(w, h, n, seed, low_contrast=False):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 256, (n, w * h * 3 // 2), dtype=np.uint8)
    if low_contrast:
        a = (a // 4 + 90).astype(np.uint8)
    return a


@pytest.mark.parametrize("size", [(2, 2), (16, 2), (18, 6), (32, 32), (62, 34), (640, 360), (1920, 1080)], ids=str)
def test_nv12_bgr_channel_equalize_host_form(ctx, size):
    """BASELINE config 5 read literally: NV12 -> BGR -> equalizeHist on B, G, R -> NV12, bit-exact vs the oracle
    (vector path when W % 16 == 0, one 2x2 block per lane otherwise)."""
    w, h = size
    for lc in (False, True):
        a = _nv12_frames(w, h, 1, 21 + w, lc)[0]
        got = ctx.nv12_bgr_equalize(a, w, h)
        assert np.array_equal(got, oracle.nv12_bgr_equalize(a, w, h)), (size, lc)
    # in place
    b = a.copy()
    ctx.nv12_bgr_equalize(b, w, h, out=b)
    assert np.array_equal(b, oracle.nv12_bgr_equalize(a, w, h))


def test_nv12_bgr_channel_equalize_batch_and_errors(ctx):
    import torch
    w, h, n = 320, 180, 5
    a = _nv12_frames(w, h, n, 5)
    d_in = dev(a)
    d_out = torch.empty_like(d_in)
    ctx.nv12_bgr_equalize_batch_dev(d_in, d_out, w, h, n)
    torch.cuda.synchronize()
    got = host(d_out)
    for k in range(n):
        assert np.array_equal(got[k], oracle.nv12_bgr_equalize(a[k], w, h)), k
    # unaligned frame base (vector path must not be taken): frames start 1 byte into the buffer
    w2, h2 = 64, 16
    fb = w2 * h2 * 3 // 2
    a2 = _nv12_frames(w2, h2, 3, 6)
    raw = torch.zeros(3 * fb + 1, dtype=torch.uint8, device="cuda")
    raw[1:] = dev(a2.reshape(-1))
    out2 = torch.zeros_like(raw)
    torch.cuda.synchronize()
    ctx.nv12_bgr_equalize_batch_dev(raw.data_ptr() + 1, out2.data_ptr() + 1, w2, h2, 3)
    torch.cuda.synchronize()
    got2 = host(out2[1:]).reshape(3, fb)
    for k in range(3):
        assert np.array_equal(got2[k], oracle.nv12_bgr_equalize(a2[k], w2, h2)), k
    assert int(out2[0]) == 0
    # full-size frame: every channel of the decoded output is an equalized plane -> idempotence is not exact in 4:2:0,
    # so check against the oracle on one 4K frame
    a4 = _nv12_frames(3840, 2160, 1, 8, True)
    d4 = dev(a4)
    ctx.nv12_bgr_equalize_batch_dev(d4, d4, 3840, 2160, 1)
    torch.cuda.synchronize()
    assert np.array_equal(host(d4)[0], oracle.nv12_bgr_equalize(a4[0], 3840, 2160))
    # errors: odd sizes, null pointers; empty is a no-op
    with pytest.raises(mi_lumaeq.MiError):
        ctx.nv12_bgr_equalize(np.zeros(3 * 2 * 3 // 2, np.uint8), 3, 2)
    with pytest.raises(mi_lumaeq.MiError):
        ctx.nv12_bgr_equalize_batch_dev(0, 0, 16, 16, 1)
    ctx.nv12_bgr_equalize_batch_dev(0, 0, 0, 0, 0)
    assert ctx.nv12_bgr_equalize(np.zeros(0, np.uint8), 0, 0).size == 0


@pytest.mark.parametrize("size", [(2, 2), (6, 4), (16, 2), (48, 18), (62, 34), (640, 360), (1920, 1080), (3840, 2160)], ids=str)
def test_cvt_color_420_codes(ctx, size):
    """cv::cvtColor COLOR_BGR2YUV_I420 (1frameMeasure.cpp:32) and COLOR_YUV2BGR_NV12 vs the oracle, host and device forms."""
    import torch
    w, h = size
    bgr = _bgr(w, h, 77)
    want_i420 = oracle.bgr_to_i420(bgr)
    assert np.array_equal(ctx.cvt_color_420(bgr, mi_lumaeq.COLOR_BGR2YUV_I420), want_i420)
    nv = _nv12_frames(w, h, 1, 78)[0].reshape(h * 3 // 2, w)
    want_bgr = oracle.nv12_to_bgr(nv, w, h)
    assert np.array_equal(ctx.cvt_color_420(nv, mi_lumaeq.COLOR_YUV2BGR_NV12), want_bgr)
    # strided host views on both sides
    big = np.zeros((h + 3, w + 5, 3), np.uint8); big[1:h + 1, 2:w + 2] = bgr
    outbig = np.zeros((h * 3 // 2 + 2, w + 7), np.uint8)
    ctx.cvt_color_420(big[1:h + 1, 2:w + 2], mi_lumaeq.COLOR_BGR2YUV_I420, dst=outbig[1:h * 3 // 2 + 1, 3:w + 3])
    assert np.array_equal(outbig[1:h * 3 // 2 + 1, 3:w + 3], want_i420) and outbig[0].sum() == 0 and outbig[:, :3].sum() == 0
    # device batch
    n = 3
    d_bgr = dev(np.stack([_bgr(w, h, 80 + k) for k in range(n)]))
    d_pl = torch.empty((n, h * 3 // 2, w), dtype=torch.uint8, device="cuda")
    ctx.cvt_color_420_batch_dev(d_bgr, d_pl, w, h, n, mi_lumaeq.COLOR_BGR2YUV_I420)
    d_nv = dev(_nv12_frames(w, h, n, 90))
    d_out = torch.empty((n, h, w, 3), dtype=torch.uint8, device="cuda")
    ctx.cvt_color_420_batch_dev(d_nv, d_out, w, h, n, mi_lumaeq.COLOR_YUV2BGR_NV12)
    torch.cuda.synchronize()
    for k in range(n):
        assert np.array_equal(host(d_pl[k]), oracle.bgr_to_i420(host(d_bgr[k]))), k
        assert np.array_equal(host(d_out[k]), oracle.nv12_to_bgr(host(d_nv[k]), w, h)), k


def test_colour_known_answers_on_gpu(ctx):
    """kat.json's "color" list (COL-1..8: one per family of tests/color_mutants.py, all 26 mutants killed on the CPU) through the HIP
    colour kernels: cvtColor BGR2YUV / YUV2BGR on a one-row image, NV12 -> BGR and BGR -> I420 (the NV12 encode with planar chroma)."""
    import json
    from pathlib import Path
    kats = json.loads((Path(__file__).parent / "golden" / "kat.json").read_text())["color"]
    assert len(kats) >= 8
    for k in kats:
        if k["op"] in ("bgr2yuv", "yuv2bgr"):
            px = np.array(k["src"], np.uint8).reshape(1, -1, 3).copy()
            got = ctx.cvt_color(px, mi_lumaeq.COLOR_BGR2YUV if k["op"] == "bgr2yuv" else mi_lumaeq.COLOR_YUV2BGR)
            assert got.reshape(-1).tolist() == k["dst"], k["id"]
            continue
        w, h = k["shape"]
        if k["op"] == "nv12_to_bgr":
            nv = np.array(k["src"], np.uint8).reshape(h * 3 // 2, w).copy()
            assert ctx.cvt_color_420(nv, mi_lumaeq.COLOR_YUV2BGR_NV12).reshape(-1).tolist() == k["dst"], k["id"]
        else:
            bgr = np.array(k["src"], np.uint8).reshape(h, w, 3).copy()
            i420 = ctx.cvt_color_420(bgr, mi_lumaeq.COLOR_BGR2YUV_I420).reshape(-1).tolist()
            n = w * h
            assert i420[:n] == k["dst"][:n] and i420[n:n + n // 4] == k["dst"][n::2] and i420[n + n // 4:] == k["dst"][n + 1::2], k["id"]


def test_cvt_color_420_errors(ctx):
    with pytest.raises(mi_lumaeq.MiError):
        ctx.cvt_color_420(np.zeros((3, 4, 3), np.uint8), mi_lumaeq.COLOR_BGR2YUV_I420)          # odd height
    with pytest.raises(mi_lumaeq.MiError):
        ctx.cvt_color_420_batch_dev(0, 0, 16, 16, 1, mi_lumaeq.COLOR_BGR2YUV_I420)               # null pointers
    with pytest.raises(mi_lumaeq.MiError):
        ctx.cvt_color_420_batch_dev(1, 1, 16, 16, 1, 82)                                          # not a 4:2:0 code
    ctx.cvt_color_420_batch_dev(0, 0, 0, 0, 0, mi_lumaeq.COLOR_BGR2YUV_I420)


def test_photo_like_scene(ctx):
    """A synthetic photo-like scene (mi_lumaeq.synth.photo_like, 1919 x 1079: piecewise-smooth gradients, large flat and saturated
    areas) -- hot histogram bins, runs of equal neighbours and LDS broadcasts that the noise distributions do not produce."""
    import torch
    y = synth.photo_like(1919, 1079, 20261004)
    crop = synth.photo_like(384, 256, 20261005, channels=3)
    assert np.array_equal(ctx.equalize_hist(y), oracle.equalize_hist(y))
    for cfg in [(2.0, 8, 8), (3.0, 4, 4), (40.0, 16, 2)]:
        assert np.array_equal(ctx.clahe(y, *cfg), oracle.clahe(y, *cfg)), cfg
    assert np.array_equal(ctx.bgr_luma_op(crop, mi_lumaeq.OP_EQUALIZE), oracle.bgr_luma_op(crop, 0))
    assert np.array_equal(ctx.bgr_luma_op(crop, mi_lumaeq.OP_CLAHE, 3.0, 4, 4), oracle.bgr_luma_op(crop, 1, 3.0, 4, 4))
    # 4K NV12 frames built from the photo (tiled and shifted), through the fused batch path and CLAHE
    w, h, n = 3840, 2160, 3
    big = np.tile(y, (3, 3))
    frames = np.empty((n, w * h * 3 // 2), np.uint8)
    for k in range(n):
        frames[k, : w * h] = big[37 * k: 37 * k + h, 91 * k: 91 * k + w].reshape(-1)
        frames[k, w * h:] = np.random.default_rng(k).integers(0, 256, w * h // 2, dtype=np.uint8)
    d_in = dev(frames)
    d_out = torch.empty_like(d_in)
    ctx.equalize_hist_nv12_batch_dev(d_in, d_out, w, h, n, mi_lumaeq.UV_COPY)
    ctx.synchronize()
    got = host(d_out)
    for k in range(n):
        yk = frames[k, : w * h].reshape(h, w)
        assert np.array_equal(got[k, : w * h].reshape(h, w), oracle.equalize_hist(yk)), k
        assert np.array_equal(got[k, w * h:], frames[k, w * h:]), k
    ctx.clahe_nv12_batch_dev(d_in, d_out, w, h, n, mi_lumaeq.UV_FILL128, 2.0, 8, 8)
    ctx.synchronize()
    got = host(d_out)
    for k in range(n):
        yk = frames[k, : w * h].reshape(h, w)
        assert np.array_equal(got[k, : w * h].reshape(h, w), oracle.clahe(yk, 2.0, 8, 8)), k
        assert (got[k, w * h:] == 128).all()
    ctx.nv12_bgr_equalize_batch_dev(d_in, d_out, w, h, 1)
    torch.cuda.synchronize()
    assert np.array_equal(host(d_out[0]), oracle.nv12_bgr_equalize(frames[0], w, h))


def test_host_forms_unpinned_and_pinned_memory_agree(ctx):
    """Unpinned host planes are packed through the context's pinned staging buffers (by the calling thread alone or together with the
    context's helper thread, option "host_copy_threads"); memory the caller pinned (here: pinned torch tensors, which the library
    recognises through hipPointerGetAttributes) is DMA'd as it is.  Same bytes every way, for every host-pointer form; the
    library has no mode any more in which pageable memory reaches hipMemcpyAsync."""
    w, h = 1920, 1080                                               # large enough for the helper thread to take part (>= 1 MiB planes)
    y = synth.y_plane(w, h, "D2", 3)
    nv = _nv12_frames(w, h, 1, 4)[0]
    bgr = _bgr(w, h, 5)
    s16 = (np.random.default_rng(6).integers(0, 4096, (h, w))).astype(np.uint16)
    with pytest.raises(mi_lumaeq.MiError):
        ctx.set_option("host_direct", 1)                            # the option of rounds 1-2 is gone
    try:
        outs = []
        shared0 = ctx.get_stat("host_copies_shared")
        for threads in (2, 1):
            ctx.set_option("host_copy_threads", threads)
            outs.append((ctx.equalize_hist(y), ctx.clahe(y, 2.0, 8, 8), ctx.equalize_hist_nv12(nv, w, h, mi_lumaeq.UV_COPY),
                         ctx.clahe_nv12(nv, w, h, mi_lumaeq.UV_FILL128, 3.0, 4, 4), ctx.bgr_luma_op(bgr, mi_lumaeq.OP_EQUALIZE),
                         ctx.cvt_color(bgr, mi_lumaeq.COLOR_BGR2YUV), ctx.cvt_color_420(bgr, mi_lumaeq.COLOR_BGR2YUV_I420),
                         ctx.nv12_bgr_equalize(nv, w, h), ctx.clahe16(s16, 2.0, 8, 8)))
            if threads == 2:
                shared1 = ctx.get_stat("host_copies_shared")
        assert ctx.get_stat("host_copies_shared") == shared1        # one thread: the helper took part in nothing
        assert shared1 >= shared0                                   # (it may lose every race for its half on a loaded host: >=, not >)
        for a, b in zip(*outs):
            assert np.array_equal(a, b)
        assert np.array_equal(outs[0][0], oracle.equalize_hist(y)) and np.array_equal(outs[1][1], oracle.clahe(y, 2.0, 8, 8))
        # strided views through the helper as well
        big = np.zeros((h + 4, w + 64), np.uint8)
        big[2:2 + h, 32:32 + w] = y
        dstbig = np.zeros_like(big)
        ctx.set_option("host_copy_threads", 2)
        ctx.equalize_hist(big[2:2 + h, 32:32 + w], dstbig[2:2 + h, 32:32 + w])
        assert np.array_equal(dstbig[2:2 + h, 32:32 + w], outs[0][0]) and not dstbig[:2].any() and not dstbig[:, :32].any()
        pin_in, pin_out = torch.from_numpy(y.copy()).pin_memory(), torch.empty((h, w), dtype=torch.uint8).pin_memory()
        got = ctx.equalize_hist(pin_in.numpy(), pin_out.numpy())
        assert np.array_equal(got, outs[0][0]) and got.ctypes.data == pin_out.numpy().ctypes.data
    finally:
        ctx.set_option("host_copy_threads", 2)


def test_clahe_fp_contract_mode(ctx):
    """Second arithmetic mode of the CLAHE interpolation (option "clahe_fp_contract"): the FMAs a GCC build of OpenCV forms on
    FMA targets such as the reference's aarch64 board.  Bit-exact against the oracle in the same mode for every kernel
    variant (f32 pair tables, uchar quads, global-LUT fallback, 16-bit, BGR wrapper), and really different from mode 0."""
    rng = np.random.default_rng(77)
    old = oracle.set_fp_contract(True)
    try:
        ctx.set_option("clahe_fp_contract", 1)
        differing = 0
        for (h, w), cfg in [((1080, 1920), (2.0, 8, 8)), ((1079, 1919), (3.0, 4, 4)), ((360, 640), (2.0, 16, 2)), ((47, 63), (40.0, 8, 8)),
                            ((128, 1024), (2.0, 70, 3)), ((2160, 3840), (2.0, 8, 8))]:
            y = synth.y_plane(w, h, "D2", 9)
            got = ctx.clahe(y, *cfg)
            assert np.array_equal(got, oracle.clahe(y, *cfg)), ((h, w), cfg)
            oracle.set_fp_contract(False)
            differing += int((got != oracle.clahe(y, *cfg)).sum())
            oracle.set_fp_contract(True)
        assert differing > 0
        ctx.set_option("clahe_float_tables", 0)                            # uchar-quad tables
        y = synth.y_plane(1920, 1080, "D1", 10)
        assert np.array_equal(ctx.clahe(y, 2.0, 8, 8), oracle.clahe(y, 2.0, 8, 8))
        ctx.set_option("clahe_float_tables", 1)
        s16 = rng.integers(0, 65536, (360, 640), dtype=np.uint16)
        assert np.array_equal(ctx.clahe16(s16, 2.0, 8, 8), oracle.clahe16(s16, 2.0, 8, 8))
        bgr = _bgr(640, 360, 12)
        assert np.array_equal(ctx.bgr_luma_op(bgr, mi_lumaeq.OP_CLAHE, 2.0, 8, 8), oracle.bgr_luma_op(bgr, 1, 2.0, 8, 8))
        import torch
        nv = dev(np.stack([synth.nv12_frame(1920, 1080, "D2", 13 + k) for k in range(3)]))
        out = torch.empty_like(nv)
        ctx.clahe_nv12_batch_dev(nv, out, 1920, 1080, 3, mi_lumaeq.UV_COPY, 2.0, 8, 8)
        ctx.synchronize()
        for k in range(3):
            assert np.array_equal(host(out[k]), oracle.nv12_frame(host(nv[k]), 1920, 1080, uv_mode=1, op=1, clip_limit=2.0, tiles_x=8, tiles_y=8)), k
    finally:
        oracle.set_fp_contract(old)
        ctx.set_option("clahe_fp_contract", 0)
        ctx.set_option("clahe_float_tables", 1)


def _pipe_expected(frame, w, h, op, uv_mode, cfg):
    if op == mi_lumaeq.OP_CHANNELS:
        return oracle.nv12_bgr_equalize(frame, w, h)
    return oracle.nv12_frame(frame, w, h, uv_mode=uv_mode, op=op, clip_limit=cfg[0], tiles_x=cfg[1], tiles_y=cfg[2])


@pytest.mark.parametrize("pinned", [True, False], ids=["registered", "pageable"])
@pytest.mark.parametrize("case", [(mi_lumaeq.OP_EQUALIZE, 0, mi_lumaeq.PIPE_UV_AUTO), (mi_lumaeq.OP_EQUALIZE, 1, mi_lumaeq.PIPE_UV_HOST),
                                  (mi_lumaeq.OP_EQUALIZE, 1, mi_lumaeq.PIPE_UV_DEVICE), (mi_lumaeq.OP_EQUALIZE, 0, mi_lumaeq.PIPE_UV_DEVICE),
                                  (mi_lumaeq.OP_CLAHE, 0, mi_lumaeq.PIPE_UV_HOST), (mi_lumaeq.OP_CLAHE, 1, mi_lumaeq.PIPE_UV_DEVICE),
                                  (mi_lumaeq.OP_CHANNELS, 0, mi_lumaeq.PIPE_UV_AUTO)], ids=str)
def test_pipe_frames_in_flight_complete_in_order_and_match_oracle(case, pinned):
    """mi_pipe (the per-GPU worker's device side, OpenCVequalHist.cpp:102-196 with frames in flight): every frame that goes
    through -- whatever the op, the UV mode, who writes the UV half, registered or pageable host memory -- comes back in
    submission order with the oracle's bytes; a full pipe says MI_ERR_BUSY instead of blocking or dropping."""
    op, uv_mode, policy = case
    w, h, n, depth = 640, 368, 11, 3
    cfg = (3.0, 4, 4)
    frames = [synth.nv12_frame(w, h, synth.DISTS[k % 5], 900 + k) for k in range(n)]
    ins = [f.copy() for f in frames]
    outs = [np.zeros_like(f) for f in frames]
    if pinned:
        for a in ins + outs:
            mi_lumaeq.host_register(a)
    c = mi_lumaeq.Context(0)
    try:
        with mi_lumaeq.Pipe(c, w, h, op=op, uv_mode=uv_mode, clip_limit=cfg[0], tiles_x=cfg[1], tiles_y=cfg[2], depth=depth, uv_policy=policy) as pipe:
            assert pipe.depth == depth and pipe.pending == 0
            done = []
            for k in range(n):
                while not pipe.submit(ins[k], outs[k], 1000 + k):          # full: complete the oldest first
                    assert pipe.pending == depth
                    done.append(pipe.wait())
            assert pipe.pending == min(depth, n)
            while pipe.pending:
                done.append(pipe.wait())
            assert [t for t, _ in done] == [1000 + k for k in range(n)]
            for k, (_, out) in enumerate(done):
                assert out is outs[k]
                assert np.array_equal(out, _pipe_expected(frames[k], w, h, op, uv_mode, cfg)), (case, pinned, k)
            with pytest.raises(mi_lumaeq.MiError):                          # nothing pending
                pipe.wait()
            # in place (the zero-copy variant, nextimprovement.cpp:159-168): out == in
            buf = frames[2].copy()
            if pinned:
                mi_lumaeq.host_register(buf)
            try:
                assert pipe.submit(buf, buf, 7)
                tag, out = pipe.wait()
                assert tag == 7 and np.array_equal(out, _pipe_expected(frames[2], w, h, op, uv_mode, cfg))
            finally:
                if pinned:
                    mi_lumaeq.host_unregister(buf)
        # the context's synchronous entry points still work once the pipe is gone
        y = frames[0][: w * h].reshape(h, w)
        assert np.array_equal(c.equalize_hist(y), oracle.equalize_hist(y))
    finally:
        c.close()
        if pinned:
            for a in ins + outs:
                mi_lumaeq.host_unregister(a)


@pytest.mark.parametrize("shape", [(322, 182), (2, 2), (4098, 6)], ids=str)
def test_pipe_odd_geometries(shape):
    """Frames whose Y plane is not a multiple of 16 bytes (the fused kernel does not apply: staged kernels inside the pipe), the
    smallest NV12 frame, and a very wide one -- every op, host and device UV."""
    w, h = shape
    frames = [synth.nv12_frame(w, h, synth.DISTS[k % 5], 1700 + k) for k in range(5)]
    c = mi_lumaeq.Context(0)
    try:
        for op in (mi_lumaeq.OP_EQUALIZE, mi_lumaeq.OP_CLAHE, mi_lumaeq.OP_CHANNELS):
            for policy in (mi_lumaeq.PIPE_UV_HOST, mi_lumaeq.PIPE_UV_DEVICE):
                outs = [np.zeros_like(f) for f in frames]
                with mi_lumaeq.Pipe(c, w, h, op=op, uv_mode=1, clip_limit=2.0, tiles_x=2, tiles_y=2, depth=2, uv_policy=policy) as pipe:
                    done = []
                    for k, f in enumerate(frames):
                        while not pipe.submit(f, outs[k], k):
                            done.append(pipe.wait()[0])
                    while pipe.pending:
                        done.append(pipe.wait()[0])
                assert done == list(range(len(frames)))
                for k, f in enumerate(frames):
                    assert np.array_equal(outs[k], _pipe_expected(f, w, h, op, 1, (2.0, 2, 2))), (shape, op, policy, k)
    finally:
        c.close()


def test_pipe_argument_errors_and_fail_soft():
    c = hooks_ctx()
    try:
        for bad in (dict(width=0), dict(width=641), dict(op=5), dict(uv_mode=3), dict(uv_policy=9), dict(op=mi_lumaeq.OP_CLAHE, tiles_x=0)):
            kw = dict(width=640, height=360, op=mi_lumaeq.OP_EQUALIZE, uv_mode=0, uv_policy=0, tiles_x=8, tiles_y=8)
            kw.update(bad)
            with pytest.raises(mi_lumaeq.MiError):
                mi_lumaeq.Pipe(c, kw["width"], kw["height"], op=kw["op"], uv_mode=kw["uv_mode"], tiles_x=kw["tiles_x"], tiles_y=kw["tiles_y"],
                               uv_policy=kw["uv_policy"])
        # a hand-off failure inside a pipelined frame is repaired on the device like anywhere else
        w, h = 1920, 1080
        f = synth.nv12_frame(w, h, "D2", 77)
        o = np.zeros_like(f)
        with mi_lumaeq.Pipe(c, w, h, depth=2) as pipe:
            c.set_option("fused_fault_inject", 1)
            assert pipe.submit(f, o, 1)
            c.set_option("fused_fault_inject", 0)
            f2, o2 = f.copy(), np.zeros_like(f)
            assert pipe.submit(f2, o2, 2)
            assert pipe.wait()[0] == 1 and pipe.wait()[0] == 2
        want = oracle.nv12_frame(f, w, h, uv_mode=0, op=0)
        assert np.array_equal(o, want) and np.array_equal(o2, want)
        assert c.get_stat("fused_fallbacks") == 1 and c.get_stat("fused_hard_errors") == 0
    finally:
        c.close()


def test_clahe_tuning_options_do_not_change_bytes(ctx):
    """The speed-only knobs of the CLAHE tile-histogram pass (XCD-aware tile order, 256- or 512-thread workgroups) give the oracle's
    bytes in every combination, for batches (one workgroup per tile, LUT folded in) and single frames (split tiles + LUT kernel)."""
    try:
        for (w, h, n, cfg) in ((1920, 1080, 5, (2.0, 8, 8)), (640, 368, 1, (3.0, 4, 4)), (1280, 720, 2, (2.0, 16, 8))):
            frames = np.stack([synth.nv12_frame(w, h, synth.DISTS[k % 5], 1200 + k) for k in range(n)])
            want = [oracle.nv12_frame(frames[k], w, h, uv_mode=1, op=1, clip_limit=cfg[0], tiles_x=cfg[1], tiles_y=cfg[2]) for k in range(n)]
            d_in = dev(frames)
            for xcd in (0, 1):
                for threads in (256, 512):
                    ctx.set_option("clahe_xcd_map", xcd)
                    ctx.set_option("clahe_hist_threads", threads)
                    d_out = torch.zeros_like(d_in)
                    ctx.clahe_nv12_batch_dev(d_in, d_out, w, h, n, 1, *cfg)
                    ctx.synchronize()
                    out = host(d_out)
                    for k in range(n):
                        assert np.array_equal(out[k], want[k]), (w, h, xcd, threads, k)
        with pytest.raises(mi_lumaeq.MiError):
            ctx.set_option("clahe_hist_threads", 300)
    finally:
        ctx.set_option("clahe_xcd_map", 1)
        ctx.set_option("clahe_hist_threads", 512)


@pytest.mark.parametrize("case", [(640, 360, 96, (2.0, 8, 8), 0), (638, 358, 96, (3.0, 8, 8), 0), (320, 180, 200, (2.0, 6, 4), 0),
                                  (640, 360, 140, (2.0, 8, 8), 4), (640, 360, 300, (2.0, 8, 8), 8), (1280, 720, 70, (2.0, 8, 8), 2),
                                  (1920, 1080, 40, (2.0, 8, 8), 2)], ids=str)
def test_clahe_small_tiles_in_large_batches(ctx, case):
    """Batches of small tiles let one tile-histogram workgroup walk several tiles (tile_hist_multi_kernel: chosen by tile size, or forced
    through option "clahe_tiles_per_wg"), with and without the XCD-aware order (a 6 x 4 grid has none), padded geometry included:
    every frame of the batch against the oracle."""
    w, h, n, cfg, k = case
    base = [synth.nv12_frame(w + (w & 1), h + (h & 1), synth.DISTS[i % 5], 2100 + i) for i in range(7)]
    ys = np.stack([base[i % 7][: (w + (w & 1)) * (h + (h & 1))].reshape(h + (h & 1), w + (w & 1))[:h, :w] for i in range(n)])
    ys = np.ascontiguousarray((ys.astype(np.uint16) + (np.arange(n, dtype=np.uint16) % 11)[:, None, None]).clip(0, 255).astype(np.uint8))
    want = {}
    try:
        ctx.set_option("clahe_tiles_per_wg", k)
        d_in = dev(ys)
        d_out = torch.zeros_like(d_in)
        ctx.clahe_batch_dev(d_in, d_out, w, h, n, *cfg)
        ctx.synchronize()
        out = host(d_out)
        for i in range(n):
            key = (i % 7, i % 11)
            if key not in want:
                want[key] = oracle.clahe(ys[i], *cfg)
            assert np.array_equal(out[i], want[key]), (case, i)
    finally:
        ctx.set_option("clahe_tiles_per_wg", 0)
    with pytest.raises(mi_lumaeq.MiError):
        ctx.set_option("clahe_tiles_per_wg", 9)


def test_contexts_and_pipes_release_their_device_memory():
    """Creating and destroying contexts, pipes and their scratch (fused hand-off block, ticket stamps, 16-bit CLAHE scratch, pipe
    frames, retired buffers of a captured context) gives the device memory back: 25 cycles must not move the free-memory mark."""
    w, h = 1280, 720
    frame = synth.nv12_frame(w, h, "D2", 5)
    y16 = np.random.default_rng(1).integers(0, 4096, (h, w), dtype=np.uint16)

    def cycle():
        c = mi_lumaeq.Context(0)
        try:
            out = np.zeros_like(frame)
            with mi_lumaeq.Pipe(c, w, h, depth=3) as pipe:
                assert pipe.submit(frame, out, 1)
                pipe.wait()
            d = dev(np.stack([frame] * 70))                          # > 64 frames: the hand-off block is laid out twice
            o = torch.empty_like(d)
            c.equalize_hist_nv12_batch_dev(d, o, w, h, 3, 0)
            c.equalize_hist_nv12_batch_dev(d, o, w, h, 70, 0)
            c.clahe_nv12_batch_dev(d, o, w, h, 4, 1, 2.0, 8, 8)
            c.synchronize()
            c.clahe16(y16, 2.0, 8, 8)
            del d, o
        finally:
            c.close()

    for _ in range(3):
        cycle()                                                      # allocator pools of torch and of the runtime settle
    torch.cuda.synchronize(); torch.cuda.empty_cache()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(25):
        cycle()
    torch.cuda.synchronize(); torch.cuda.empty_cache()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < (64 << 20), (free0, free1)               # one cycle allocates several hundred MiB
