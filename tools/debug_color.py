import sys, numpy as np
sys.path.insert(0, "opencv-opencl_amd/python"); sys.path.insert(0, ".")
import mi_lumaeq, oracle
ctx = mi_lumaeq.Context(0)
rng = np.random.default_rng(5)
for (h, w) in [(3, 5), (4, 16), (2, 32), (47, 63)]:
    a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    got = ctx.bgr_luma_op(a, 0)
    want = oracle.bgr_luma_op(a, 0)
    bad = np.argwhere(got != want)
    print((h, w), "mismatches", len(bad), "of", a.size)
    if len(bad):
        print(" first bad idx", bad[:6].tolist())
        y, x, c = bad[0]
        print(" got", got[y, x].tolist(), "want", want[y, x].tolist(), "src", a[y, x].tolist())
        yuv = oracle.bgr2yuv(a)
        print(" yuv of src", yuv[y, x].tolist(), " got->yuv", oracle.bgr2yuv(got)[y, x].tolist(), "want->yuv", oracle.bgr2yuv(want)[y, x].tolist())
        # which channels differ
        print(" per-channel mismatch counts", [(got[..., k] != want[..., k]).sum() for k in range(3)])
