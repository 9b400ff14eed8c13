"""Bounded random differential run, HIP vs oracle, aimed at the code paths a fixed test list visits only at a few points: ROI pitches
(multiples of 16 and not) with random origins, wide CLAHE grids at sizes where the per-segment float tables apply, large batches of small
tiles (several tiles per histogram workgroup), 16-bit CLAHE at random value ranges, the 4:2:0 codes at random aligned / unaligned sizes.
    python tools/stress_random.py [seconds] [seed] [option=value ...]        prints a line every ~15 s, exits non-zero on the first mismatch"""
import sys, time
sys.path.insert(0, "opencv-opencl_amd/python"); sys.path.insert(0, ".")
import numpy as np, torch
import mi_lumaeq, oracle
from mi_lumaeq import xfer
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = mi_lumaeq.Context(0)
for kv in sys.argv[3:]:                                          # name=value options for the whole run, e.g. clahe16_wide=2 (mid kernel always launched)
    k_, v_ = kv.split("="); ctx.set_option(k_, int(v_)); print("option", k_, "=", v_, flush=True)
t0 = last = time.time(); n = {"roi": 0, "grid": 0, "small": 0, "c16": 0, "420": 0, "nv12": 0}
def dev(a): return xfer.to_device(np.ascontiguousarray(a))
def fail(what, *info):
    print("MISMATCH", what, info, flush=True); sys.exit(1)
while time.time() - t0 < budget:
    k = int(rng.integers(0, 6))
    if k == 0:      # ROI batch on the device: random pitches, origins, sizes
        w, h, nf = int(rng.integers(1, 700)), int(rng.integers(1, 90)), int(rng.integers(1, 6))
        sp = w + int(rng.integers(0, 40)); sp += (16 - sp % 16) % 16 if rng.integers(0, 2) else 0
        dp = w + int(rng.integers(0, 40)); dp += (16 - dp % 16) % 16 if rng.integers(0, 2) else 0
        src = rng.integers(0, int(rng.integers(2, 257)), (nf, h + 4, sp), dtype=np.uint8)
        sx, dx = int(rng.integers(0, sp - w + 1)), int(rng.integers(0, dp - w + 1))
        d_src, d_dst = dev(src), torch.zeros((nf, h + 4, dp), dtype=torch.uint8, device="cuda")
        ctx.equalize_hist_batch_dev(d_src.data_ptr() + 2 * sp + sx, d_dst.data_ptr() + dp + dx, w, h, nf, src_step=sp, src_frame=(h + 4) * sp,
                                    dst_step=dp, dst_frame=(h + 4) * dp)
        ctx.synchronize(); out = xfer.to_host(d_dst)
        for f in range(nf):
            if not np.array_equal(out[f, 1:1 + h, dx:dx + w], oracle.equalize_hist(src[f, 2:2 + h, sx:sx + w])): fail("roi", w, h, nf, sp, dp, sx, dx, f)
        if out[:, 0].sum() or out[:, 1 + h:].sum(): fail("roi wrote outside", w, h, sp, dp)
        n["roi"] += 1
    elif k == 1:    # wide grids
        w, h = int(rng.integers(200, 2400)), int(rng.integers(20, 200))
        tx, ty = int(rng.integers(15, 64)), int(rng.integers(1, 9)); clip = float(rng.choice([0.0, 1.0, 2.0, 40.0]))
        y = rng.integers(0, 256, (h, w), dtype=np.uint8)
        if not np.array_equal(ctx.clahe(y, clip, tx, ty), oracle.clahe(y, clip, tx, ty)): fail("grid", w, h, tx, ty, clip)
        n["grid"] += 1
    elif k == 2:    # many small tiles in a large batch
        tw, th = int(rng.integers(2, 12)) * 8, int(rng.integers(4, 40)); tx, ty = int(rng.choice([4, 8, 8, 16])), int(rng.integers(2, 9))
        w, h = tw * tx - int(rng.integers(0, 2)) * int(rng.integers(0, tx)), th * ty
        nf = int(rng.integers(2100, 2600)) // max(1, tx * ty // 2) + 1
        ys = rng.integers(0, 256, (min(nf, 9), h, w), dtype=np.uint8)
        batch = np.ascontiguousarray(ys[np.arange(nf) % ys.shape[0]])
        d_in = dev(batch); d_out = torch.zeros_like(d_in)
        ctx.clahe_batch_dev(d_in, d_out, w, h, nf, 2.0, tx, ty); ctx.synchronize(); out = xfer.to_host(d_out)
        want = [oracle.clahe(ys[i], 2.0, tx, ty) for i in range(ys.shape[0])]
        for f in range(nf):
            if not np.array_equal(out[f], want[f % ys.shape[0]]): fail("small tiles", w, h, tx, ty, nf, f)
        n["small"] += 1
    elif k == 3:    # 16-bit CLAHE, random ranges
        w, h = int(rng.integers(4, 80)) * 8, int(rng.integers(4, 60)) * 4
        lo = int(rng.integers(0, 65000)); hi = min(65536, lo + int(rng.choice([1, 50, 1000, 4096, 4097, 8192, 8193, 20000, 65536])))
        y = rng.integers(lo, hi, (h, w), dtype=np.uint16)
        if rng.integers(0, 3) == 0: y[: h // 2] = lo
        sh = 0
        if rng.integers(0, 2) == 0:                                   # samples in the high bits of the word (P010 / P016): every value a multiple of 1 << sh
            sh = int(rng.integers(1, 9)); bits = int(rng.integers(1, 17 - sh))
            y = (rng.integers(0, 1 << bits, (h, w), dtype=np.uint32) << sh).astype(np.uint16)
            r = int(rng.integers(0, 4))
            if r == 0: y[: h // 3] = 0                                  # a black bar: tiles with a larger shift than the frame's
            elif r == 1: y[int(rng.integers(0, h)), int(rng.integers(0, w))] |= 1 << int(rng.integers(0, sh))   # one value voids (or lowers) the shift
        want16 = oracle.clahe16(y, 2.0, 4, 4)
        if not np.array_equal(ctx.clahe16(y, 2.0, 4, 4), want16): fail("c16", w, h, lo, hi, sh)
        if rng.integers(0, 3) == 0:                                   # the same frame three times in a device batch, out of place and in place
            d16 = dev(np.stack([y, y, y]).view(np.int16)); o16 = torch.zeros_like(d16)
            ctx.clahe16_batch_dev(d16, o16, w, h, 3, 2.0, 4, 4); ctx.synchronize()
            if not all(np.array_equal(xfer.to_host(o16)[f].view(np.uint16), want16) for f in range(3)): fail("c16 batch", w, h, lo, hi, sh)
            ctx.clahe16_batch_dev(d16, d16, w, h, 3, 2.0, 4, 4); ctx.synchronize()
            if not all(np.array_equal(xfer.to_host(d16)[f].view(np.uint16), want16) for f in range(3)): fail("c16 in place", w, h, lo, hi, sh)
        n["c16"] += 1
    elif k == 5:    # NV12 batches of random even sizes through the fused kernel (or the three-kernel path where it does not apply), both ops
        w, h, nf = int(rng.integers(1, 400)) * 2, int(rng.integers(1, 200)) * 2, int(rng.integers(1, 9))
        uv = int(rng.integers(0, 2)); op = int(rng.integers(0, 2))
        fr = rng.integers(0, int(rng.integers(2, 257)), (nf, w * h * 3 // 2), dtype=np.uint8)
        d_in = dev(fr); inplace = bool(rng.integers(0, 2)); d_out = d_in if inplace else torch.zeros_like(d_in)
        if op == 0: ctx.equalize_hist_nv12_batch_dev(d_in, d_out, w, h, nf, uv)
        else: ctx.clahe_nv12_batch_dev(d_in, d_out, w, h, nf, uv, 2.0, 4, 4)
        ctx.synchronize(); out = xfer.to_host(d_out)
        for f in range(nf):
            if not np.array_equal(out[f], oracle.nv12_frame(fr[f], w, h, uv_mode=uv, op=op, clip_limit=2.0, tiles_x=4, tiles_y=4)): fail("nv12", w, h, nf, uv, op, inplace, f)
        n["nv12"] += 1
    else:           # 4:2:0 codes
        w, h = int(rng.integers(1, 60)) * (16 if rng.integers(0, 2) else 2), int(rng.integers(1, 40)) * 2
        bgr = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        if not np.array_equal(ctx.cvt_color_420(bgr, mi_lumaeq.COLOR_BGR2YUV_I420), oracle.bgr_to_i420(bgr)): fail("i420", w, h)
        nv = rng.integers(0, 256, (h * 3 // 2, w), dtype=np.uint8)
        if not np.array_equal(ctx.cvt_color_420(nv, mi_lumaeq.COLOR_YUV2BGR_NV12), oracle.nv12_to_bgr(nv, w, h)): fail("nv12->bgr", w, h)
        n["420"] += 1
    if time.time() - last > 15:
        last = time.time(); print(f"[{last - t0:5.0f} s] cases {n}", flush=True)
print(f"stress_random: {sum(n.values())} cases {n} in {time.time() - t0:.0f} s, no mismatch")
