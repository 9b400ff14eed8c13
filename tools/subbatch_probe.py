"""Does the 256 MiB Infinity Cache pay for the second read of a two-pass op?  A 64-frame 4K batch is far larger than the cache: by the time
the second pass (CLAHE interpolation / LUT apply) re-reads a frame, the first pass's reads and the second pass's own traffic of ~29 MB per
frame in between have evicted it (the kernels already walk the frames in opposite orders, which saves the last ~9 frames of a launch).
Here the same batch is processed in SUB-BATCHES of n frames, each with its own pair of launches: with n * 29 MB under ~256 MiB every
second read should be served on-die.    python tools/subbatch_probe.py [batch]"""
import sys, time, torch
sys.path.insert(0, "opencv-opencl_amd/python"); sys.path.insert(0, ".")
import mi_lumaeq
from mi_lumaeq import synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ctx = mi_lumaeq.Context(0)
stream = torch.cuda.current_stream().cuda_stream
for (w, h, total) in ((3840, 2160, B), (1920, 1080, 4 * B)):
    fb = w * h * 3 // 2
    d_in = synth.nv12_batch_torch(w, h, total, "D2", "cuda", seed=1)
    d_out = torch.empty_like(d_in)
    ref = None
    for name in ("clahe 8x8", "equalize three-kernel"):
        ctx.set_option("fused", 0 if "three" in name else 1)
        ctx.set_option("two_kernel_max_frames", 0)          # sub-batches of <= 8 frames must take the same kernels as the whole batch
        for n in [total] + [k for k in (32, 16, 12, 8, 6, 4) if k < total and (w, h) == (3840, 2160)] + [k for k in (128, 64, 48, 32, 24, 16) if k < total and (w, h) == (1920, 1080)]:
            def run():
                for s in range(0, total, n):
                    m = min(n, total - s)
                    a, o = d_in.data_ptr() + s * fb, d_out.data_ptr() + s * fb
                    if "clahe" in name: ctx.clahe_nv12_batch_dev(a, o, w, h, m, 0, 2.0, 8, 8, stream=stream)
                    else: ctx.equalize_hist_nv12_batch_dev(a, o, w, h, m, 0, stream=stream)
            for _ in range(3): run()
            torch.cuda.synchronize()
            ts = []
            for rep in range(5):
                t0 = time.perf_counter()
                for _ in range(20): run()
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) / 20 * 1e6)
            ts.sort(); us = ts[len(ts) // 2]
            if n == total: ref = d_out.clone()
            same = bool(torch.equal(ref, d_out))
            print(f"{w}x{h} x{total} {name:22s} sub-batches of {n:3d}: {us:7.1f} us per {total} frames  {total / (us * 1e-6):9.0f} frames/s  "
                  f"whole path {3.5 * w * h * total / (us * 1e-6) / 8e12:.3f} of 8 TB/s  same bytes: {same}", flush=True)
    del d_in, d_out, ref
ctx.set_option("fused", 1); ctx.set_option("two_kernel_max_frames", 8)
ctx.close()
