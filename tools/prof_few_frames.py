"""Few device-resident frames per call (a stream's frame, a cv::Mat call): back-to-back call rate of the two-kernel path (histogram + LUT by the
last workgroup, then apply), the fused pair (fused + finish kernel) and the three-kernel path, by frame count and size.
    python tools/prof_few_frames.py"""
import sys, time, torch
sys.path.insert(0, "opencv-opencl_amd/python"); sys.path.insert(0, ".")
import mi_lumaeq
from mi_lumaeq import synth
ctx = mi_lumaeq.Context(0)
def rate(fn, reps=300):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
for (w, h) in ((3840, 2160), (1920, 1080), (1280, 720)):
    for n in (1, 2, 3, 4, 8, 16):
        nv = synth.nv12_batch_torch(w, h, n, "D2", "cuda", seed=7)
        out = torch.empty_like(nv)
        res = {}
        for name, opts in (("two-kernel", dict(two_kernel_max_frames=64, fused=1)), ("fused pair", dict(two_kernel_max_frames=0, fused=1)),
                           ("three-kernel", dict(two_kernel_max_frames=0, fused=0))):
            for k, v in opts.items(): ctx.set_option(k, v)
            res[name] = rate(lambda: ctx.equalize_hist_nv12_batch_dev(nv, out, w, h, n, mi_lumaeq.UV_FILL128))
        print(f"{w}x{h} frames={n:2d}: " + "  ".join(f"{k} {v:7.1f} us/call" for k, v in res.items()), flush=True)
        del nv, out
ctx.set_option("two_kernel_max_frames", 8); ctx.set_option("fused", 1)
