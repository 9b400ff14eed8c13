"""Device-resident throughput by frame size (equalizeHist + UV=128 and CLAHE 8x8 clip 2.0 + UV=128), batches of ~0.8 GB."""
import sys, time, torch
sys.path.insert(0, "opencv-opencl_amd/python"); sys.path.insert(0, ".")
import mi_lumaeq
from mi_lumaeq import synth
ctx = mi_lumaeq.Context(0)
print(f"{'size':>11} {'batch':>5} {'equalize frames/s':>18} {'TB/s alg':>9} {'clahe frames/s':>15} {'TB/s alg':>9}")
for (w, h) in [(1280, 720), (1920, 1080), (2560, 1440), (3840, 2160), (7680, 4320)]:
    n = max(4, int(64 * 3840 * 2160 / (w * h)))
    d_in = synth.nv12_batch_torch(w, h, n, "D2", "cuda", seed=3)
    d_out = torch.empty_like(d_in)
    res = []
    for op in ("eq", "clahe"):
        fn = (lambda: ctx.equalize_hist_nv12_batch_dev(d_in, d_out, w, h, n, mi_lumaeq.UV_FILL128)) if op == "eq" else \
             (lambda: ctx.clahe_nv12_batch_dev(d_in, d_out, w, h, n, mi_lumaeq.UV_FILL128, 2.0, 8, 8))
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30): fn()
        ctx.synchronize(); torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 30
        res += [n / dt, 3.5 * w * h * n / dt / 1e12]
    print(f"{w:>5}x{h:<5} {n:>5} {res[0]:>18.0f} {res[1]:>9.2f} {res[2]:>15.0f} {res[3]:>9.2f}", flush=True)
    del d_in, d_out
