import sys, time, torch
sys.path.insert(0, "opencv-opencl_amd/python"); sys.path.insert(0, ".")
import mi_lumaeq
from mi_lumaeq import synth
ctx = mi_lumaeq.Context(0)
w, h, Bc = 3840, 2160, 16
stream = torch.cuda.current_stream().cuda_stream
bgr = torch.randint(0, 256, (Bc, h, w, 3), dtype=torch.uint8, device="cuda"); out = torch.empty_like(bgr)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for rep in range(3):
    ms = timeit(lambda: ctx.bgr_luma_op_batch_dev(bgr, out, w, h, Bc, mi_lumaeq.OP_EQUALIZE, stream=stream))
    ms2 = timeit(lambda: ctx.bgr_luma_op_batch_dev(bgr, out, w, h, Bc, mi_lumaeq.OP_CLAHE, 2.0, 8, 8, stream=stream))
    ms3 = timeit(lambda: ctx.cvt_color_batch_dev(bgr, out, w, h, Bc, mi_lumaeq.COLOR_BGR2YUV, stream=stream))
    print(f"bgr_yuv_equalize_bgr {Bc / (ms * 1e-3):9.0f} frames/s   bgr_yuv_clahe8x8_bgr {Bc / (ms2 * 1e-3):9.0f}   cvtcolor_bgr2yuv {2 * 3 * w * h * Bc / (ms3 * 1e-3) / 1e9:7.1f} GB/s", flush=True)
ctx.profile_read(True); ctx.set_profiling(True)
for _ in range(10): ctx.bgr_luma_op_batch_dev(bgr, out, w, h, Bc, mi_lumaeq.OP_EQUALIZE, stream=stream)
ctx.set_profiling(False); torch.cuda.synchronize()
print({k: round(v["total_ms"] / v["launches"] * 1e3, 1) for k, v in ctx.profile_read(True).items() if v["launches"]})
