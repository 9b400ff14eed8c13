"""Throughput of the 4:2:0 colour codes on their own (COLOR_BGR2YUV_I420, 1frameMeasure.cpp:32, and its NV12 inverse), 16 x 4K frames."""
import sys, time, torch
sys.path.insert(0, "opencv-opencl_amd/python"); sys.path.insert(0, ".")
import mi_lumaeq
ctx = mi_lumaeq.Context(0)
w, h, n = 3840, 2160, 16
bgr = torch.randint(0, 256, (n, h, w, 3), dtype=torch.uint8, device="cuda")
pl = torch.empty((n, h * 3 // 2, w), dtype=torch.uint8, device="cuda")
back = torch.empty_like(bgr)
for name, src, dst, code in (("BGR2YUV_I420", bgr, pl, mi_lumaeq.COLOR_BGR2YUV_I420), ("YUV2BGR_NV12", pl, back, mi_lumaeq.COLOR_YUV2BGR_NV12)):
    for _ in range(3): ctx.cvt_color_420_batch_dev(src, dst, w, h, n, code)
    ctx.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): ctx.cvt_color_420_batch_dev(src, dst, w, h, n, code)
    ctx.synchronize(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print(f"{name}: {n / dt:9.0f} frames/s, {4.5 * w * h * n / dt / 1e12:.2f} TB/s of 4.5 B/px")
