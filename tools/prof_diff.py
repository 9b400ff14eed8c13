"""Throughput of the analyzeDiff operator (absdiff + statistics) on device batches, 16 x 4K NV12 frames."""
import sys, time, torch
sys.path.insert(0, "opencv-opencl_amd/python"); sys.path.insert(0, ".")
import mi_lumaeq
ctx = mi_lumaeq.Context(0)
w, h, n = 3840, 2160 * 3 // 2, 16
a = torch.randint(0, 256, (n, h, w), dtype=torch.uint8, device="cuda")
b = a.clone(); b[:, ::7, ::5] += 1
stats = torch.zeros((n, 4), dtype=torch.int32, device="cuda")
diff = torch.empty_like(a)
for name, d in (("statistics only", None), ("with the difference image", diff)):
    for _ in range(3): ctx.analyze_diff_batch_dev(a, b, w, h, n, stats, threshold=1, diff=d)
    ctx.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): ctx.analyze_diff_batch_dev(a, b, w, h, n, stats, threshold=1, diff=d)
    ctx.synchronize(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    byt = (2 if d is None else 3) * w * h * n
    print(f"{name}: {dt * 1e6:7.1f} us per {n} frames, {byt / dt / 1e12:.2f} TB/s")
