"""Per-call cost of the synchronous host-pointer form (cv::Mat in -> cv::Mat out, PCIe inclusive) on a 4K Y plane:
unpinned planes packed by one thread or by the caller + the context's helper thread, and pinned planes.
    python tools/host_form_ab.py [width height]"""
import sys, time
sys.path.insert(0, "opencv-opencl_amd/python"); sys.path.insert(0, ".")
import numpy as np, torch, mi_lumaeq
from mi_lumaeq import synth
w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
ctx = mi_lumaeq.Context(0)
y = synth.y_plane(w, h, "D2", 1)
dst = np.empty_like(y)

def timeit(fn, n=40):
    for _ in range(5):
        fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[len(ts) // 10], ts[-len(ts) // 10 - 1]

for streams, threads in ((1, 1), (1, 2), (2, 1), (2, 2), (1, 1), (2, 2)):
    if True:
        ctx.set_option("host_copy_threads", threads)
        ctx.set_option("host_copy_streams", streams)
        s0 = ctx.get_stat("host_copies_shared")
        p50, p10, p90 = timeit(lambda: ctx.equalize_hist(y, dst))
        print(f"unpinned {w}x{h} equalizeHist, copy streams={streams} threads={threads}: p50 {p50:.3f} ms (p10 {p10:.3f}, p90 {p90:.3f}); "
              f"copies shared with the helper: {ctx.get_stat('host_copies_shared') - s0}", flush=True)
        p50, p10, p90 = timeit(lambda: ctx.clahe(y, 2.0, 8, 8, dst))
        print(f"unpinned {w}x{h} CLAHE 8x8,    copy streams={streams} threads={threads}: p50 {p50:.3f} ms (p10 {p10:.3f}, p90 {p90:.3f})", flush=True)
py, pd = torch.from_numpy(y.copy()).pin_memory(), torch.empty((h, w), dtype=torch.uint8).pin_memory()
p50, p10, p90 = timeit(lambda: ctx.equalize_hist(py.numpy(), pd.numpy()))
print(f"pinned   {w}x{h} equalizeHist: p50 {p50:.3f} ms (p10 {p10:.3f}, p90 {p90:.3f})")
# a caller that works at 60 fps: one call every 16.7 ms (the helper sleeps in between and is woken per call)
ctx.set_option("host_copy_threads", 2); ctx.set_option("host_copy_streams", 2)
ts = []
for k in range(60):
    time.sleep(1 / 60)
    t0 = time.perf_counter(); ctx.equalize_hist(y, dst); ts.append((time.perf_counter() - t0) * 1e3)
ts.sort()
print(f"unpinned, one call per 16.7 ms, 2 threads: p50 {ts[30]:.3f} ms p90 {ts[54]:.3f}")
ctx.set_option("host_copy_threads", 1)
ts = []
for k in range(60):
    time.sleep(1 / 60)
    t0 = time.perf_counter(); ctx.equalize_hist(y, dst); ts.append((time.perf_counter() - t0) * 1e3)
ts.sort()
print(f"unpinned, one call per 16.7 ms, 1 thread:  p50 {ts[30]:.3f} ms p90 {ts[54]:.3f}")
