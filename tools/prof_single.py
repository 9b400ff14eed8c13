"""Single 4K frame, device-resident: fused kernel duration (HIP events) and back-to-back call rate vs slice size."""
import sys, time, torch
sys.path.insert(0, "opencv-opencl_amd/python"); sys.path.insert(0, ".")
import mi_lumaeq
from mi_lumaeq import synth
ctx = mi_lumaeq.Context(0)
w, h = 3840, 2160
for n in (1, 2, 4):
    nv = synth.nv12_batch_torch(w, h, n, "D2", "cuda", seed=7)
    out = torch.empty_like(nv)
    for vpt in (0, 20, 16, 8):
        ctx.set_option("fused_vpt", vpt)
        for _ in range(20): ctx.equalize_hist_nv12_batch_dev(nv, out, w, h, n, mi_lumaeq.UV_FILL128)
        torch.cuda.synchronize()
        ctx.profile_read(True); ctx.set_profiling(True)
        for _ in range(50): ctx.equalize_hist_nv12_batch_dev(nv, out, w, h, n, mi_lumaeq.UV_FILL128)
        torch.cuda.synchronize(); ctx.set_profiling(False)
        p = ctx.profile_read(True)["equalize_fused_kernel"]
        t0 = time.perf_counter()
        for _ in range(300): ctx.equalize_hist_nv12_batch_dev(nv, out, w, h, n, mi_lumaeq.UV_FILL128)
        torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 300
        print(f"frames={n} vpt={vpt or 'auto':>4}: kernel {p['total_ms'] / max(p['launches'], 1) * 1e3:7.1f} us   back-to-back {wall * 1e6:7.1f} us/call", flush=True)

# where a single frame's ~22 us go: Y only (no UV tickets), the three-kernel path, and a bare launch
nv = synth.nv12_batch_torch(w, h, 1, "D2", "cuda", seed=7)
out = torch.empty_like(nv)
ctx.set_option("fused_vpt", 0)
def rate(fn, reps=300):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
print("fused NV12 (Y + UV fill) us/call:", round(rate(lambda: ctx.equalize_hist_nv12_batch_dev(nv, out, w, h, 1, mi_lumaeq.UV_FILL128)), 1))
print("fused Y plane only       us/call:", round(rate(lambda: ctx.equalize_hist_batch_dev(nv, out, w, h, 1)), 1))
ctx.set_option("fused", 0)
print("three-kernel NV12        us/call:", round(rate(lambda: ctx.equalize_hist_nv12_batch_dev(nv, out, w, h, 1, mi_lumaeq.UV_FILL128)), 1))
ctx.set_option("fused", 1)
x = torch.zeros(64, device="cuda")
print("bare torch kernel        us/call:", round(rate(lambda: x.add_(1.0)), 1))
hist = torch.empty((1, 256), dtype=torch.int32, device="cuda")
print("hist stage only (2 launches) us/call:", round(rate(lambda: ctx.hist_batch_dev(nv, w, h, 1, hist)), 1))
