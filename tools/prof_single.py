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
