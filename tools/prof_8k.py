import sys, time, torch
sys.path.insert(0, "opencv-opencl_amd/python"); sys.path.insert(0, ".")
import mi_lumaeq, oracle, numpy as np
from mi_lumaeq import synth, xfer
ctx = mi_lumaeq.Context(0)
w, h, n = 7680, 4320, 16
nv = synth.nv12_batch_torch(w, h, n, "D2", "cuda", seed=7)
out = torch.empty_like(nv)
for fused in (1, 0):
    ctx.set_option("fused", fused)
    for _ in range(3): ctx.equalize_hist_nv12_batch_dev(nv, out, w, h, n, mi_lumaeq.UV_FILL128)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): ctx.equalize_hist_nv12_batch_dev(nv, out, w, h, n, mi_lumaeq.UV_FILL128)
    ctx.synchronize(); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f"8K fused={fused}: {dt*1e6:.1f} us per {n} frames = {n/dt:.0f} frames/s, {3.5*w*h*n/dt/1e12:.2f} TB/s algorithmic", flush=True)
ctx.set_option("fused", 1)
ctx.equalize_hist_nv12_batch_dev(nv, out, w, h, n, mi_lumaeq.UV_FILL128); ctx.synchronize()
y0 = xfer.to_host(nv[0, :w*h]).reshape(h, w)
print("8K fused parity:", np.array_equal(xfer.to_host(out[0, :w*h]).reshape(h, w), oracle.equalize_hist(y0)))
