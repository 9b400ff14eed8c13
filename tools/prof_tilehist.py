"""Stand-alone timing of the CLAHE stages (no neighbours in the stream): tile histograms + LUTs vs the whole op."""
import sys, time, torch
sys.path.insert(0, "opencv-opencl_amd/python"); sys.path.insert(0, ".")
import mi_lumaeq
from mi_lumaeq import synth
ctx = mi_lumaeq.Context(0)
w, h, n = 3840, 2160, 64
nv = synth.nv12_batch_torch(w, h, n, "D2", "cuda", seed=7)
out = torch.empty_like(nv)
luts = torch.empty((n, 64, 256), dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
def t(fn, reps=30):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
fs = w * h * 3 // 2
print("tile_luts only (hist+lut) us:", round(t(lambda: ctx.clahe_tile_luts_batch_dev(nv, w, h, n, 2.0, 8, 8, luts, src_step=w, src_frame=fs)), 1))
print("whole clahe nv12 us:", round(t(lambda: ctx.clahe_nv12_batch_dev(nv, out, w, h, n, mi_lumaeq.UV_FILL128, 2.0, 8, 8)), 1))
hist = torch.empty((n, 256), dtype=torch.int32, device="cuda")
print("hist only (equalize stage api: hist_partial + reduce) us:", round(t(lambda: ctx.hist_batch_dev(nv, w, h, n, hist, src_step=w, src_frame=fs)), 1))
