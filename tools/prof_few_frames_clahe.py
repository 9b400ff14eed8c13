"""CLAHE 8x8 on few device-resident frames: back-to-back us per call (see profiles/r03_m_clahe_splits.txt for the split-count probe)."""
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
for (w, h, tx, ty) in ((3840, 2160, 8, 8), (1920, 1080, 8, 8), (1280, 720, 8, 8), (3840, 2160, 16, 16), (3840, 2160, 2, 2), (3840, 2160, 1, 1)):
    line = []
    for n in (1, 2, 4, 8, 16):
        nv = synth.nv12_batch_torch(w, h, n, "D2", "cuda", seed=7); out = torch.empty_like(nv)
        line.append(f"n={n}: {rate(lambda: ctx.clahe_nv12_batch_dev(nv, out, w, h, n, 0, 2.0, tx, ty)):.1f}")
    print(f"CLAHE {w}x{h} {tx}x{ty} us/call  " + "  ".join(line), flush=True)
