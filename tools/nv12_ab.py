"""A/B of an option on the literal config-5 path (NV12 -> BGR -> per-channel equalizeHist -> NV12), interleaved rounds in one process:
    python tools/nv12_ab.py <option> [modes e.g. 0,1] [frames]"""
import sys, time, torch
sys.path.insert(0, "opencv-opencl_amd/python"); sys.path.insert(0, ".")
import mi_lumaeq
from mi_lumaeq import synth
opt = sys.argv[1] if len(sys.argv) > 1 else "nv12_hist_wide"
MODES = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 1]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 32
ctx = mi_lumaeq.Context(0)
for (w, h) in ((3840, 2160), (1920, 1080)):
    nv = synth.nv12_batch_torch(w, h, n, "D2", "cuda", seed=7)
    out = torch.empty_like(nv)
    wall = {m: [] for m in MODES}
    for rnd in range(7):
        for m in MODES:
            ctx.set_option(opt, m)
            for _ in range(2): ctx.nv12_bgr_equalize_batch_dev(nv, out, w, h, n)
            ctx.synchronize(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10): ctx.nv12_bgr_equalize_batch_dev(nv, out, w, h, n)
            ctx.synchronize(); torch.cuda.synchronize()
            wall[m].append((time.perf_counter() - t0) / 10 * 1e6)
    for m in MODES:
        r = sorted(wall[m])
        print(f"{w}x{h} B={n} {opt}={m}: wall median {r[len(r)//2]:7.1f} us -> {n/(r[len(r)//2]*1e-6):9.0f} frames/s", flush=True)
