"""A/B of a fused-kernel option in one process (interleaved rounds): python tools/fused_ab.py <option> [modes e.g. 0,1]"""
import sys, time, torch
sys.path.insert(0, "opencv-opencl_amd/python"); sys.path.insert(0, ".")
import mi_lumaeq
from mi_lumaeq import synth
opt = sys.argv[1] if len(sys.argv) > 1 else "fused_acquire"
MODES = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 1]
a = mi_lumaeq.Context(0)
for (w, h, B, uv) in ((3840, 2160, 64, 0), (3840, 2160, 64, 1), (3840, 2160, 8, 0), (1920, 1080, 256, 0), (3840, 2160, 1, 0)):
    d_in = synth.nv12_batch_torch(w, h, B, "D2", "cuda", seed=1)
    d_out = torch.empty_like(d_in)
    res = {m: [] for m in MODES}
    for rnd in range(9):
        for mode in MODES:
            a.set_option(opt, mode)
            for _ in range(2): a.equalize_hist_nv12_batch_dev(d_in, d_out, w, h, B, uv)
            a.synchronize()
            t0 = time.perf_counter()
            for _ in range(20): a.equalize_hist_nv12_batch_dev(d_in, d_out, w, h, B, uv)
            a.synchronize()
            res[mode].append((time.perf_counter() - t0) / 20 * 1e6)
    for mode in MODES:
        r = sorted(res[mode]); print(f"{w}x{h} B={B} uv={uv} {opt}={mode}: median {r[len(r)//2]:7.1f} us  min {r[0]:7.1f}  -> {B/(r[len(r)//2]*1e-6):9.0f} frames/s")
