import sys, time, torch
sys.path.insert(0, "opencv-opencl_amd/python"); sys.path.insert(0, ".")
import mi_lumaeq
from mi_lumaeq import synth
w, h = 3840, 2160
a = mi_lumaeq.Context(0)
for B in (64, 32, 16):
    d_in = synth.nv12_batch_torch(w, h, B, "D2", "cuda", seed=1)
    d_out = torch.empty_like(d_in)
    res = {0: [], 1: []}
    for rnd in range(7):
        for mode in (0, 1):
            a.set_option("clahe_two_streams", mode)
            for _ in range(2): a.clahe_nv12_batch_dev(d_in, d_out, w, h, B, 0, 2.0, 8, 8)
            a.synchronize()
            t0 = time.perf_counter()
            for _ in range(10): a.clahe_nv12_batch_dev(d_in, d_out, w, h, B, 0, 2.0, 8, 8)
            a.synchronize()
            res[mode].append((time.perf_counter() - t0) / 10 * 1e6)
    for mode in (0, 1):
        r = sorted(res[mode]); print(f"B={B} two_streams={mode}: median {r[len(r)//2]:7.1f} us  min {r[0]:7.1f}  -> {B/(r[len(r)//2]*1e-6):9.0f} frames/s")
