"""Sweep the fused equalize kernel's knobs on one GPU (interleaved rounds in one process)."""
import sys, time, itertools
sys.path.insert(0, "opencv-opencl_amd/python"); sys.path.insert(0, ".")
import torch, mi_lumaeq
from mi_lumaeq import synth
w, h, B = 3840, 2160, int(sys.argv[1]) if len(sys.argv) > 1 else 64
dist = sys.argv[2] if len(sys.argv) > 2 else "D2"
ctx = mi_lumaeq.Context(0)
d_in = synth.nv12_batch_torch(w, h, B, dist, "cuda", seed=1)
d_out = torch.empty_like(d_in)
torch.cuda.synchronize()
combos = [dict(fused=0)] + [dict(fused=1, fused_vpt=v, fused_wgs_per_cu=g, fused_acquire=a)
                            for v in (8, 16, 20) for g in (2, 3, 4) for a in (1, 0)]
res = {i: [] for i in range(len(combos))}
for rnd in range(5):
    for i, cb in enumerate(combos):
        for k, v in cb.items():
            ctx.set_option(k, v)
        for _ in range(2):
            ctx.equalize_hist_nv12_batch_dev(d_in, d_out, w, h, B, 0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 10
        for _ in range(n):
            ctx.equalize_hist_nv12_batch_dev(d_in, d_out, w, h, B, 0)
        ctx.synchronize()
        res[i].append((time.perf_counter() - t0) / n * 1e6)
for i, cb in enumerate(combos):
    r = sorted(res[i])
    print(f"{str(cb):80s} median {r[len(r)//2]:8.1f} us/step  min {r[0]:8.1f}  -> {B / (r[len(r)//2] * 1e-6):10.0f} frames/s")
