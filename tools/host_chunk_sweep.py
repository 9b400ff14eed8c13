"""Chunk geometry of the staged host form (MI_LUMAEQ_HOST_CHUNK_KB x MI_LUMAEQ_HOST_CHUNK_RAMP), one process per setting."""
import os, subprocess, sys
code = r'''
import sys, time
sys.path.insert(0, "opencv-opencl_amd/python"); sys.path.insert(0, ".")
import numpy as np, mi_lumaeq
from mi_lumaeq import synth
w, h = 3840, 2160
ctx = mi_lumaeq.Context(0)
y = synth.y_plane(w, h, "D2", 1); dst = np.empty_like(y)
for _ in range(8): ctx.equalize_hist(y, dst)
ts = []
for _ in range(60):
    t0 = time.perf_counter(); ctx.equalize_hist(y, dst); ts.append((time.perf_counter() - t0) * 1e3)
ts.sort(); print(f"p50 {ts[30]:.3f} p10 {ts[6]:.3f}")
'''
for kb in (1024, 2048, 3072, 4096):
    for ramp in (1, 2, 4, 8, 16):
        env = dict(os.environ, MI_LUMAEQ_HOST_CHUNK_KB=str(kb), MI_LUMAEQ_HOST_CHUNK_RAMP=str(ramp))
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        print(f"chunk {kb:5d} KiB ramp {ramp:2d}: {r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-200:]}", flush=True)
