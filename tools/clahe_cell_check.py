"""Bit-exactness of the single-read CLAHE cell kernel (option clahe_single_read, docs/experiments.md R5.4/R5.5) against the oracle,
then tools/clahe_ab.py-style timing is run separately.   python tools/clahe_cell_check.py"""
import sys
import numpy as np
import torch
sys.path.insert(0, "opencv-opencl_amd/python"); sys.path.insert(0, ".")
import mi_lumaeq, oracle
from mi_lumaeq import synth, xfer
ctx = mi_lumaeq.Context(0)
ctx.set_option("clahe_single_read", int(sys.argv[1]) if len(sys.argv) > 1 else 1)
ctx.set_option("clahe_single_read_min_frames", 1)
bad = 0
for (w, h, tx, ty, clip, B, uvm) in [(3840, 2160, 8, 8, 2.0, 5, 0), (3840, 2160, 8, 8, 40.0, 3, 1), (1920, 1080, 4, 4, 3.0, 6, 0), (1280, 720, 4, 4, 2.0, 4, 0),
                                     (3840, 2160, 4, 8, 2.0, 2, 0), (1920, 1080, 8, 8, 2.0, 3, 0), (640, 360, 2, 2, 0.0, 3, 1), (256, 64, 4, 4, 2.0, 7, 0)]:
    frames = np.stack([synth.nv12_frame(w, h, synth.DISTS[k % 5], 300 + k) for k in range(B)])
    d_in = xfer.to_device(frames)
    d_out = torch.zeros_like(d_in)
    ctx.profile_read(True); ctx.set_profiling(True)
    ctx.clahe_nv12_batch_dev(d_in, d_out, w, h, B, uvm, clip, tx, ty)
    ctx.synchronize(); ctx.set_profiling(False)
    kern = {k: v["launches"] for k, v in ctx.profile_read(True).items() if v["launches"]}
    out = xfer.to_host(d_out)
    ok = all(np.array_equal(out[k], oracle.nv12_frame(frames[k], w, h, uv_mode=uvm, op=1, clip_limit=clip, tiles_x=tx, tiles_y=ty)) for k in range(B))
    bad += not ok
    print(f"{w}x{h} {tx}x{ty} clip {clip} B={B} uv={uvm}: {'bit-exact' if ok else 'MISMATCH'}  kernels {kern}", flush=True)
    # in place
    d_io = d_in.clone()
    ctx.clahe_nv12_batch_dev(d_io, d_io, w, h, B, uvm, clip, tx, ty)
    ctx.synchronize()
    ok = np.array_equal(xfer.to_host(d_io), out)
    bad += not ok
    print(f"    in place: {'same bytes' if ok else 'MISMATCH'}", flush=True)
print({k: ctx.get_stat(k) for k in ("clahe_fused_fallbacks", "clahe_cells_repaired", "clahe_fused_last_status", "clahe_fused_demotions")})
bad += ctx.get_stat("clahe_fused_fallbacks") != 0
sys.exit(1 if bad else 0)
