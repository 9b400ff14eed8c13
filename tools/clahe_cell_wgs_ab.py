"""Fused CLAHE cell kernel: two passes vs 4 vs 5 resident workgroups per CU (docs/experiments.md R5.6).  python tools/clahe_cell_wgs_ab.py"""
import sys, time, torch
sys.path.insert(0, "opencv-opencl_amd/python"); sys.path.insert(0, ".")
import mi_lumaeq
from mi_lumaeq import synth
a = mi_lumaeq.Context(0)
a.set_option("clahe_single_read_min_frames", 1)
cases = [(3840, 2160, 8, 64), (1920, 1080, 4, 256), (3840, 2160, 8, 16), (3840, 2160, 8, 4), (1280, 720, 4, 576)]
modes = [("two passes", 0, 0), ("fused (two cells per 512-thread workgroup)", 1, 0)]
for (w, h, tiles, B) in cases:
    d_in = synth.nv12_batch_torch(w, h, B, "D2", "cuda", seed=1)
    d_out = torch.empty_like(d_in)
    res = {m[0]: [] for m in modes}
    for rnd in range(5):
        for name, sr, wgs in modes:
            a.set_option("clahe_single_read", sr)
            for _ in range(2): a.clahe_nv12_batch_dev(d_in, d_out, w, h, B, 0, 2.0, tiles, tiles)
            a.synchronize(); a.profile_read(True); a.set_profiling(True)
            for _ in range(10): a.clahe_nv12_batch_dev(d_in, d_out, w, h, B, 0, 2.0, tiles, tiles)
            a.synchronize(); a.set_profiling(False)
            p = a.profile_read(True)
            res[name].append(sum(v["total_ms"] / max(1, v["launches"]) for v in p.values() if v["launches"]) * 1e3)
    for name, _, _ in modes:
        r = sorted(res[name])
        us = r[len(r) // 2]
        print(f"{w}x{h} {tiles}x{tiles} B={B} {name}: kernels {us:7.1f} us per call  ({B / us * 1e6:9.0f} frames/s, whole path {(3.5 * w * h * B) / us / 8e6:.3f} of 8 TB/s)", flush=True)
print({k: a.get_stat(k) for k in ("clahe_fused_fallbacks", "clahe_cells_repaired")})
