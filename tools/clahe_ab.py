"""A/B of a CLAHE option in one process (interleaved rounds), per-kernel HIP-event averages:
    python tools/clahe_ab.py <option> [modes e.g. 0,1] [batch] [cases e.g. 1280x720x8x576,3840x2160x8]"""
import sys, time, torch
sys.path.insert(0, "opencv-opencl_amd/python"); sys.path.insert(0, ".")
import mi_lumaeq
from mi_lumaeq import synth
opt = sys.argv[1] if len(sys.argv) > 1 else "clahe_xcd_map"
MODES = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 1]
B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
a = mi_lumaeq.Context(0)
import os
for kv in os.environ.get("MI_AB_SET", "").split(","):            # further options for every mode, e.g. MI_AB_SET=clahe_fp_contract=1
    if "=" in kv:
        a.set_option(kv.split("=")[0], int(kv.split("=")[1]))
CASES = [(3840, 2160, 8, B), (1920, 1080, 8, B), (3840, 2160, 16, B)]
if len(sys.argv) > 4:
    CASES = []
    for c in sys.argv[4].split(","):
        f = [int(x) for x in c.split("x")]
        CASES.append((f[0], f[1], f[2], f[3] if len(f) > 3 else B))
for (w, h, tiles, B) in CASES:
    d_in = synth.nv12_batch_torch(w, h, B, "D2", "cuda", seed=1)
    d_out = torch.empty_like(d_in)
    wall = {m: [] for m in MODES}
    kern = {m: {} for m in MODES}
    for rnd in range(7):
        for mode in MODES:
            a.set_option(opt, mode)
            for _ in range(2): a.clahe_nv12_batch_dev(d_in, d_out, w, h, B, 0, 2.0, tiles, tiles)
            a.synchronize()
            a.profile_read(True); a.set_profiling(True)
            t0 = time.perf_counter()
            for _ in range(10): a.clahe_nv12_batch_dev(d_in, d_out, w, h, B, 0, 2.0, tiles, tiles)
            a.synchronize()
            wall[mode].append((time.perf_counter() - t0) / 10 * 1e6)
            a.set_profiling(False)
            for k, v in a.profile_read(True).items():
                if v["launches"]: kern[mode].setdefault(k, []).append(v["total_ms"] / v["launches"] * 1e3)
    for mode in MODES:
        r = sorted(wall[mode])
        ks = {k: round(sorted(v)[len(v) // 2], 1) for k, v in kern[mode].items()}
        print(f"{w}x{h} {tiles}x{tiles} B={B} {opt}={mode}: wall median {r[len(r)//2]:7.1f} us -> {B/(r[len(r)//2]*1e-6):9.0f} frames/s  kernels(us) {ks}", flush=True)
