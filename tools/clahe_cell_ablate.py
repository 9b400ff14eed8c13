"""Stage-1 ablations of the cell kernel (docs/experiments.md R5.5): which of its phases costs what.  python tools/clahe_cell_ablate.py"""
import sys, time, torch
sys.path.insert(0, "opencv-opencl_amd/python"); sys.path.insert(0, ".")
import mi_lumaeq
from mi_lumaeq import synth
a = mi_lumaeq.Context(0)
w, h, B = 3840, 2160, 64
d_in = synth.nv12_batch_torch(w, h, B, "D2", "cuda", seed=1)
d_out = torch.empty_like(d_in)
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 2
a.set_option("clahe_single_read", mode)
a.set_option("clahe_single_read_min_frames", 1)
if mode == 1:
    names = {0: "fused, full hand-off", 64: "fused, LUTs from a tile-histogram pass (no waits; partials still published, tiles still computed)", 192: "fused, no hand-off at all"}
else:
  names = {0: "full", 8: "full, XCD-aware", 7: "loads + stores only", 15: "loads + stores only, XCD-aware", 23: "loads only", 31: "loads only, XCD-aware",
         39: "stores only", 47: "stores only, XCD-aware", 9: "no histogram, XCD-aware", 10: "no blend, XCD-aware"}
res = {v: [] for v in names}
for rnd in range(5):
    for v in names:
        a.set_option("clahe_cell_variant", v)
        for _ in range(2): a.clahe_nv12_batch_dev(d_in, d_out, w, h, B, 0, 2.0, 8, 8)
        a.synchronize(); a.profile_read(True); a.set_profiling(True)
        for _ in range(10): a.clahe_nv12_batch_dev(d_in, d_out, w, h, B, 0, 2.0, 8, 8)
        a.synchronize(); a.set_profiling(False)
        p = a.profile_read(True)["clahe_interp_kernel"]
        res[v].append(p["total_ms"] / p["launches"] * 1e3)
for v, n in names.items():
    r = sorted(res[v])
    print(f"variant {v} ({n}): cell kernel median {r[len(r)//2]:.1f} us  (min {r[0]:.1f})", flush=True)
