"""Is the second read of a two-pass op served by the 256 MiB Infinity Cache when the first pass has JUST read the same frames?
Same launches, same sizes -- only what was read immediately before differs:
    flush   : the tile-histogram pass over 40 OTHER 4K frames (332 MB: everything else leaves the cache)
    hot     : tile histograms over frames A (n frames), then LUT apply on A         -> A's second read should come from the cache
    cold    : tile histograms over frames B (n other frames), then LUT apply on A   -> A comes from HBM
The LUT-apply launch (read Y, write Y: the interpolation's traffic shape) is timed by the events of its own dispatch.
    python tools/mall_probe.py"""
import sys, torch
sys.path.insert(0, "opencv-opencl_amd/python"); sys.path.insert(0, ".")
import mi_lumaeq
from mi_lumaeq import synth
w, h = 3840, 2160
fs = w * h * 3 // 2
ctx = mi_lumaeq.Context(0)
d_in = synth.nv12_batch_torch(w, h, 96, "D2", "cuda", seed=1)
d_out = torch.empty((32, fs), dtype=torch.uint8, device="cuda")
lut = torch.arange(256, dtype=torch.uint8, device="cuda").repeat(32, 1).contiguous()
luts = torch.empty((40, 64, 256), dtype=torch.uint8, device="cuda")
base = d_in.data_ptr()
A, Bf, FL = base, base + 32 * fs, base + 56 * fs                # frames 0.., 32.., 56..95
def hist(ptr, n): ctx.clahe_tile_luts_batch_dev(ptr, w, h, n, 2.0, 8, 8, luts, src_frame=fs)
def apply(ptr, n): ctx.lut_apply_batch_dev(ptr, d_out, w, h, n, lut, src_frame=fs, dst_frame=fs)
print("4K frames; LUT apply of n frames (read n x 8.3 MB, write n x 8.3 MB) right after the tile-histogram pass over the SAME frames (hot) "
      "or over n OTHER frames (cold); 332 MB of other reads before each pair")
for n in (2, 4, 8, 12, 16, 24):
    res = {}
    for case in ("hot", "cold", "hot", "cold"):
        ts = []
        for rep in range(12):
            hist(FL, 40)                                         # flush
            hist(A if case == "hot" else Bf, n)
            ctx.synchronize()
            ctx.profile_read(True); ctx.set_profiling(True)
            apply(A, n)
            ctx.set_profiling(False)
            ctx.synchronize()
            p = ctx.profile_read(True)["lut_apply_kernel"]
            ts.append(p["total_ms"] / max(1, p["launches"]) * 1e3)
        ts.sort()
        res.setdefault(case, []).append(ts[len(ts) // 2])
    hot, cold = min(res["hot"]), min(res["cold"])
    mb = 2 * n * w * h / 1e6
    print(f"n={n:3d}: apply after hist of the same frames {hot:7.1f} us ({mb / hot:5.2f} TB/s)   after hist of other frames {cold:7.1f} us ({mb / cold:5.2f} TB/s)   "
          f"hot / cold = {hot / cold:.3f}", flush=True)
# the other order of the same question: histogram pass right after something WROTE the frames (does a freshly written frame get read from the cache?)
ctx.close()
