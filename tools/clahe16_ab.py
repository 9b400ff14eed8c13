"""16-bit CLAHE A/B between library builds (GPU boxes differ by several per cent, so variants are compared inside ONE run):
    MI_LUMAEQ_LIB=<lib.so> MI_AB_OPTS="name=value ..." python tools/clahe16_ab.py [frames ...]
prints frames/s of mi_clahe_u16_batch_dev on 4K 12-bit noise (8x8 tiles, clip 2.0), 30 back-to-back calls, unprofiled."""
import os, sys, time, torch
sys.path.insert(0, "opencv-opencl_amd/python"); sys.path.insert(0, ".")
import mi_lumaeq
ctx = mi_lumaeq.Context(0)
opts = os.environ.get("MI_AB_OPTS", "")                      # e.g. MI_AB_OPTS="clahe16_strips=0"
for kv in opts.split():
    k, v = kv.split("="); ctx.set_option(k, int(v))
w, h = 3840, 2160
out = []
for n in [int(a) for a in sys.argv[1:]] or [16, 32]:
    s16 = torch.randint(0, 4096, (n, h, w), dtype=torch.int32, device="cuda").to(torch.int16)
    o16 = torch.empty_like(s16)
    for _ in range(3): ctx.clahe16_batch_dev(s16, o16, w, h, n, 2.0, 8, 8)
    ctx.synchronize()
    best = 0.0
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(30): ctx.clahe16_batch_dev(s16, o16, w, h, n, 2.0, 8, 8)
        ctx.synchronize()
        best = max(best, 30 * n / (time.perf_counter() - t0))
    out.append(f"{n} frames/call: {best:8.0f} frames/s ({best / n * 1e-3:.3f} k calls/s, {n / best * 1e6:.1f} us per call)")
    del s16, o16
print(os.path.basename(os.environ.get("MI_LUMAEQ_LIB", "libmi_lumaeq.so")), opts, " | ".join(out), flush=True)
