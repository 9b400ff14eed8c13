import sys, torch
sys.path.insert(0, "opencv-opencl_amd/python"); sys.path.insert(0, ".")
import mi_lumaeq
ctx = mi_lumaeq.Context(0)
w, h, n = 3840, 2160, 4
for name, s16 in (("uniform15", torch.randint(0, 32768, (n, h, w), dtype=torch.int16, device="cuda")),
                  ("narrow", torch.randint(1000, 1400, (n, h, w), dtype=torch.int16, device="cuda")),
                  ("const", torch.full((n, h, w), 777, dtype=torch.int16, device="cuda"))):
    o16 = torch.empty_like(s16)
    ctx.clahe16_batch_dev(s16, o16, w, h, n, 2.0, 8, 8); ctx.synchronize()
    ctx.profile_read(True); ctx.set_profiling(True)
    for _ in range(3): ctx.clahe16_batch_dev(s16, o16, w, h, n, 2.0, 8, 8)
    ctx.set_profiling(False)
    p = ctx.profile_read(True)
    print(name, {k: round(v["total_ms"] / max(v["launches"], 1), 3) for k, v in p.items() if v["launches"]})
