"""Raw PCIe ceilings on the GPU box: pinned H2D, D2H, and both at once on two streams (8.3 MB = one 4K Y plane, and 64 MB)."""
import time, torch
for mb in (8.2944, 64):
    n = int(mb * 1e6)
    h_in = torch.empty(n, dtype=torch.uint8).pin_memory()
    h_out = torch.empty(n, dtype=torch.uint8).pin_memory()
    d_a = torch.empty(n, dtype=torch.uint8, device="cuda")
    d_b = torch.empty(n, dtype=torch.uint8, device="cuda")
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    def run(fn, reps=30):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps
    def h2d():
        with torch.cuda.stream(s1): d_a.copy_(h_in, non_blocking=True)
    def d2h():
        with torch.cuda.stream(s2): h_out.copy_(d_b, non_blocking=True)
    def both():
        h2d(); d2h()
    for name, fn in (("h2d", h2d), ("d2h", d2h), ("both", both)):
        t = run(fn)
        print(f"{mb:8.1f} MB {name:5s} {t*1e3:7.3f} ms  {n/t/1e9:6.1f} GB/s per direction", flush=True)
# pageable
n = int(8.2944e6)
hp = torch.empty(n, dtype=torch.uint8)
d = torch.empty(n, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): d.copy_(hp)
torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 20
print(f"pageable h2d {t*1e3:.3f} ms {n/t/1e9:.1f} GB/s")
