"""16-bit CLAHE, content wider than 8192 values: option clahe16_wide = 0 (round-3 paths) against 1 (round 6) in ONE process, unprofiled,
alternating: python tools/clahe16_wide_ab.py [frames per call]   (4K, 8x8, clip 2.0; frames/s, best of three runs of ten calls)"""
import sys, time, torch
sys.path.insert(0, "opencv-opencl_amd/python"); sys.path.insert(0, ".")
import mi_lumaeq
ctx = mi_lumaeq.Context(0)
w, h = 3840, 2160
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
def u16(lo, hi): return torch.randint(lo, hi, (n, h, w), dtype=torch.int32, device="cuda").to(torch.int16)
def hot():
    s = u16(0, 4096); s[:, 1000, 2000] = -1; return s
def ramp():
    return ((torch.arange(h, device="cuda").view(1, h, 1) * 12 + torch.arange(w, device="cuda").view(1, 1, w) * 10
             + torch.randint(0, 512, (n, h, w), device="cuda")) % 65536).to(torch.int32).to(torch.int16)
cases = (("12-bit", lambda: u16(0, 4096)), ("10-bit << 6 (P010)", lambda: (torch.randint(0, 1024, (n, h, w), dtype=torch.int32, device="cuda") << 6).to(torch.int16)),
         ("13-bit", lambda: u16(0, 8192)), ("14-bit", lambda: u16(0, 16384)),
         ("14-bit << 2", lambda: (torch.randint(0, 16384, (n, h, w), dtype=torch.int32, device="cuda") << 2).to(torch.int16)),
         ("12-bit + one hot pixel", hot), ("15-bit", lambda: u16(0, 32768)), ("full range", lambda: u16(0, 65536)), ("full range, smooth ramp", ramp))
cases = cases + (("12-bit, IN PLACE (+ a copy)", "inplace12"), ("14-bit, IN PLACE (+ a copy)", "inplace14"), ("12-bit + hot pixel, IN PLACE (+ a copy)", "inplacehot"),
                 ("full range, IN PLACE (+ a copy)", "inplace16"))
for name, make in cases:
    inplace = isinstance(make, str)
    if inplace:                                                  # the call overwrites its input with a full-range result, so the input is copied back
        make = {"inplace12": lambda: u16(0, 4096), "inplace14": lambda: u16(0, 16384), "inplacehot": hot, "inplace16": lambda: u16(0, 65536)}[make]     # before every call: both columns include that copy
    s16 = make(); o16 = torch.empty_like(s16)
    work = torch.empty_like(s16) if inplace else None
    def call():
        if inplace:
            work.copy_(s16)
            ctx.clahe16_batch_dev(work, work, w, h, n, 2.0, 8, 8, stream=torch.cuda.current_stream().cuda_stream)
        else:
            ctx.clahe16_batch_dev(s16, o16, w, h, n, 2.0, 8, 8, stream=torch.cuda.current_stream().cuda_stream)
    best = {0: 0.0, 1: 0.0}
    for rep in range(3):
        for wide in (0, 1):
            ctx.set_option("clahe16_wide", wide)
            for _ in range(2): call()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10): call()
            torch.cuda.synchronize()
            best[wide] = max(best[wide], 10 * n / (time.perf_counter() - t0))
    print(f"{name:40s} {n} per call: round-3 paths {best[0]:9.0f} frames/s ({n / best[0] * 1e6:7.1f} us per call)   clahe16_wide {best[1]:9.0f} frames/s "
          f"({n / best[1] * 1e6:7.1f} us per call)   x{best[1] / best[0]:.2f}", flush=True)
    del s16, o16, work
