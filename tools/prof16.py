"""16-bit CLAHE: per-kernel HIP-event averages and frames/s by content, 4K 8x8 clip 2.0:  python tools/prof16.py [frames]"""
import sys, time, torch
sys.path.insert(0, "opencv-opencl_amd/python"); sys.path.insert(0, ".")
import mi_lumaeq
ctx = mi_lumaeq.Context(0)
w, h = 3840, 2160
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
only = sys.argv[2] if len(sys.argv) > 2 else ""            # "12bit": just the 12-bit case (profiling runs)
for kv in sys.argv[3:]:                                      # name=value options, e.g. clahe16_fast12=0
    k, v = kv.split("="); ctx.set_option(k, int(v)); print("option", k, "=", v)
def u16(lo, hi): return torch.randint(lo, hi, (n, h, w), dtype=torch.int32, device="cuda").to(torch.int16)   # bit pattern of the ushort
cases = (("12-bit 0..4095", u16(0, 4096)), ("10-bit 0..1023", u16(0, 1024)), ("narrow 1000..1399", u16(1000, 1400)), ("13-bit 0..8191", u16(0, 8192)), ("14-bit 0..16383", u16(0, 16384)),
         ("14-bit << 2 (MSB-aligned)", (torch.randint(0, 16384, (n, h, w), dtype=torch.int32, device="cuda") << 2).to(torch.int16)),
         ("10-bit << 6 (P010)", (torch.randint(0, 1024, (n, h, w), dtype=torch.int32, device="cuda") << 6).to(torch.int16)),
         ("12-bit << 4 (MSB-aligned)", (torch.randint(0, 4096, (n, h, w), dtype=torch.int32, device="cuda") << 4).to(torch.int16)),
         ("12-bit, black bars at level 256", "bars"),
         ("P010 with bars at 64 << 6", "p010bars"),
         ("12-bit + one 65535 pixel per frame", None),
         ("15-bit 0..32767", u16(0, 32768)), ("16-bit full", u16(0, 65536)), ("const 777", torch.full((n, h, w), 777, dtype=torch.int16, device="cuda")),
         # full 16-bit range over the frame, locally smooth (a diagonal ramp + 9 bits of noise): what a 16-bit photograph looks like to the windows
         ("16-bit smooth ramp", ((torch.arange(h, device="cuda").view(1, h, 1) * 12 + torch.arange(w, device="cuda").view(1, 1, w) * 10
                                  + torch.randint(0, 512, (n, h, w), device="cuda")) % 65536).to(torch.int32).to(torch.int16)))
for name, s16 in cases:
    if only == "12bit" and not name.startswith("12-bit 0"):
        continue
    if isinstance(s16, str):                                     # letterboxed ordinary content: flat tiles whose value is a multiple of 256
        if s16 == "bars":
            s16 = u16(0, 4096); s16[:, : h // 8] = 256; s16[:, -(h // 8):] = 256
        else:                                                    # the same in P010: picture = 10 bits << 6, bars = 64 << 6 (trailing zeros 12)
            s16 = (torch.randint(64, 941, (n, h, w), dtype=torch.int32, device="cuda") << 6).to(torch.int16); s16[:, : h // 8] = 4096; s16[:, -(h // 8):] = 4096
    if s16 is None:                                              # a hot pixel: one tile per frame loses its bet and the frame's range is the whole word
        s16 = u16(0, 4096); s16[:, 1000, 2000] = -1
    o16 = torch.empty_like(s16)
    for _ in range(2): ctx.clahe16_batch_dev(s16, o16, w, h, n, 2.0, 8, 8)
    ctx.synchronize()
    ctx.profile_read(True); ctx.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(5): ctx.clahe16_batch_dev(s16, o16, w, h, n, 2.0, 8, 8)
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / 5
    ctx.set_profiling(False)
    p = ctx.profile_read(True)
    print(f"{name:20s} {n / dt:9.0f} frames/s  ({3 * w * h * 2 * n / dt / 1e12:.2f} TB/s of 3*W*H*2 B)  kernels(us):",
          {k: round(v["total_ms"] / max(v["launches"], 1) * 1e3, 1) for k, v in p.items() if v["launches"]}, flush=True)
    del s16, o16
