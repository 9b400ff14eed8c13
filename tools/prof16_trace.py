"""One content kind of 16-bit CLAHE for a rocprofv3 kernel trace:  python tools/prof16_trace.py <kind> [frames] [clahe16_wide]
kind: 12bit | 14bit | full | ramp | hot"""
import sys, torch
sys.path.insert(0, "opencv-opencl_amd/python"); sys.path.insert(0, ".")
import mi_lumaeq
kind = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 16; wide = int(sys.argv[3]) if len(sys.argv) > 3 else 2
ctx = mi_lumaeq.Context(0); ctx.set_option("clahe16_wide", wide)
w, h = 3840, 2160
def u16(lo, hi): return torch.randint(lo, hi, (n, h, w), dtype=torch.int32, device="cuda").to(torch.int16)
if kind == "12bit": s = u16(0, 4096)
elif kind == "14bit": s = u16(0, 16384)
elif kind == "full": s = u16(0, 65536)
elif kind == "hot": s = u16(0, 4096); s[:, 1000, 2000] = -1
else: s = ((torch.arange(h, device="cuda").view(1, h, 1) * 12 + torch.arange(w, device="cuda").view(1, 1, w) * 10 + torch.randint(0, 512, (n, h, w), device="cuda")) % 65536).to(torch.int32).to(torch.int16)
o = torch.empty_like(s)
for _ in range(12): ctx.clahe16_batch_dev(s, o, w, h, n, 2.0, 8, 8)
ctx.synchronize()
