"""rocprofv3 target: the literal BASELINE config-5 path (NV12 -> BGR -> per-channel equalizeHist -> NV12), 32 x 4K frames."""
import sys, torch
sys.path.insert(0, "opencv-opencl_amd/python"); sys.path.insert(0, ".")
import mi_lumaeq
from mi_lumaeq import synth
ctx = mi_lumaeq.Context(0)
w, h, n = 3840, 2160, 32
dist = sys.argv[1] if len(sys.argv) > 1 else "D2"
nv = synth.nv12_batch_torch(w, h, n, dist, "cuda", seed=7)
out = torch.empty_like(nv)
torch.cuda.synchronize()
for _ in range(12):
    ctx.nv12_bgr_equalize_batch_dev(nv, out, w, h, n)
torch.cuda.synchronize()
