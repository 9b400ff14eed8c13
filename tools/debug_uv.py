import sys, numpy as np, torch
sys.path.insert(0, "opencv-opencl_amd/python"); sys.path.insert(0, ".")
import mi_lumaeq
from mi_lumaeq import synth
ctx = mi_lumaeq.Context(0)
w, h = 3840, 2160
for n in (1, 2, 7, 8, 9, 16):
    d_in = synth.nv12_batch_torch(w, h, n, "D1", "cuda:0", seed=1)
    d_out = torch.full_like(d_in, 7)
    torch.cuda.synchronize()
    ctx.equalize_hist_nv12_batch_dev(d_in, d_out, w, h, n, 1, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    uv_in, uv_out = d_in[:, w*h:], d_out[:, w*h:]
    bad = (uv_in != uv_out)
    print("n", n, "bad per frame", bad.sum(dim=1).tolist())
    if bad.any():
        f = int(torch.nonzero(bad.any(dim=1))[0])
        idx = torch.nonzero(bad[f]).view(-1)
        print("  frame", f, "first bad", int(idx[0]), "last bad", int(idx[-1]), "count", idx.numel(), "vals", uv_out[f, idx[:4]].tolist())
    ysame = (d_out[:, :w*h] == 7).all(dim=1).tolist()
    print("  Y untouched:", ysame)
