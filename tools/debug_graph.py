import sys, numpy as np, torch
sys.path.insert(0, "opencv-opencl_amd/python"); sys.path.insert(0, ".")
import mi_lumaeq, oracle
from mi_lumaeq import synth
w, h, n = 1920, 1080, 6
c = mi_lumaeq.Context(0)
d_in = synth.nv12_batch_torch(w, h, n, "D2", "cuda:0", seed=11)
d_out = torch.zeros_like(d_in)
c.equalize_hist_nv12_batch_dev(d_in, d_out, w, h, n, 1); c.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    c.equalize_hist_nv12_batch_dev(d_in, d_out, w, h, n, 1, stream=torch.cuda.current_stream().cuda_stream)
for rep in range(3):
    d_in.copy_(synth.nv12_batch_torch(w, h, n, synth.DISTS[rep], "cuda:0", seed=100 + rep))
    d_out.zero_()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    try:
        c.synchronize()
        st = "ok"
    except Exception as e:
        st = str(e)
    src, out = d_in.cpu().numpy(), d_out.cpu().numpy()
    for k in range(n):
        ref = oracle.nv12_frame(src[k], w, h, uv_mode=1, op=0)
        bad = np.flatnonzero(out[k] != ref)
        print("rep", rep, "frame", k, "bad", bad.size, "first", bad[:3].tolist(), "Ybad", int((bad < w*h).sum()), "out zero frac", float((out[k] == 0).mean()), st)
