"""A handful of synchronous host-form calls for a rocprofv3 timeline (--hip-trace --memory-copy-trace --kernel-trace):
    rocprofv3 --hip-trace --memory-copy-trace --kernel-trace --output-format csv -d gpurun_out/trace -- python3 tools/host_form_trace.py"""
import sys, time
sys.path.insert(0, "opencv-opencl_amd/python"); sys.path.insert(0, ".")
import numpy as np, mi_lumaeq
from mi_lumaeq import synth
w, h = 3840, 2160
threads = int(sys.argv[1]) if len(sys.argv) > 1 else 2
ctx = mi_lumaeq.Context(0)
ctx.set_option("host_copy_threads", threads)
y = synth.y_plane(w, h, "D2", 1)
dst = np.empty_like(y)
for _ in range(12):
    t0 = time.perf_counter()
    ctx.equalize_hist(y, dst)
    print(f"call {(time.perf_counter() - t0) * 1e3:.3f} ms", flush=True)
    time.sleep(0.002)
