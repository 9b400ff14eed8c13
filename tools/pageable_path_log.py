"""Which path does the HIP runtime take for a PAGEABLE hipMemcpyAsync (option host_direct = 1, the library's old default)?
Run as   AMD_LOG_LEVEL=4 python tools/pageable_path_log.py 2> log   and look for the blit / pin / staging lines.
One host-form call per size, no fault injection."""
import sys
sys.path.insert(0, "opencv-opencl_amd/python"); sys.path.insert(0, ".")
import numpy as np
import mi_lumaeq
from mi_lumaeq import synth
ctx = mi_lumaeq.Context(0)
ctx.set_option("host_direct", 1)
for (w, h) in ((1920, 1080), (3840, 2160)):
    y = synth.y_plane(w, h, "D2", 3)
    sys.stderr.write(f"=== MARK begin {w}x{h}\n"); sys.stderr.flush()
    out = ctx.equalize_hist(y)
    sys.stderr.write(f"=== MARK end {w}x{h}\n"); sys.stderr.flush()
print("done")
