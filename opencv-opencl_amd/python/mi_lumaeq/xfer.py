"""Host <-> device copies for tests, bench.py and smoke() that never hand PAGEABLE host memory to the HIP runtime.

Why: torch's `.to("cuda")` / `.cpu()` on ordinary numpy / torch CPU memory become hipMemcpy calls on pageable memory; the runtime pins
the caller's pages on the fly and keeps that pinning in a cache of its own after the call (profiles/r02_v_pageable_path_log.txt).  In a
long test session that frees and reuses host memory all the time this ended, three times over three rounds, in
"Memory access fault by GPU node-N on address 0x61d5fa090000" -- a HOST heap address -- with the main thread inside exactly such a copy
(gpurun_out of round 3; DESIGN.md "the abort").  The library itself has not used that path since round 2; these helpers take the
test harness off it as well: every copy goes through pinned memory torch allocates (and recycles) itself."""
from __future__ import annotations

import numpy as np


def to_device(a, device="cuda:0"):
    """numpy array (any layout) -> device tensor, staged through pinned memory."""
    import torch
    t = torch.from_numpy(np.ascontiguousarray(a))
    p = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    p.copy_(t)                                        # CPU memcpy into pinned memory
    d = p.to(device, non_blocking=True)
    torch.cuda.current_stream(d.device).synchronize()  # the pinned block may be recycled once the copy has finished
    return d


def to_host(t) -> np.ndarray:
    """device tensor -> ordinary numpy array, staged through pinned memory."""
    import torch
    if not t.is_cuda:
        return t.numpy().copy()
    t = t.contiguous()
    p = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    p.copy_(t, non_blocking=True)
    torch.cuda.current_stream(t.device).synchronize()
    return p.numpy().copy()
