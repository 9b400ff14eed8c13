"""Generates tests/golden/golden_small.npz: seeded synthetic inputs (mi_lumaeq.synth) and the
outputs of the CPU oracle (oracle/lumaeq_oracle.c).  PARITY UNPINNED: no OpenCV exists in this
image, so these vectors pin the restatement against drift, not against cv::equalizeHist itself.
Run from the repo root:  python tests/golden/make_golden.py
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "opencv-opencl_amd" / "python"))
import oracle  # noqa: E402
from mi_lumaeq import synth  # noqa: E402

out = {}
for (w, h) in [(64, 48), (63, 47), (16, 15)]:
    for dist in synth.DISTS:
        src = synth.y_plane(w, h, dist, 100)
        n = f"eq_{w}x{h}_{dist}"
        out[n + "__src"] = src
        out[n + "__dst"] = oracle.equalize_hist(src)
        for (clip, tx, ty) in [(2.0, 8, 8), (3.0, 4, 4)]:
            n = f"clahe_{w}x{h}_{dist}_c{clip}_t{tx}x{ty}"
            out[n + "__src"] = src
            out[n + "__cfg"] = np.array([clip, tx, ty], np.float64)
            out[n + "__dst"] = oracle.clahe(src, clip, tx, ty)
np.savez_compressed(Path(__file__).parent / "golden_small.npz", **out)
print("wrote", len(out), "arrays")
