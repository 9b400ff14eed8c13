"""Generates tests/golden/photo_like_crc.json: CRC32s of the synthetic photo-like scene (mi_lumaeq.synth.photo_like: piecewise-smooth
gradients, flat and saturated regions; odd-sized 1919 x 1079, so CLAHE takes its REFLECT_101 padding path) and of the oracle's
outputs on it.  A DRIFT GUARD for the generator and the oracle, nothing more -- PARITY UNPINNED (no OpenCV in this image).
The scene itself is regenerated from its seed wherever it is needed; no image file ships.   python tests/golden/make_photo_like_crc.py
"""
import json
import sys
import zlib
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "opencv-opencl_amd" / "python"))
import oracle  # noqa: E402
from mi_lumaeq import synth  # noqa: E402

SEED, W, H = 20261004, 1919, 1079
y = synth.photo_like(W, H, SEED)
crop = synth.photo_like(384, 256, SEED + 1, channels=3)
crc = lambda a: int(zlib.crc32(np.ascontiguousarray(a).tobytes()))
old = oracle.set_fp_contract(True)
c8f = oracle.clahe(y, 2.0, 8, 8)
oracle.set_fp_contract(old)
out = {
    "seed": SEED, "width": W, "height": H,
    "crc_input_y": crc(y), "crc_input_bgr_crop": crc(crop),
    "crc_equalize": crc(oracle.equalize_hist(y)),
    "crc_clahe_2_8x8": crc(oracle.clahe(y, 2.0, 8, 8)),
    "crc_clahe_3_4x4": crc(oracle.clahe(y, 3.0, 4, 4)),
    "crc_clahe_2_8x8_fp_contract": crc(c8f),
    "crc_bgr_luma_equalize_crop": crc(oracle.bgr_luma_op(crop, 0)),
    "crc_bgr_luma_clahe_crop": crc(oracle.bgr_luma_op(crop, 1, 3.0, 4, 4)),
}
(Path(__file__).parent / "photo_like_crc.json").write_text(json.dumps(out, indent=1) + "\n")
print(out)
