"""Generates tests/golden/photo_luma_1919x1079.npz: a real-photograph input vector.

Source: the one image the reference ships, the input of its image benches (singlecolor.cpp / clahe1frame.cpp read it with
cv::imread and equalize the Y plane of its BGR2YUV conversion).  Stored here, derived: the luma plane (1919 x 1079 -- odd in
both dimensions, so CLAHE takes its REFLECT_101 padding path), a 384 x 256 BGR crop for the colour pipeline, and CRC32s of
the oracle's outputs (drift guard only -- PARITY UNPINNED, no OpenCV in this image).
Needs /root/reference (authoring container only); the tests read the .npz.   python tests/golden/make_photo_fixture.py
"""
import sys
import zlib
from pathlib import Path

import numpy as np
from PIL import Image

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "opencv-opencl_amd" / "python"))
import oracle  # noqa: E402

rgb = np.asarray(Image.open("/root/reference/hun.png").convert("RGB"))
bgr = np.ascontiguousarray(rgb[..., ::-1])
y = np.ascontiguousarray(oracle.bgr2yuv(bgr)[..., 0])
crop = np.ascontiguousarray(bgr[400:656, 700:1084])
crc = lambda a: np.uint32(zlib.crc32(np.ascontiguousarray(a).tobytes()))
out = {
    "y": y, "bgr_crop": crop,
    "crc_equalize": crc(oracle.equalize_hist(y)),
    "crc_clahe_2_8x8": crc(oracle.clahe(y, 2.0, 8, 8)),
    "crc_clahe_3_4x4": crc(oracle.clahe(y, 3.0, 4, 4)),
    "crc_clahe_2_8x8_fp_contract": crc((oracle.set_fp_contract(True), oracle.clahe(y, 2.0, 8, 8), oracle.set_fp_contract(False))[1]),
    "crc_bgr_luma_equalize_crop": crc(oracle.bgr_luma_op(crop, 0)),
    "crc_bgr_luma_clahe_crop": crc(oracle.bgr_luma_op(crop, 1, 3.0, 4, 4)),
}
np.savez_compressed(Path(__file__).parent / "photo_luma_1919x1079.npz", **out)
print(y.shape, crop.shape, {k: int(v) for k, v in out.items() if k.startswith("crc")})
