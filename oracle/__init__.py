"""CPU oracle for the luma-equalization hot path -- TEST INFRASTRUCTURE, never the product.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package.  The shipped path (opencv-opencl_amd/) never does and has no CPU fallback.

PARITY UNPINNED: the arithmetic of the reference path lives in OpenCV 4.4 (third-party, absent
from /root/reference and from this image); see the header of ``lumaeq_oracle.c``.

Two independent restatements are provided so they can be checked against each other:
  * ``c``  -- ctypes view of ``lumaeq_oracle.c`` (fast; also the CPU baseline),
  * ``np_*`` functions in ``np_oracle.py`` -- numpy / pure-Python, small inputs only.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB_PATH = _HERE / "build" / "liblumaeq_oracle.so"
_lib = None

_u8p = ctypes.POINTER(ctypes.c_uint8)
_i32p = ctypes.POINTER(ctypes.c_int32)


def build(force: bool = False) -> Path:
    """Compile the C restatement with gcc (oracle/Makefile)."""
    newest = max((_HERE / n).stat().st_mtime for n in ("lumaeq_oracle.c", "color_oracle.c", "Makefile"))
    if force or not _LIB_PATH.exists() or _LIB_PATH.stat().st_mtime < newest:
        subprocess.run(["make", "-C", str(_HERE), "-s"] + (["-B"] if force else []), check=True)
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(str(_LIB_PATH))
        L.orc_set_threads.argtypes = [ctypes.c_int]
        L.orc_set_threads.restype = ctypes.c_int
        L.orc_hist_u8.argtypes = [_u8p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, _i32p]
        L.orc_equalize_lut.argtypes = [_i32p, ctypes.c_int64, _u8p, ctypes.POINTER(ctypes.c_int)]
        L.orc_lut_apply_u8.argtypes = [_u8p, ctypes.c_size_t, _u8p, ctypes.c_size_t,
                                       ctypes.c_int, ctypes.c_int, _u8p]
        L.orc_equalize_hist_u8.argtypes = [_u8p, ctypes.c_size_t, _u8p, ctypes.c_size_t,
                                           ctypes.c_int, ctypes.c_int]
        L.orc_clahe_tile_luts.argtypes = [_u8p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_double, ctypes.c_int, ctypes.c_int, _u8p]
        L.orc_clahe_u8.argtypes = [_u8p, ctypes.c_size_t, _u8p, ctypes.c_size_t, ctypes.c_int,
                                   ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_int]
        L.orc_nv12_frame.argtypes = [_u8p, _u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_int]
        L.orc_bgr2yuv_u8.argtypes = [_u8p, ctypes.c_size_t, _u8p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int]
        L.orc_yuv2bgr_u8.argtypes = [_u8p, ctypes.c_size_t, _u8p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int]
        L.orc_bgr_luma_op.argtypes = [_u8p, _u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_int]
        L.orc_clahe_u16.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int,
                                    ctypes.c_double, ctypes.c_int, ctypes.c_int]
        L.orc_clahe_u16.restype = ctypes.c_int
        L.orc_nv12_to_bgr.argtypes = [_u8p, _u8p, ctypes.c_int, ctypes.c_int]
        L.orc_bgr_to_nv12.argtypes = [_u8p, _u8p, ctypes.c_int, ctypes.c_int]
        L.orc_nv12_bgr_equalize.argtypes = [_u8p, _u8p, ctypes.c_int, ctypes.c_int]
        L.orc_bgr_to_i420.argtypes = [_u8p, _u8p, ctypes.c_int, ctypes.c_int]
        L.orc_bgr_to_i420.restype = ctypes.c_int
        for f in (L.orc_bgr2yuv_u8, L.orc_yuv2bgr_u8, L.orc_bgr_luma_op, L.orc_nv12_to_bgr, L.orc_bgr_to_nv12, L.orc_nv12_bgr_equalize):
            f.restype = ctypes.c_int
        for f in (L.orc_hist_u8, L.orc_equalize_lut, L.orc_lut_apply_u8, L.orc_equalize_hist_u8,
                  L.orc_clahe_tile_luts, L.orc_clahe_u8, L.orc_nv12_frame):
            f.restype = ctypes.c_int
        _lib = L
    return _lib


def set_threads(n: int) -> int:
    """Set the OpenMP thread count of the C restatement; returns the count in effect."""
    return lib().orc_set_threads(int(n))


def _check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"oracle {what} failed with status {rc}")


def _as2d(a: np.ndarray) -> np.ndarray:
    a = np.asarray(a)
    if a.dtype != np.uint8 or a.ndim != 2:
        raise TypeError("oracle: expected a 2-D uint8 array (CV_8UC1)")
    if a.size and a.strides[1] != 1:
        raise ValueError("oracle: pixel stride must be 1")
    return a


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(_u8p)


def _step(a: np.ndarray) -> int:
    return int(a.strides[0]) if a.shape[0] > 1 else max(int(a.strides[0]), a.shape[1])


def hist(src: np.ndarray) -> np.ndarray:
    src = _as2d(src)
    h = np.zeros(256, np.int32)
    _check(lib().orc_hist_u8(_ptr(src), _step(src), src.shape[1], src.shape[0],
                             h.ctypes.data_as(_i32p)), "hist")
    return h


def equalize_lut(h: np.ndarray, total: int):
    h = np.ascontiguousarray(h, np.int32)
    lut = np.zeros(256, np.uint8)
    first = ctypes.c_int(-1)
    _check(lib().orc_equalize_lut(h.ctypes.data_as(_i32p), int(total), _ptr(lut),
                                  ctypes.byref(first)), "equalize_lut")
    return lut, first.value


def equalize_hist(src: np.ndarray, dst: np.ndarray | None = None) -> np.ndarray:
    """cv::equalizeHist semantics on a 2-D uint8 array (rows may be strided views)."""
    src = _as2d(src)
    if dst is None:
        dst = np.empty(src.shape, np.uint8)
    dst = _as2d(dst)
    if dst.shape != src.shape:
        raise ValueError("dst shape mismatch")
    if src.size == 0:
        return dst
    _check(lib().orc_equalize_hist_u8(_ptr(src), _step(src), _ptr(dst), _step(dst),
                                      src.shape[1], src.shape[0]), "equalize_hist")
    return dst


def clahe_tile_luts(src: np.ndarray, clip_limit: float, tiles_x: int, tiles_y: int) -> np.ndarray:
    src = _as2d(src)
    luts = np.zeros((tiles_y * tiles_x, 256), np.uint8)
    _check(lib().orc_clahe_tile_luts(_ptr(src), _step(src), src.shape[1], src.shape[0],
                                     float(clip_limit), tiles_x, tiles_y, _ptr(luts)), "clahe_tile_luts")
    return luts


def clahe(src: np.ndarray, clip_limit: float = 40.0, tiles_x: int = 8, tiles_y: int = 8,
          dst: np.ndarray | None = None) -> np.ndarray:
    """cv::createCLAHE(clip_limit, Size(tiles_x, tiles_y))->apply semantics."""
    src = _as2d(src)
    if dst is None:
        dst = np.empty(src.shape, np.uint8)
    dst = _as2d(dst)
    if src.size == 0:
        return dst
    _check(lib().orc_clahe_u8(_ptr(src), _step(src), _ptr(dst), _step(dst), src.shape[1],
                              src.shape[0], float(clip_limit), tiles_x, tiles_y), "clahe")
    return dst


def nv12_frame(frame: np.ndarray, width: int, height: int, uv_mode: int = 0, op: int = 0,
               clip_limit: float = 2.0, tiles_x: int = 8, tiles_y: int = 8) -> np.ndarray:
    """Whole tightly packed NV12 frame: op on Y, UV = 128 (uv_mode 0) or copied (uv_mode 1)."""
    frame = np.ascontiguousarray(frame, np.uint8).reshape(-1)
    n = width * height + (width * height) // 2
    if frame.size < n:
        raise ValueError("NV12 frame too small")
    out = np.empty(n, np.uint8)
    _check(lib().orc_nv12_frame(_ptr(frame), _ptr(out), width, height, uv_mode, op,
                                float(clip_limit), tiles_x, tiles_y), "nv12_frame")
    return out


def clahe16(src: np.ndarray, clip_limit: float = 40.0, tiles_x: int = 8, tiles_y: int = 8) -> np.ndarray:
    """cv::createCLAHE(...)->apply on CV_16UC1."""
    src = np.asarray(src)
    if src.dtype != np.uint16 or src.ndim != 2:
        raise TypeError("oracle: expected a 2-D uint16 array (CV_16UC1)")
    if src.size and src.strides[1] != 2:
        raise ValueError("oracle: pixel stride must be 2")
    dst = np.empty(src.shape, np.uint16)
    if src.size == 0:
        return dst
    sstep = int(src.strides[0]) if src.shape[0] > 1 else max(int(src.strides[0]), src.shape[1] * 2)
    _check(lib().orc_clahe_u16(src.ctypes.data, sstep, dst.ctypes.data, dst.shape[1] * 2, src.shape[1], src.shape[0],
                               float(clip_limit), tiles_x, tiles_y), "clahe16")
    return dst


def _as3(a: np.ndarray) -> np.ndarray:
    a = np.asarray(a)
    if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
        raise TypeError("oracle: expected an HxWx3 uint8 array (CV_8UC3)")
    if a.size and (a.strides[2] != 1 or a.strides[1] != 3):
        raise ValueError("oracle: pixels must be interleaved and tightly packed within a row")
    return a


def bgr2yuv(src: np.ndarray) -> np.ndarray:
    """cv::cvtColor(src, COLOR_BGR2YUV) on CV_8UC3 (parity unpinned)."""
    src = _as3(src)
    dst = np.empty(src.shape, np.uint8)
    if src.size:
        _check(lib().orc_bgr2yuv_u8(_ptr(src), int(src.strides[0]), _ptr(dst), int(dst.strides[0]), src.shape[1], src.shape[0]), "bgr2yuv")
    return dst


def yuv2bgr(src: np.ndarray) -> np.ndarray:
    src = _as3(src)
    dst = np.empty(src.shape, np.uint8)
    if src.size:
        _check(lib().orc_yuv2bgr_u8(_ptr(src), int(src.strides[0]), _ptr(dst), int(dst.strides[0]), src.shape[1], src.shape[0]), "yuv2bgr")
    return dst


def bgr_luma_op(src: np.ndarray, op: int = 0, clip_limit: float = 3.0, tiles_x: int = 4, tiles_y: int = 4) -> np.ndarray:
    """BGR2YUV -> split -> equalizeHist (op 0) / CLAHE (op 1) on Y -> merge -> YUV2BGR (singlecolor.cpp:39-66, clahe1frame.cpp:83-102)."""
    src = np.ascontiguousarray(_as3(src))
    dst = np.empty(src.shape, np.uint8)
    if src.size:
        _check(lib().orc_bgr_luma_op(_ptr(src), _ptr(dst), src.shape[1], src.shape[0], op, float(clip_limit), tiles_x, tiles_y), "bgr_luma_op")
    return dst


def set_fp_contract(on: bool) -> bool:
    """CLAHE interpolation arithmetic: False (default) = separately rounded multiplies and adds (x86-64 baseline OpenCV);
    True = the fused multiply-adds GCC forms on FMA targets (a distribution OpenCV for aarch64, the reference's own
    platform).  Returns the previous setting.  Process-wide test switch."""
    L = lib()
    L.orc_set_fp_contract.argtypes = [ctypes.c_int]
    L.orc_set_fp_contract.restype = ctypes.c_int
    return bool(L.orc_set_fp_contract(1 if on else 0))


def _nv12(a: np.ndarray, width: int, height: int) -> np.ndarray:
    a = np.ascontiguousarray(a, np.uint8).reshape(-1)
    if width < 0 or height < 0 or width % 2 or height % 2:
        raise ValueError("oracle: NV12 4:2:0 needs even width and height")
    if a.size != width * height * 3 // 2:
        raise ValueError("oracle: NV12 frame must hold width*height*3/2 bytes")
    return a


def nv12_to_bgr(nv12: np.ndarray, width: int, height: int) -> np.ndarray:
    """cv::cvtColor(nv12, COLOR_YUV2BGR_NV12) (BASELINE config 5 read literally; parity unpinned)."""
    a = _nv12(nv12, width, height)
    dst = np.empty((height, width, 3), np.uint8)
    if dst.size:
        _check(lib().orc_nv12_to_bgr(_ptr(a), _ptr(dst), width, height), "nv12_to_bgr")
    return dst


def bgr_to_nv12(bgr: np.ndarray) -> np.ndarray:
    """cv::cvtColor(bgr, COLOR_BGR2YUV_I420) with U and V interleaved into an NV12 chroma plane."""
    bgr = np.ascontiguousarray(_as3(bgr))
    h, w = bgr.shape[:2]
    dst = np.empty(w * h * 3 // 2, np.uint8)
    if dst.size:
        _check(lib().orc_bgr_to_nv12(_ptr(bgr), _ptr(dst), w, h), "bgr_to_nv12")
    return dst


def bgr_to_i420(bgr: np.ndarray) -> np.ndarray:
    """cv::cvtColor(bgr, COLOR_BGR2YUV_I420) (1frameMeasure.cpp:32): returns the (H*3/2, W) CV_8UC1 matrix."""
    bgr = np.ascontiguousarray(_as3(bgr))
    h, w = bgr.shape[:2]
    dst = np.empty((h * 3 // 2, w), np.uint8)
    if dst.size:
        _check(lib().orc_bgr_to_i420(_ptr(bgr), _ptr(dst), w, h), "bgr_to_i420")
    return dst


def nv12_bgr_equalize(nv12: np.ndarray, width: int, height: int) -> np.ndarray:
    """NV12 -> BGR -> equalizeHist on B, G and R -> NV12 (BASELINE.json config 5 read literally)."""
    a = _nv12(nv12, width, height)
    dst = np.empty_like(a)
    if dst.size:
        _check(lib().orc_nv12_bgr_equalize(_ptr(a), _ptr(dst), width, height), "nv12_bgr_equalize")
    return dst


from .np_oracle import np_equalize_hist, np_clahe, np_clahe_geometry, np_nv12_bgr_equalize, np_analyze_diff  # noqa: E402,F401
