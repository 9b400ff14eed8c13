"""ctypes binding of include/mi_lumaeq.h.  Fails loudly when the HIP library is missing."""
from __future__ import annotations

import atexit
import ctypes as C
import os
import sys
import weakref
from pathlib import Path

import numpy as np

_ROOT = Path(__file__).resolve().parents[2]          # opencv-opencl_amd/
_LIB_PATH = _ROOT / "lib" / "libmi_lumaeq.so"
_TEST_LIB_PATH = _ROOT / "lib" / "libmi_lumaeq_test.so"      # same sources + test hooks (-DMI_TEST_HOOKS); loaded by tests only

UV_FILL128, UV_COPY = 0, 1
STREAM_CTX = C.c_void_p(-1).value      # MI_STREAM_CTX: the context's private stream; 0/None = HIP null stream
KERNEL_NAMES = ["hist_partial_kernel", "equalize_lut_kernel", "lut_apply_kernel",
                "tile_hist_kernel", "tile_lut_kernel", "clahe_interp_kernel", "equalize_fused_kernel", "color_kernel",
                "fused_finish_kernel", "analyze_diff_kernel"]
COLOR_BGR2YUV, COLOR_YUV2BGR = 82, 84
COLOR_YUV2BGR_NV12, COLOR_BGR2YUV_I420 = 93, 128
OP_EQUALIZE, OP_CLAHE, OP_CHANNELS = 0, 1, 2
PIPE_UV_AUTO, PIPE_UV_HOST, PIPE_UV_DEVICE = 0, 1, 2
ERR_BUSY = 6

# every extern "C" symbol include/mi_lumaeq.h declares (tests check the .so exports them all)
DECLARED_SYMBOLS = [
    "mi_ctx_create", "mi_ctx_destroy", "mi_ctx_device", "mi_ctx_last_hip_error", "mi_ctx_last_error_msg",
    "mi_status_str", "mi_version", "mi_device_count",
    "mi_equalize_hist_u8", "mi_clahe_u8", "mi_equalize_hist_nv12", "mi_clahe_nv12",
    "mi_equalize_hist_u8_batch_dev", "mi_clahe_u8_batch_dev",
    "mi_equalize_hist_nv12_batch_dev", "mi_clahe_nv12_batch_dev",
    "mi_hist_u8_batch_dev", "mi_equalize_lut_batch_dev", "mi_lut_apply_u8_batch_dev",
    "mi_clahe_tile_luts_batch_dev",
    "mi_ctx_set_profiling", "mi_ctx_profile_read", "mi_kernel_name",
    "mi_ctx_synchronize", "mi_ctx_set_option", "mi_ctx_get_stat",
    "mi_pipe_create", "mi_pipe_destroy", "mi_pipe_submit", "mi_pipe_wait", "mi_pipe_pending", "mi_pipe_depth",
    "mi_host_register", "mi_host_unregister", "mi_clahe_u16", "mi_clahe_u16_batch_dev",
    "mi_cvt_color_u8c3", "mi_cvt_color_u8c3_batch_dev", "mi_bgr_luma_op_u8c3", "mi_bgr_luma_op_u8c3_batch_dev",
    "mi_nv12_bgr_equalize", "mi_nv12_bgr_equalize_batch_dev", "mi_cvt_color_420_u8", "mi_cvt_color_420_u8_batch_dev",
    "mi_analyze_diff_u8", "mi_analyze_diff_u8_batch_dev",
    "mi_device_pci_bus_id", "mi_thread_bind_near_device",
]

_K = len(KERNEL_NAMES)


class _Profile(C.Structure):
    _fields_ = [("total_ms", C.c_double * _K), ("launches", C.c_uint64 * _K), ("min_ms", C.c_double * _K), ("p10_ms", C.c_double * _K),
                ("p50_ms", C.c_double * _K), ("p90_ms", C.c_double * _K), ("max_ms", C.c_double * _K)]


class _NumaBinding(C.Structure):
    _fields_ = [("node", C.c_int), ("cpus", C.c_int), ("why", C.c_char * 192)]


class _PipeConfig(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("op", C.c_int), ("uv_mode", C.c_int), ("clip_limit", C.c_double),
                ("tiles_x", C.c_int), ("tiles_y", C.c_int), ("depth", C.c_int), ("uv_policy", C.c_int)]


class MiError(RuntimeError):
    def __init__(self, status: int, what: str, detail: str = ""):
        self.status = status
        super().__init__(f"{what}: {status_str(status)}" + (f" ({detail})" if detail else ""))


_lib = None
_test_lib = None


def lib_path() -> Path:
    return Path(os.environ.get("MI_LUMAEQ_LIB", str(_LIB_PATH)))


def lib() -> C.CDLL:
    """Load libmi_lumaeq.so.  No fallback: a missing library is an error."""
    global _lib
    if _lib is None:
        _lib = _load(lib_path())
    return _lib


def test_lib() -> C.CDLL:
    """libmi_lumaeq_test.so: the product sources built with the test hooks (options "fused_fault_inject", "fused_timeout_us",
    "hip_fail_after").  Tests pass it to Context(device, lib=test_lib()); nothing else loads it."""
    global _test_lib
    if _test_lib is None:
        _test_lib = _load(Path(os.environ.get("MI_LUMAEQ_TEST_LIB", str(_TEST_LIB_PATH))))
    return _test_lib


def _load(p: Path) -> C.CDLL:
    if not p.exists():
        raise FileNotFoundError(
            f"{p} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C opencv-opencl_amd/csrc` (there is no CPU fallback)")
    # One HIP runtime per process: torch bundles its own libamdhip64.so.7 (same soname as /opt/rocm's).
    # If the library were loaded first it would bring in /opt/rocm's runtime and a later `import torch`
    # would start a second one (its devices then look absent to us).  Importing torch first makes the
    # dynamic loader resolve our NEEDED libamdhip64.so.7 to the copy torch already mapped.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(str(p))
    vp, sz, i, d, i64 = C.c_void_p, C.c_size_t, C.c_int, C.c_double, C.c_int64
    L.mi_ctx_create.argtypes = [i, C.POINTER(vp)]
    L.mi_ctx_destroy.argtypes = [vp]; L.mi_ctx_destroy.restype = None
    L.mi_ctx_device.argtypes = [vp]
    L.mi_ctx_last_hip_error.argtypes = [vp]
    L.mi_ctx_last_error_msg.argtypes = [vp]; L.mi_ctx_last_error_msg.restype = C.c_char_p
    L.mi_status_str.argtypes = [i]; L.mi_status_str.restype = C.c_char_p
    L.mi_version.restype = C.c_char_p
    L.mi_kernel_name.argtypes = [i]; L.mi_kernel_name.restype = C.c_char_p
    L.mi_equalize_hist_u8.argtypes = [vp, vp, sz, vp, sz, i, i]
    L.mi_clahe_u8.argtypes = [vp, vp, sz, vp, sz, i, i, d, i, i]
    L.mi_equalize_hist_nv12.argtypes = [vp, vp, vp, i, i, i]
    L.mi_clahe_nv12.argtypes = [vp, vp, vp, i, i, i, d, i, i]
    L.mi_equalize_hist_u8_batch_dev.argtypes = [vp, vp, sz, sz, vp, sz, sz, i, i, i, vp]
    L.mi_clahe_u8_batch_dev.argtypes = [vp, vp, sz, sz, vp, sz, sz, i, i, i, d, i, i, vp]
    L.mi_equalize_hist_nv12_batch_dev.argtypes = [vp, vp, vp, i, i, i, i, vp]
    L.mi_clahe_nv12_batch_dev.argtypes = [vp, vp, vp, i, i, i, i, d, i, i, vp]
    L.mi_hist_u8_batch_dev.argtypes = [vp, vp, sz, sz, i, i, i, vp, vp]
    L.mi_equalize_lut_batch_dev.argtypes = [vp, vp, i64, i, vp, vp]
    L.mi_lut_apply_u8_batch_dev.argtypes = [vp, vp, sz, sz, vp, sz, sz, i, i, i, vp, vp]
    L.mi_clahe_tile_luts_batch_dev.argtypes = [vp, vp, sz, sz, i, i, i, d, i, i, vp, vp]
    L.mi_cvt_color_u8c3.argtypes = [vp, vp, sz, vp, sz, i, i, i]
    L.mi_cvt_color_u8c3_batch_dev.argtypes = [vp, vp, sz, sz, vp, sz, sz, i, i, i, i, vp]
    L.mi_bgr_luma_op_u8c3.argtypes = [vp, vp, sz, vp, sz, i, i, i, d, i, i]
    L.mi_bgr_luma_op_u8c3_batch_dev.argtypes = [vp, vp, sz, sz, vp, sz, sz, i, i, i, i, d, i, i, vp]
    L.mi_nv12_bgr_equalize.argtypes = [vp, vp, vp, i, i]
    L.mi_cvt_color_420_u8.argtypes = [vp, vp, sz, vp, sz, i, i, i]
    L.mi_cvt_color_420_u8_batch_dev.argtypes = [vp, vp, sz, sz, vp, sz, sz, i, i, i, i, vp]
    L.mi_nv12_bgr_equalize_batch_dev.argtypes = [vp, vp, sz, vp, sz, i, i, i, vp]
    L.mi_clahe_u16.argtypes = [vp, vp, sz, vp, sz, i, i, d, i, i]
    L.mi_clahe_u16_batch_dev.argtypes = [vp, vp, sz, sz, vp, sz, sz, i, i, i, d, i, i, vp]
    L.mi_analyze_diff_u8.argtypes = [vp, vp, sz, vp, sz, vp, sz, i, i, i, vp]
    L.mi_analyze_diff_u8_batch_dev.argtypes = [vp, vp, sz, sz, vp, sz, sz, vp, sz, sz, i, i, i, i, vp, vp]
    L.mi_host_register.argtypes = [vp, sz]
    L.mi_host_unregister.argtypes = [vp]
    L.mi_ctx_synchronize.argtypes = [vp, vp]
    L.mi_ctx_set_option.argtypes = [vp, C.c_char_p, i]
    L.mi_ctx_get_stat.argtypes = [vp, C.c_char_p, C.POINTER(C.c_uint64)]
    L.mi_pipe_create.argtypes = [vp, C.POINTER(_PipeConfig), C.POINTER(vp)]
    L.mi_pipe_destroy.argtypes = [vp]; L.mi_pipe_destroy.restype = None
    L.mi_pipe_submit.argtypes = [vp, vp, vp, C.c_uint64]
    L.mi_pipe_wait.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(vp)]
    L.mi_pipe_pending.argtypes = [vp]
    L.mi_pipe_depth.argtypes = [vp]
    L.mi_ctx_set_profiling.argtypes = [vp, i]
    L.mi_ctx_profile_read.argtypes = [vp, C.POINTER(_Profile), i]
    L.mi_device_pci_bus_id.argtypes = [i, C.c_char_p, sz]
    L.mi_thread_bind_near_device.argtypes = [i, C.POINTER(_NumaBinding)]
    return L


def status_str(s: int) -> str:
    try:
        return lib().mi_status_str(int(s)).decode()
    except Exception:
        return f"status {s}"


def version() -> str:
    return lib().mi_version().decode()


def device_count() -> int:
    return int(lib().mi_device_count())


def device_pci_bus_id(device: int) -> str:
    buf = C.create_string_buffer(64)
    rc = lib().mi_device_pci_bus_id(int(device), buf, 64)
    if rc != 0:
        raise MiError(rc, "mi_device_pci_bus_id")
    return buf.value.decode()


def bind_thread_near_device(device: int) -> dict:
    """Bind the CALLING thread to the CPUs of the device's NUMA node (mi_thread_bind_near_device): call before Context(device).
    Returns {"node", "cpus", "why"}; never raises for a platform that reports no node."""
    b = _NumaBinding()
    rc = lib().mi_thread_bind_near_device(int(device), C.byref(b))
    if rc != 0:
        raise MiError(rc, "mi_thread_bind_near_device", b.why.decode(errors="replace"))
    return {"node": int(b.node), "cpus": int(b.cpus), "why": b.why.decode(errors="replace")}


def host_register(a: np.ndarray) -> None:
    """Pin a caller-owned numpy buffer (mi_host_register); keep `a` alive until host_unregister(a)."""
    rc = lib().mi_host_register(a.ctypes.data, a.nbytes)
    if rc != 0:
        raise MiError(rc, "mi_host_register")


def host_unregister(a: np.ndarray) -> None:
    rc = lib().mi_host_unregister(a.ctypes.data)
    if rc != 0:
        raise MiError(rc, "mi_host_unregister")


def _host2d(a: np.ndarray, name: str) -> np.ndarray:
    if not isinstance(a, np.ndarray) or a.dtype != np.uint8 or a.ndim != 2:
        raise MiError(2, name, "expected a 2-D uint8 ndarray (CV_8UC1)")
    if a.size and a.strides[1] != 1:
        raise MiError(1, name, "pixel stride must be 1")
    return a


def _step(a: np.ndarray) -> int:
    return int(a.strides[0]) if a.shape[0] > 1 else max(int(a.strides[0]), a.shape[1])


def _dptr(t) -> int:
    """Device pointer of a torch CUDA tensor (or a raw int)."""
    if isinstance(t, int):
        return t
    if not t.is_cuda:
        raise MiError(1, "device pointer", "tensor is not on a HIP device")
    return int(t.data_ptr())


_live_contexts: "weakref.WeakSet[Context]" = weakref.WeakSet()


def _close_live_contexts() -> None:
    """atexit: destroy every context while the HIP runtime is still up.  Destroying one from ``__del__`` during
    interpreter finalisation can run after the runtime's own teardown (hipFree on a dead runtime)."""
    for c in list(_live_contexts):
        c.close()


atexit.register(_close_live_contexts)


class Context:
    """mi_ctx wrapper.  One per (thread x device), like the reference's per-worker OpenCL objects."""

    def __init__(self, device: int = 0, lib: "C.CDLL | None" = None):
        self._L = lib if lib is not None else globals()["lib"]()
        self._h = C.c_void_p()
        rc = self._L.mi_ctx_create(int(device), C.byref(self._h))
        if rc != 0:
            raise MiError(rc, f"mi_ctx_create(device={device})")
        _live_contexts.add(self)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            h, self._h = self._h, None
            self._L.mi_ctx_destroy(h)

    def __del__(self):
        # Never call into the library while the interpreter is finalising: the atexit hook has already closed
        # every live context, and anything that slipped past it is leaked rather than freed on a dead runtime.
        if sys is None or sys.is_finalizing():
            return
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _chk(self, rc: int, what: str):
        if rc != 0:
            raise MiError(rc, what, self._L.mi_ctx_last_error_msg(self._h).decode())

    # ---- host-pointer forms (numpy = stand-in for cv::Mat memory) ----
    def equalize_hist(self, src: np.ndarray, dst: np.ndarray | None = None) -> np.ndarray:
        src = _host2d(src, "equalize_hist")
        if dst is None or dst.shape != src.shape or dst.dtype != np.uint8:
            dst = np.empty(src.shape, np.uint8)      # Mat::create semantics: reallocate only on mismatch
        _host2d(dst, "equalize_hist")
        h, w = src.shape
        self._chk(self._L.mi_equalize_hist_u8(self._h, src.ctypes.data, _step(src), dst.ctypes.data, _step(dst), w, h),
                  "mi_equalize_hist_u8")
        return dst

    def clahe(self, src: np.ndarray, clip_limit: float = 40.0, tiles_x: int = 8, tiles_y: int = 8,
              dst: np.ndarray | None = None) -> np.ndarray:
        src = _host2d(src, "clahe")
        if dst is None or dst.shape != src.shape or dst.dtype != np.uint8:
            dst = np.empty(src.shape, np.uint8)
        h, w = src.shape
        self._chk(self._L.mi_clahe_u8(self._h, src.ctypes.data, _step(src), dst.ctypes.data, _step(dst), w, h,
                                    float(clip_limit), int(tiles_x), int(tiles_y)), "mi_clahe_u8")
        return dst

    def equalize_hist_nv12(self, frame: np.ndarray, width: int, height: int, uv_mode: int = UV_FILL128,
                           out: np.ndarray | None = None) -> np.ndarray:
        n = width * height + (width * height) // 2
        frame = np.ascontiguousarray(frame, np.uint8).reshape(-1)
        if frame.size < n:
            raise MiError(1, "equalize_hist_nv12", "frame smaller than W*H*3/2")
        if out is None:
            out = np.empty(n, np.uint8)
        self._chk(self._L.mi_equalize_hist_nv12(self._h, frame.ctypes.data, out.ctypes.data, width, height, uv_mode),
                  "mi_equalize_hist_nv12")
        return out

    def clahe_nv12(self, frame: np.ndarray, width: int, height: int, uv_mode: int = UV_FILL128,
                   clip_limit: float = 2.0, tiles_x: int = 8, tiles_y: int = 8,
                   out: np.ndarray | None = None) -> np.ndarray:
        n = width * height + (width * height) // 2
        frame = np.ascontiguousarray(frame, np.uint8).reshape(-1)
        if frame.size < n:
            raise MiError(1, "clahe_nv12", "frame smaller than W*H*3/2")
        if out is None:
            out = np.empty(n, np.uint8)
        self._chk(self._L.mi_clahe_nv12(self._h, frame.ctypes.data, out.ctypes.data, width, height, uv_mode,
                                      float(clip_limit), int(tiles_x), int(tiles_y)), "mi_clahe_nv12")
        return out

    # ---- device-resident batched forms (torch tensors only carry the memory) ----
    def equalize_hist_batch_dev(self, src, dst, width, height, n_frames, src_step=None, src_frame=None,
                                dst_step=None, dst_frame=None, stream=0):
        ss = width if src_step is None else src_step
        ds = width if dst_step is None else dst_step
        sf = ss * height if src_frame is None else src_frame
        df = ds * height if dst_frame is None else dst_frame
        self._chk(self._L.mi_equalize_hist_u8_batch_dev(self._h, _dptr(src), ss, sf, _dptr(dst), ds, df,
                                                      width, height, n_frames, stream), "mi_equalize_hist_u8_batch_dev")

    def clahe_batch_dev(self, src, dst, width, height, n_frames, clip_limit, tiles_x, tiles_y,
                        src_step=None, src_frame=None, dst_step=None, dst_frame=None, stream=0):
        ss = width if src_step is None else src_step
        ds = width if dst_step is None else dst_step
        sf = ss * height if src_frame is None else src_frame
        df = ds * height if dst_frame is None else dst_frame
        self._chk(self._L.mi_clahe_u8_batch_dev(self._h, _dptr(src), ss, sf, _dptr(dst), ds, df, width, height,
                                              n_frames, float(clip_limit), tiles_x, tiles_y, stream),
                  "mi_clahe_u8_batch_dev")

    def equalize_hist_nv12_batch_dev(self, d_in, d_out, width, height, n_frames, uv_mode=UV_FILL128, stream=0):
        self._chk(self._L.mi_equalize_hist_nv12_batch_dev(self._h, _dptr(d_in), _dptr(d_out), width, height,
                                                        n_frames, uv_mode, stream), "mi_equalize_hist_nv12_batch_dev")

    def clahe_nv12_batch_dev(self, d_in, d_out, width, height, n_frames, uv_mode=UV_FILL128,
                             clip_limit=2.0, tiles_x=8, tiles_y=8, stream=0):
        self._chk(self._L.mi_clahe_nv12_batch_dev(self._h, _dptr(d_in), _dptr(d_out), width, height, n_frames,
                                                uv_mode, float(clip_limit), tiles_x, tiles_y, stream),
                  "mi_clahe_nv12_batch_dev")

    # ---- the reference's own check: cv::absdiff + xf::cv::analyzeDiff (1frameMeasure.cpp:91-100) ----
    def analyze_diff(self, a: np.ndarray, b: np.ndarray | None = None, threshold: int = 1, want_diff: bool = False):
        """Host planes.  Returns {"above", "max_diff", "min_diff", "total", "err_per"} (+ "diff" = |a - b| when asked);
        b = None: `a` already is a difference image."""
        a = _host2d(a, "analyze_diff")
        h, w = a.shape
        if b is not None:
            b = _host2d(b, "analyze_diff")
            if b.shape != a.shape:
                raise ValueError("analyze_diff: planes differ in size")
        diff = np.empty((h, w), np.uint8) if want_diff else None
        out = (C.c_uint32 * 4)()
        self._chk(self._L.mi_analyze_diff_u8(self._h, a.ctypes.data, _step(a), b.ctypes.data if b is not None else None,
                                           _step(b) if b is not None else 0, diff.ctypes.data if want_diff else None, w, w, h,
                                           int(threshold), C.cast(out, C.c_void_p)), "mi_analyze_diff_u8")
        r = {"above": int(out[0]), "max_diff": int(out[1]), "min_diff": int(out[2]), "total": int(out[3]),
             "err_per": 100.0 * out[0] / out[3] if out[3] else 0.0}
        if want_diff:
            r["diff"] = diff
        return r

    def analyze_diff_batch_dev(self, a, b, width, height, n_frames, d_stats, threshold=1, diff=None, a_step=None, a_frame=None,
                               b_step=None, b_frame=None, diff_step=None, diff_frame=None, stream=0):
        """Device planes; d_stats = n_frames x 4 uint32 (above, max, min, total) in device memory."""
        as_ = width if a_step is None else a_step
        bs_ = width if b_step is None else b_step
        ds_ = width if diff_step is None else diff_step
        af = as_ * height if a_frame is None else a_frame
        bf = bs_ * height if b_frame is None else b_frame
        df = ds_ * height if diff_frame is None else diff_frame
        self._chk(self._L.mi_analyze_diff_u8_batch_dev(self._h, _dptr(a), as_, af, _dptr(b) if b is not None else None, bs_, bf,
                                                     _dptr(diff) if diff is not None else None, ds_, df, width, height, n_frames,
                                                     int(threshold), _dptr(d_stats), stream), "mi_analyze_diff_u8_batch_dev")

    # ---- stages ----
    def hist_batch_dev(self, src, width, height, n_frames, d_hist, src_step=None, src_frame=None, stream=0):
        ss = width if src_step is None else src_step
        sf = ss * height if src_frame is None else src_frame
        self._chk(self._L.mi_hist_u8_batch_dev(self._h, _dptr(src), ss, sf, width, height, n_frames,
                                             _dptr(d_hist), stream), "mi_hist_u8_batch_dev")

    def equalize_lut_batch_dev(self, d_hist, total, n_frames, d_lut, stream=0):
        self._chk(self._L.mi_equalize_lut_batch_dev(self._h, _dptr(d_hist), int(total), n_frames, _dptr(d_lut), stream),
                  "mi_equalize_lut_batch_dev")

    def lut_apply_batch_dev(self, src, dst, width, height, n_frames, d_lut, src_step=None, src_frame=None,
                            dst_step=None, dst_frame=None, stream=0):
        ss = width if src_step is None else src_step
        ds = width if dst_step is None else dst_step
        sf = ss * height if src_frame is None else src_frame
        df = ds * height if dst_frame is None else dst_frame
        self._chk(self._L.mi_lut_apply_u8_batch_dev(self._h, _dptr(src), ss, sf, _dptr(dst), ds, df, width, height,
                                                  n_frames, _dptr(d_lut), stream), "mi_lut_apply_u8_batch_dev")

    def clahe_tile_luts_batch_dev(self, src, width, height, n_frames, clip_limit, tiles_x, tiles_y, d_luts,
                                  src_step=None, src_frame=None, stream=0):
        ss = width if src_step is None else src_step
        sf = ss * height if src_frame is None else src_frame
        self._chk(self._L.mi_clahe_tile_luts_batch_dev(self._h, _dptr(src), ss, sf, width, height, n_frames,
                                                     float(clip_limit), tiles_x, tiles_y, _dptr(d_luts), stream),
                  "mi_clahe_tile_luts_batch_dev")

    # ---- 16-bit CLAHE (N4) ----
    def clahe16(self, src: np.ndarray, clip_limit: float = 40.0, tiles_x: int = 8, tiles_y: int = 8) -> np.ndarray:
        if not isinstance(src, np.ndarray) or src.dtype != np.uint16 or src.ndim != 2:
            raise MiError(2, "clahe16", "expected a 2-D uint16 ndarray (CV_16UC1)")
        if src.size and src.strides[1] != 2:
            raise MiError(1, "clahe16", "pixel stride must be 2")
        dst = np.empty(src.shape, np.uint16)
        h, w = src.shape
        sstep = int(src.strides[0]) if h > 1 else max(int(src.strides[0]), w * 2)
        self._chk(self._L.mi_clahe_u16(self._h, src.ctypes.data, sstep, dst.ctypes.data, w * 2, w, h, float(clip_limit),
                                     int(tiles_x), int(tiles_y)), "mi_clahe_u16")
        return dst

    def clahe16_batch_dev(self, src, dst, width, height, n_frames, clip_limit, tiles_x, tiles_y, stream=0):
        self._chk(self._L.mi_clahe_u16_batch_dev(self._h, _dptr(src), width * 2, width * 2 * height, _dptr(dst), width * 2,
                                               width * 2 * height, width, height, n_frames, float(clip_limit), tiles_x, tiles_y, stream),
                  "mi_clahe_u16_batch_dev")

    # ---- colour-domain neighbours (N3) ----
    @staticmethod
    def _host3(a, name):
        if not isinstance(a, np.ndarray) or a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
            raise MiError(2, name, "expected an HxWx3 uint8 ndarray (CV_8UC3)")
        if a.size and (a.strides[2] != 1 or a.strides[1] != 3):
            raise MiError(1, name, "pixels must be interleaved")
        return a

    def cvt_color(self, src: np.ndarray, code: int, dst: np.ndarray | None = None) -> np.ndarray:
        src = self._host3(src, "cvt_color")
        if dst is None or dst.shape != src.shape:
            dst = np.empty(src.shape, np.uint8)
        h, w = src.shape[:2]
        self._chk(self._L.mi_cvt_color_u8c3(self._h, src.ctypes.data, int(src.strides[0]) if h > 1 else w * 3, dst.ctypes.data,
                                          int(dst.strides[0]) if h > 1 else w * 3, w, h, int(code)), "mi_cvt_color_u8c3")
        return dst

    def bgr_luma_op(self, src: np.ndarray, op: int = OP_EQUALIZE, clip_limit: float = 3.0, tiles_x: int = 4, tiles_y: int = 4,
                    dst: np.ndarray | None = None) -> np.ndarray:
        src = self._host3(src, "bgr_luma_op")
        if dst is None or dst.shape != src.shape:
            dst = np.empty(src.shape, np.uint8)
        h, w = src.shape[:2]
        self._chk(self._L.mi_bgr_luma_op_u8c3(self._h, src.ctypes.data, int(src.strides[0]) if h > 1 else w * 3, dst.ctypes.data,
                                            int(dst.strides[0]) if h > 1 else w * 3, w, h, int(op), float(clip_limit),
                                            int(tiles_x), int(tiles_y)), "mi_bgr_luma_op_u8c3")
        return dst

    def cvt_color_batch_dev(self, src, dst, width, height, n_frames, code, stream=0):
        self._chk(self._L.mi_cvt_color_u8c3_batch_dev(self._h, _dptr(src), width * 3, width * 3 * height, _dptr(dst), width * 3,
                                                    width * 3 * height, width, height, n_frames, int(code), stream),
                  "mi_cvt_color_u8c3_batch_dev")

    def bgr_luma_op_batch_dev(self, src, dst, width, height, n_frames, op=OP_EQUALIZE, clip_limit=3.0, tiles_x=4, tiles_y=4, stream=0):
        self._chk(self._L.mi_bgr_luma_op_u8c3_batch_dev(self._h, _dptr(src), width * 3, width * 3 * height, _dptr(dst), width * 3,
                                                      width * 3 * height, width, height, n_frames, int(op), float(clip_limit),
                                                      int(tiles_x), int(tiles_y), stream), "mi_bgr_luma_op_u8c3_batch_dev")

    def cvt_color_420(self, src: np.ndarray, code: int, dst: np.ndarray | None = None) -> np.ndarray:
        """cv::cvtColor with COLOR_BGR2YUV_I420 (HxWx3 -> (H*3/2)xW) or COLOR_YUV2BGR_NV12 ((H*3/2)xW -> HxWx3)."""
        if not isinstance(src, np.ndarray) or src.dtype != np.uint8:
            raise MiError(2, "cvt_color_420", "expected a uint8 ndarray")
        if code == COLOR_BGR2YUV_I420:
            src = self._host3(src, "cvt_color_420")
            h, w = src.shape[:2]
            shape = (h * 3 // 2, w)
        else:
            src = _host2d(src, "cvt_color_420")
            if src.shape[0] % 3:
                raise MiError(1, "cvt_color_420", "NV12 matrix must have H*3/2 rows")
            h, w = src.shape[0] * 2 // 3, src.shape[1]
            shape = (h, w, 3)
        if dst is None or dst.shape != shape:
            dst = np.empty(shape, np.uint8)
        sstep = int(src.strides[0]) if src.shape[0] > 1 else max(int(src.strides[0]), 1)
        dstep = int(dst.strides[0]) if dst.shape[0] > 1 else max(int(dst.strides[0]), 1)
        self._chk(self._L.mi_cvt_color_420_u8(self._h, src.ctypes.data, sstep, dst.ctypes.data, dstep, w, h, int(code)), "mi_cvt_color_420_u8")
        return dst

    def cvt_color_420_batch_dev(self, src, dst, width, height, n_frames, code, stream=0):
        c3, pl = width * 3, width
        enc = code == COLOR_BGR2YUV_I420
        self._chk(self._L.mi_cvt_color_420_u8_batch_dev(self._h, _dptr(src), c3 if enc else pl, (c3 * height) if enc else pl * height * 3 // 2,
                                                      _dptr(dst), pl if enc else c3, (pl * height * 3 // 2) if enc else c3 * height,
                                                      width, height, n_frames, int(code), stream), "mi_cvt_color_420_u8_batch_dev")

    def nv12_bgr_equalize(self, nv12: np.ndarray, width: int, height: int, out: np.ndarray | None = None) -> np.ndarray:
        """NV12 -> BGR -> equalizeHist on B, G, R -> NV12 (BASELINE.json config 5 read literally)."""
        if not isinstance(nv12, np.ndarray) or nv12.dtype != np.uint8 or not nv12.flags.c_contiguous:
            raise MiError(2, "nv12_bgr_equalize", "expected a contiguous uint8 ndarray")
        if width >= 0 and height >= 0 and nv12.size != width * height * 3 // 2:
            raise MiError(1, "nv12_bgr_equalize", "NV12 frame must hold width*height*3/2 bytes")
        if out is None:
            out = np.empty_like(nv12)
        self._chk(self._L.mi_nv12_bgr_equalize(self._h, nv12.ctypes.data, out.ctypes.data, int(width), int(height)), "mi_nv12_bgr_equalize")
        return out

    def nv12_bgr_equalize_batch_dev(self, src, dst, width, height, n_frames, stream=0, frame_stride=None):
        fs = width * height * 3 // 2 if frame_stride is None else int(frame_stride)
        self._chk(self._L.mi_nv12_bgr_equalize_batch_dev(self._h, _dptr(src), fs, _dptr(dst), fs, width, height, n_frames, stream),
                  "mi_nv12_bgr_equalize_batch_dev")

    def synchronize(self, stream=0):
        """Wait for `stream`; raises only if the fused path met a frame it refused to repair (see get_stat)."""
        self._chk(self._L.mi_ctx_synchronize(self._h, stream), "mi_ctx_synchronize")

    def get_stat(self, name: str) -> int:
        """Sticky counters of the fused path's fail-soft machinery (mi_ctx_get_stat); synchronise the work's stream first."""
        v = C.c_uint64(0)
        self._chk(self._L.mi_ctx_get_stat(self._h, name.encode(), C.byref(v)), "mi_ctx_get_stat")
        return int(v.value)

    def set_option(self, name: str, value: int):
        self._chk(self._L.mi_ctx_set_option(self._h, name.encode(), int(value)), "mi_ctx_set_option")

    # ---- profiling ----
    def set_profiling(self, on):
        """False / 0: off.  True / 1: HIP events around every kernel.  2: every kernel but the housekeeping launch behind a fused kernel."""
        self._chk(self._L.mi_ctx_set_profiling(self._h, int(on)), "mi_ctx_set_profiling")

    def profile_read(self, reset: bool = True) -> dict:
        p = _Profile()
        self._chk(self._L.mi_ctx_profile_read(self._h, C.byref(p), 1 if reset else 0), "mi_ctx_profile_read")
        return {KERNEL_NAMES[k]: {"total_ms": p.total_ms[k], "launches": int(p.launches[k]), "min_ms": p.min_ms[k], "p10_ms": p.p10_ms[k],
                                  "p50_ms": p.p50_ms[k], "p90_ms": p.p90_ms[k], "max_ms": p.max_ms[k]} for k in range(_K)}


class Pipe:
    """mi_pipe wrapper: asynchronous in-order NV12 frame pipeline on one context (frames are numpy uint8 arrays of W*H*3/2 bytes).
    The arrays handed to submit() are kept alive until wait() returns them."""

    def __init__(self, ctx: Context, width: int, height: int, op: int = OP_EQUALIZE, uv_mode: int = UV_FILL128,
                 clip_limit: float = 2.0, tiles_x: int = 8, tiles_y: int = 8, depth: int = 0, uv_policy: int = PIPE_UV_AUTO):
        self._ctx = ctx
        self._h = C.c_void_p()
        self._held = {}
        cfg = _PipeConfig(int(width), int(height), int(op), int(uv_mode), float(clip_limit), int(tiles_x), int(tiles_y), int(depth), int(uv_policy))
        ctx._chk(self._ctx._L.mi_pipe_create(ctx._h, C.byref(cfg), C.byref(self._h)), "mi_pipe_create")
        self.frame_bytes = width * height * 3 // 2

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            h, self._h = self._h, None
            self._ctx._L.mi_pipe_destroy(h)
            self._held.clear()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def pending(self) -> int:
        return int(self._ctx._L.mi_pipe_pending(self._h))

    @property
    def depth(self) -> int:
        return int(self._ctx._L.mi_pipe_depth(self._h))

    def submit(self, frame_in: np.ndarray, frame_out: np.ndarray, tag: int) -> bool:
        """False when the pipe is full (MI_ERR_BUSY: call wait() first)."""
        for a in (frame_in, frame_out):
            if not isinstance(a, np.ndarray) or a.dtype != np.uint8 or not a.flags.c_contiguous or a.size < self.frame_bytes:
                raise MiError(1, "mi_pipe_submit", "frames must be contiguous uint8 arrays of W*H*3/2 bytes")
        rc = self._ctx._L.mi_pipe_submit(self._h, frame_in.ctypes.data, frame_out.ctypes.data, int(tag))
        if rc == ERR_BUSY:
            return False
        self._ctx._chk(rc, "mi_pipe_submit")
        self._held[int(tag)] = (frame_in, frame_out)
        return True

    def wait(self):
        """Blocks for the oldest pending frame; returns (tag, output array)."""
        tag, ptr = C.c_uint64(0), C.c_void_p()
        rc = self._ctx._L.mi_pipe_wait(self._h, C.byref(tag), C.byref(ptr))
        held = self._held.pop(int(tag.value), (None, None))
        self._ctx._chk(rc, "mi_pipe_wait")
        return int(tag.value), held[1]
