"""ctypes view of libmi_lumaeq.so (include/mi_lumaeq.h) for tests and bench.py.

This is glue, not the product: the product is the C ABI + HIP kernels in ../../csrc and the C++
cv::Mat adapter in ../../cxx.  Nothing here computes pixels; there is no CPU fallback -- if the
library or a GPU is missing the calls raise.
"""
from .capi import (Context, MiError, host_register, host_unregister, lib, test_lib, lib_path, device_pci_bus_id, bind_thread_near_device, device_count, version, status_str,
                   UV_FILL128, UV_COPY, STREAM_CTX, KERNEL_NAMES, DECLARED_SYMBOLS,
                   COLOR_BGR2YUV, COLOR_YUV2BGR, COLOR_YUV2BGR_NV12, COLOR_BGR2YUV_I420, OP_EQUALIZE, OP_CLAHE, OP_CHANNELS,
                   Pipe, PIPE_UV_AUTO, PIPE_UV_HOST, PIPE_UV_DEVICE, ERR_BUSY)
from . import synth, shard, xfer

__all__ = ["Context", "MiError", "host_register", "host_unregister", "lib", "test_lib", "lib_path", "device_pci_bus_id", "bind_thread_near_device", "device_count", "version", "status_str",
           "UV_FILL128", "UV_COPY", "STREAM_CTX", "KERNEL_NAMES", "DECLARED_SYMBOLS", "COLOR_BGR2YUV", "COLOR_YUV2BGR", "COLOR_YUV2BGR_NV12", "COLOR_BGR2YUV_I420", "OP_EQUALIZE", "OP_CLAHE", "OP_CHANNELS", "Pipe", "PIPE_UV_AUTO", "PIPE_UV_HOST", "PIPE_UV_DEVICE", "ERR_BUSY",
           "synth", "shard", "xfer"]
