import faulthandler
import os
import sys
from pathlib import Path

faulthandler.enable(file=sys.__stderr__)        # a native crash (HIP runtime, our library) prints the Python stack of every thread
                                                # (sys.stderr itself is pytest's capture object under --capture=sys: no fileno)

import pytest

ROOT = Path(__file__).resolve().parents[1]

# In front of faulthandler: the native call chain of the thread that raised a fatal signal (tests/cxx/abrt_trace.c).  Round 1
# saw three silent SIGABRTs raised by a runtime thread in ~30 GPU sessions; with this loaded for the whole session a recurrence
# names the library that called abort() instead of leaving only "Fatal Python error: Aborted".  It recurred once in round 2 and
# the trace was LOST: pytest's default capture had fd 2 pointing at a temporary file.  Hence pytest.ini (--capture=sys leaves the file
# descriptors alone) and the copy the tracer appends to gpurun_out/abrt_trace.log.
_tracer = ROOT / "tests" / "cxx" / "libabrt_trace.so"
if _tracer.exists():
    try:
        import ctypes
        ctypes.CDLL(str(_tracer))
    except OSError:
        pass
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "opencv-opencl_amd" / "python"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU must fail loudly, never pass on a fallback: leave gpu tests
    # selected; they raise from mi_ctx_create.  Without -m they are skipped when no GPU exists.
    if config.getoption("-m"):
        return
    if not _have_gpu():
        skip = pytest.mark.skip(reason="no HIP device")
        for it in items:
            if "gpu" in it.keywords:
                it.add_marker(skip)


@pytest.fixture(scope="session")
def built_lib():
    """libmi_lumaeq.so, built on demand (hipcc cross-compiles without a GPU)."""
    import subprocess
    import mi_lumaeq
    if not mi_lumaeq.lib_path().exists():
        subprocess.run(["make", "-C", str(ROOT / "opencv-opencl_amd" / "csrc")], check=True)
    return mi_lumaeq.lib()


@pytest.fixture(scope="session")
def ctx(built_lib):
    import mi_lumaeq
    c = mi_lumaeq.Context(0)
    yield c
    c.close()


def pytest_sessionfinish(session, exitstatus):
    """Leave nothing of ours for interpreter shutdown: destroy every live library context and drain the device while
    the HIP runtime is certainly still up (the library also does this from an atexit hook)."""
    try:
        from mi_lumaeq import capi
        capi._close_live_contexts()
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()
    except Exception:
        pass
