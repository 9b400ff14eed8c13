"""micv::FramePool (cxx/mi_pool.hpp) with 2, 3 and 8 devices -- on the CPU, against a test-only stand-in for the C ABI
(tests/cxx/stub_mi_lumaeq.hpp: randomised completion delays, one device whose mi_pipe_submit fails, one whose mi_ctx_create fails,
one whose mi_pipe_wait fails now and then).  Every box this project can reach has ONE GPU, so these are the only runs of the pool's
multi-device branches: worker w -> GPU w mod N, at most two workers per GPU, placement before context creation on the worker's own
thread, strictly in-order delivery under out-of-order completion, per-frame drop-and-count (the reference's
OpenCVequalHist.cpp:183-193), finish() with a dead device.  Plain and under ThreadSanitizer (tests/cxx/test_pool_multidev.cpp)."""
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
CXX = ROOT / "tests" / "cxx"


def _run(target, timeout):
    r = subprocess.run(["make", "-C", str(CXX), target], capture_output=True, text=True)
    return r, (subprocess.run([str(CXX / target)], capture_output=True, text=True, timeout=timeout) if r.returncode == 0 else None)


@pytest.mark.skipif(not shutil.which("g++"), reason="no g++")
def test_frame_pool_on_eight_stubbed_devices():
    b, r = _run("test_pool_multidev", 300)
    assert b.returncode == 0, b.stdout + b.stderr
    assert r.returncode == 0 and "pool multi-device ok" in r.stdout, r.stdout + r.stderr
    assert "STUB:" not in r.stderr, r.stderr            # no pipe closed with frames pending, no context closed before its pipe
    assert "out-of-order completions at the devices" in r.stdout     # the re-sequencer was given work


@pytest.mark.skipif(not shutil.which("g++"), reason="no g++")
def test_frame_pool_on_eight_stubbed_devices_under_thread_sanitizer():
    b, r = _run("test_pool_multidev_tsan", 900)
    if b.returncode != 0:
        pytest.skip("no ThreadSanitizer runtime for g++ here: " + b.stderr[-200:])
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-4000:]
    assert r.returncode == 0 and "pool multi-device ok" in r.stdout and "STUB:" not in r.stderr, r.stdout + r.stderr


def test_the_stub_never_reaches_the_product():
    """The stand-in ABI is test infrastructure: nothing under opencv-opencl_amd/ or include/ may mention it."""
    hits = [str(p) for base in (ROOT / "opencv-opencl_amd", ROOT / "include") for p in base.rglob("*")
            if p.is_file() and p.suffix in (".hpp", ".h", ".cpp", ".hip", ".py", ".inc", "") and p.stat().st_size < (4 << 20)
            and b"stub_mi_lumaeq" in p.read_bytes()]
    assert hits == []
