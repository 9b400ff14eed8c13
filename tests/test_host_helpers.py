"""CPU unit tests of the library's stand-alone host helpers (tests/cxx/test_host_helpers.cpp):
  * drain_guard.hpp -- every exit of a host form that follows an enqueued copy on caller memory synchronises its stream(s) first
    (the error handling the reference's accelerator path lacks: OpenCLequalHist.cpp:346-367); exercised with a stubbed
    synchronise call and stubbed failures, including an exception;
  * copy_crew.hpp -- the calling thread + one helper packing planes through pinned staging; bytes compared with memcpy, plain and
    under ThreadSanitizer."""
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
CXX = ROOT / "tests" / "cxx"


def _make(target):
    r = subprocess.run(["make", "-C", str(CXX), target], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.skipif(not shutil.which("g++"), reason="no g++")
def test_drain_guard_and_copy_crew():
    _make("test_host_helpers")
    r = subprocess.run([str(CXX / "test_host_helpers")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "host helpers ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.skipif(not shutil.which("g++"), reason="no g++")
def test_copy_crew_under_thread_sanitizer():
    r = subprocess.run(["make", "-C", str(CXX), "test_host_helpers_tsan"], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("no ThreadSanitizer runtime for g++ here: " + r.stderr[-200:])
    r = subprocess.run([str(CXX / "test_host_helpers_tsan")], capture_output=True, text=True, timeout=600)
    assert "ThreadSanitizer" not in r.stderr, r.stderr
    assert r.returncode == 0 and "host helpers ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.skipif(not shutil.which("g++"), reason="no g++")
def test_host_helpers_under_address_and_ub_sanitizers():
    r = subprocess.run(["make", "-C", str(CXX), "test_host_helpers_asan"], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("no AddressSanitizer runtime for g++ here: " + r.stderr[-200:])
    r = subprocess.run([str(CXX / "test_host_helpers_asan")], capture_output=True, text=True, timeout=600,
                       env={"ASAN_OPTIONS": "detect_leaks=1", "PATH": "/usr/bin:/bin"})
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
    assert r.returncode == 0 and "host helpers ok" in r.stdout, r.stdout + r.stderr
