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


def _stream(env_extra, *args):
    """cxx/examples/nv12_stream.cpp, the shipped source, built over the stand-in ABI (under TSan when the runtime is there)."""
    import os
    target = "nv12_stream_stub_tsan"
    b = subprocess.run(["make", "-C", str(CXX), target], capture_output=True, text=True)
    if b.returncode != 0:
        target = "nv12_stream_stub"
        b = subprocess.run(["make", "-C", str(CXX), target], capture_output=True, text=True)
    assert b.returncode == 0, b.stdout + b.stderr
    env = dict(os.environ, MI_STUB_TRACE="1", **env_extra)
    r = subprocess.run([str(CXX / target), "--width", "64", "--height", "32"] + list(args), capture_output=True, text=True, timeout=300, env=env)
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-4000:]
    return r


@pytest.mark.skipif(not shutil.which("g++"), reason="no g++")
def test_nv12_stream_start_up_with_eight_devices():
    """nv12_stream.cpp:93-104 with getDeviceCount() == 8: every one of the 32 ring slots is first-touched from a thread bound next
    to the GPU of the worker that slot feeds (slot s -> worker s mod 8 -> GPU s mod 8), BEFORE any worker exists; then each worker
    places itself; frames all arrive; the ring is unpinned slot by slot."""
    import re
    r = _stream({"MI_STUB_DEVICES": "8"}, "--workers", "8", "--frames", "300")
    assert r.returncode == 0, r.stdout + r.stderr
    binds = [int(m) for m in re.findall(r"STUBTRACE bind device=(\d+)", r.stderr)]
    assert binds[:32] == [s % 8 for s in range(32)]                  # the ring, slot by slot, from the main thread
    assert sorted(binds[32:]) == list(range(8))                       # then the eight workers, each next to its own GPU
    assert "frame ring: each slot first-touched next to the GPU of the worker it feeds" in r.stdout
    for w in range(8):
        assert f"placement: worker {w} -> GPU {w}: stub: GPU {w} -> NUMA node {w // 4}" in r.stdout
    assert "workers=8 depth=6 gpus=8 frames=300" in r.stdout and re.search(r"done: 300 frames .* errors=0", r.stdout)
    assert "not unpinned" not in r.stderr and "STUB:" not in r.stderr
    # 64 workers asked for: 16 started (two per GPU), 32 % 16 == 0 so the ring is still placed slot by slot
    r = _stream({"MI_STUB_DEVICES": "8"}, "--workers", "64", "--frames", "200")
    assert r.returncode == 0 and "workers: 64 requested, 16 started (at most 2 per GPU" in r.stdout, r.stdout
    binds = [int(m) for m in re.findall(r"STUBTRACE bind device=(\d+)", r.stderr)]
    assert binds[:32] == [(s % 16) % 8 for s in range(32)] and len(binds) == 48
    # three GPUs, three workers: 32 % 3 != 0, a slot would feed changing workers -> the ring is left unplaced, and the banner says so
    r = _stream({"MI_STUB_DEVICES": "3"}, "--workers", "3", "--frames", "90")
    assert r.returncode == 0 and "submitting thread: not bound (workers spread over several GPUs)" in r.stdout
    assert len(re.findall(r"STUBTRACE bind", r.stderr)) == 3
    # one worker: the submitting thread and the ring go next to GPU 0
    r = _stream({"MI_STUB_DEVICES": "8"}, "--workers", "1", "--frames", "40")
    assert r.returncode == 0 and "placement: submitting thread + frame ring: stub: GPU 0" in r.stdout


@pytest.mark.skipif(not shutil.which("g++"), reason="no g++")
def test_nv12_stream_with_a_failing_device_and_with_none():
    """A device that refuses every submit: its frames are reported one by one, in order, and counted (the reference's drop-and-count,
    OpenCVequalHist.cpp:183-193); the stream itself finishes.  No device at all: exit 1, no fallback."""
    import re
    r = _stream({"MI_STUB_DEVICES": "8", "MI_STUB_SUBMIT_FAILS": "3"}, "--workers", "8", "--frames", "160")
    assert r.returncode == 0 and re.search(r"done: 160 frames .* errors=20", r.stdout), r.stdout
    bad = [int(m) for m in re.findall(r"frame (\d+) error: mi_pipe_submit: MI_ERR_HIP", r.stderr)]
    assert bad == [k for k in range(160) if k % 8 == 3]
    r = _stream({"MI_STUB_DEVICES": "8", "MI_STUB_CTX_FAILS": "6"}, "--workers", "8", "--frames", "80")
    assert r.returncode == 0 and re.search(r"done: 80 frames .* errors=10", r.stdout), r.stdout
    assert "mi_ctx_create(device=6) failed" in r.stderr
    r = _stream({"MI_STUB_DEVICES": "0"}, "--workers", "2", "--frames", "10")
    assert r.returncode == 1 and "no HIP device" in r.stderr


def test_the_stub_never_reaches_the_product():
    """The stand-in ABI is test infrastructure: nothing under opencv-opencl_amd/ or include/ may mention it."""
    hits = [str(p) for base in (ROOT / "opencv-opencl_amd", ROOT / "include") for p in base.rglob("*")
            if p.is_file() and p.suffix in (".hpp", ".h", ".cpp", ".hip", ".py", ".inc", "") and p.stat().st_size < (4 << 20)
            and b"stub_mi_lumaeq" in p.read_bytes()]
    assert hits == []
