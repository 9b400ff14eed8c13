"""C++ adapter (cv::Mat-style surface + worker pool): compiles on CPU, runs its parity checks on GPU."""
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


def _build():
    subprocess.run(["make", "-C", str(ROOT / "opencv-opencl_amd" / "csrc")], check=True, capture_output=True)
    subprocess.run(["make", "-C", str(ROOT / "oracle")], check=True, capture_output=True)
    subprocess.run(["make", "-C", str(ROOT / "tests" / "cxx")], check=True, capture_output=True)
    subprocess.run(["make", "-C", str(ROOT / "opencv-opencl_amd" / "cxx")], check=True, capture_output=True)


def test_adapter_compiles_and_fails_loudly_without_gpu():
    _build()
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by test_adapter_parity_gpu")
    r = subprocess.run([str(ROOT / "opencv-opencl_amd" / "lib" / "nv12_stream"), "--frames", "2"], capture_output=True, text=True)
    assert r.returncode != 0 and "no CPU fallback" in r.stderr


@pytest.mark.gpu
def test_adapter_parity_gpu():
    _build()
    r = subprocess.run([str(ROOT / "tests" / "cxx" / "test_adapter")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all checks passed" in r.stdout


@pytest.mark.gpu
def test_stream_demo_gpu():
    _build()
    r = subprocess.run([str(ROOT / "opencv-opencl_amd" / "lib" / "nv12_stream"), "--frames", "64", "--workers", "2",
                        "--width", "1280", "--height", "720", "--op", "clahe", "--uv", "copy"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "done: 64 frames" in r.stdout and "errors=0" in r.stdout
