"""CPU checks of bench.py's contract: it refuses to run without a GPU (no fallback), and the cpu_baseline leg
returns the fields the driver reads."""
import argparse
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def test_bench_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "no CPU fallback" in (r.stderr + r.stdout)
    assert not r.stdout.strip().startswith("{")                  # no number is reported


def test_cpu_baseline_fields():
    import bench
    args = argparse.Namespace(dist="D2", uv="fill128", op="equalize", cpu_seconds=1.0)
    cb = bench.cpu_baseline(args, 640, 360)
    assert set(["value", "unit", "cores", "kind", "sample"]) <= set(cb)
    assert cb["unit"] == "frames/s" and cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0
    assert "640x360" in cb["sample"]
