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


@pytest.mark.gpu
def test_bench_line_contract_on_gpu():
    """`python bench.py` prints ONE JSON line with the fields the driver reads, the roofline and cpu_baseline objects, and the
    truthfulness fields of round 2 (traffic source, moved-bytes fraction, backend, fallbacks)."""
    import json
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "5", "--warmup", "2", "--no-extras", "--cpu-seconds", "1.5"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["metric"] == "frames/sec, 3840x2160 NV12 Y equalizeHist" and d["unit"] == "frames/s"
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "u8" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 60 and abs(d["value"] - 64 * 5 / (d["ms_per_step"] * 5e-3)) / d["value"] < 0.01
    assert d["parity_spot_check"] is True and d["fused_fallbacks_in_run"] == 0
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and rf["kernel"] == "equalize_fused_kernel"
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and 0.3 < rf["frac"] < 1.0
    assert rf["traffic"] and "profiles/" in rf["traffic_source"] and 0.2 < rf["frac_moved_bytes"] < rf["frac"]
    assert abs(rf["achieved"] - rf["alg_bytes_per_launch"] / (rf["avg_launch_ms"] * 1e-3) / 1e9) / rf["achieved"] < 1e-3
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "frames/s" and cb["cores"] >= 1 and cb["value"] > 0 and "3840x2160" in cb["sample"]
    assert d["nv12_1080p"]["value"] > d["value"]                       # four times fewer pixels per frame
    assert d["dist_backend_used"] is None and d["world_seen_by_backend"] == 1
