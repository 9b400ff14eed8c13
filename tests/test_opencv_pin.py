"""First contact with a REAL OpenCV, self-verifying (parity is unpinned until this has run somewhere).

The C++ surface INTEGRATION.md section 2 hands a maintainer -- mi_cv::equalizeHist(cv::InputArray, cv::OutputArray),
mi_cv::createCLAHE(...)->apply, the LD_PRELOAD interposer -- against the machine's own OpenCV:

  tests/cxx/test_adapter_opencv.cpp   real cv::Mat through the front end vs cv::equalizeHist / cv::createCLAHE: the five BASELINE shapes,
                                      1919 x 1079, every known answer of kat.json (through OpenCV itself, too: that pins SURVEY App. A),
                                      a ROI view, a caller-owned dst, a CV_8UC3 input, CV_16UC1 CLAHE; finds the CLAHE arithmetic mode
  tests/cxx/interpose_probe.cpp       an unmodified OpenCV program, run plain and under LD_PRELOAD=libmi_cv_interpose.so

Where `pkg-config --exists opencv4`, ONE command produces the pin record:  python -m pytest tests/test_opencv_pin.py -m gpu
-> gpurun_out/opencv_pin.json (OpenCV version, the CPU baseline / dispatch lines of getBuildInformation(), which CLAHE arithmetic
mode matched, every check with pass / fail).  Neither the authoring image nor this pool's GPU boxes have an OpenCV: there the GPU
half skips, and the CPU half compiles both programs against the declaration-only headers in tests/cxx/opencv_decl -- a syntax
check that pins nothing (no mock OpenCV is built to run them against: that would pin nothing either)."""
import json
import os
import shutil
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
CXX_TESTS = ROOT / "tests" / "cxx"
sys.path.insert(0, str(ROOT / "tests"))


def have_opencv() -> bool:
    return shutil.which("pkg-config") is not None and subprocess.run(["pkg-config", "--exists", "opencv4"]).returncode == 0


def write_kat_txt(path: Path) -> int:
    """kat.json -> one line per known answer with explicit pixels: `id op rows cols clip tiles_x tiles_y src... | dst...` (what
    test_adapter_opencv reads; C++ needs no JSON parser).  Entries that state a property rather than pixels (CL-3: the padded
    geometry) are left out; entries that give a LUT are expanded through it."""
    from test_oracle import KAT, _kat_src
    lines = ["# written by tests/test_opencv_pin.py from tests/golden/kat.json"]
    for sec in ("equalize", "clahe", "derived"):
        for k in KAT[sec]:
            if "shape" not in k or not any(s in k for s in ("src", "src_runs", "src_const", "src_arange", "src_quadrants")):
                continue
            h, w = k["shape"]
            if h * w > 1 << 16:
                continue                                             # (EQ-6, a constant 4K frame: the program's own 4K frames cover the size)
            src = _kat_src(k)
            if "dst" in k:
                dst = np.array(k["dst"], np.uint8).reshape(h, w)
            elif "dst_const" in k:
                dst = np.full((h, w), k["dst_const"], np.uint8)
            elif "dst_arange" in k:
                dst = np.arange(k["dst_arange"], dtype=np.uint8).reshape(h, w)
            elif "lut" in k:
                lut = np.zeros(256, np.uint8)
                for v, o in k["lut"].items():
                    lut[int(v)] = o
                dst = lut[src]
            else:
                continue
            op = k.get("op", "equalize" if sec == "equalize" else "clahe")
            tx, ty = k.get("tiles", [1, 1])
            lines.append(f"{k['id']} {op} {h} {w} {float(k.get('clip', 0.0))!r} {tx} {ty} " + " ".join(map(str, src.reshape(-1))) + " | "
                         + " ".join(map(str, dst.reshape(-1))))
    path.write_text("\n".join(lines) + "\n")
    return len(lines) - 1


def test_kat_txt_carries_every_known_answer_with_pixels(tmp_path):
    """The text form test_adapter_opencv reads: every kat.json entry that states pixels, and its expansion agrees with the oracle (so a
    disagreement with OpenCV on the other machine is OpenCV-vs-Appendix-A, not a conversion slip)."""
    import oracle
    n = write_kat_txt(tmp_path / "kat.txt")
    assert n >= 23
    seen = set()
    for ln in (tmp_path / "kat.txt").read_text().splitlines():
        if ln.startswith("#"):
            continue
        head, _, tail = ln.partition(" | ")
        f = head.split()
        kid, op, h, w, clip, tx, ty = f[0], f[1], int(f[2]), int(f[3]), float(f[4]), int(f[5]), int(f[6])
        src = np.array(f[7:], np.uint8).reshape(h, w)
        dst = np.array(tail.split(), np.uint8).reshape(h, w)
        got = oracle.equalize_hist(src) if op == "equalize" else oracle.clahe(src, clip, tx, ty)
        assert np.array_equal(got, dst), kid
        seen.add(kid)
    assert {"EQ-1", "EQ-2", "EQ-7", "CL-1", "CL-4", "CL-5", "CL-6", "CL-19"} <= seen


def test_opencv_pin_programs_compile_against_declarations(tmp_path):
    """Both programs and the interposer go through a compiler against DECLARATION-ONLY OpenCV headers (not OpenCV; pins nothing): the
    API they touch exists with those signatures, -Wall -Wextra -Werror clean.  The Makefile rule that builds the real thing is guarded
    by pkg-config and says so when there is no OpenCV."""
    decl = CXX_TESTS / "opencv_decl"
    cxx = ROOT / "opencv-opencl_amd" / "cxx"
    for src, extra in (("test_adapter_opencv.cpp", [f"-I{cxx}"]), ("interpose_probe.cpp", [])):
        r = subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-c", f"-I{decl}", *extra, str(CXX_TESTS / src), "-o", str(tmp_path / (src + ".o"))],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    obj = tmp_path / "interpose.o"
    r = subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-fPIC", "-c", f"-I{decl}", str(cxx / "interpose" / "mi_cv_interpose.cpp"), "-o", str(obj)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    syms = subprocess.run(["nm", str(obj)], capture_output=True, text=True, check=True).stdout
    defined = {ln.split()[-1] for ln in syms.splitlines() if " T " in ln}
    assert "mi_cv_interpose_calls" in defined                       # the counter the probe looks up
    if not have_opencv():
        r = subprocess.run(["make", "-C", str(CXX_TESTS), "opencv"], capture_output=True, text=True)
        assert r.returncode == 0 and "pkg-config does not know opencv4" in r.stdout, r.stdout + r.stderr
        assert not (CXX_TESTS / "test_adapter_opencv").exists()


@pytest.mark.gpu
def test_real_opencv_pin_record(tmp_path):
    """The pin itself.  Skips where there is no OpenCV; elsewhere builds the two programs and the interposer, runs them, writes
    gpurun_out/opencv_pin.json and fails on any check that did not pass."""
    if not have_opencv():
        pytest.skip("pkg-config does not know opencv4 on this machine: the pin record cannot be produced here (SURVEY 8c)")
    r = subprocess.run(["make", "-C", str(CXX_TESTS), "opencv"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    out_dir = ROOT / "gpurun_out"
    out_dir.mkdir(exist_ok=True)
    pin = out_dir / "opencv_pin.json"
    kat = tmp_path / "kat.txt"
    write_kat_txt(kat)
    run = subprocess.run([str(CXX_TESTS / "test_adapter_opencv"), str(pin), str(kat)], capture_output=True, text=True, timeout=1200)
    rec = json.loads(pin.read_text())
    # the unmodified OpenCV program, plain and under the interposer (in the arithmetic mode this OpenCV's CLAHE turned out to use)
    env = dict(os.environ)
    plain = json.loads(subprocess.run([str(CXX_TESTS / "interpose_probe")], capture_output=True, text=True, timeout=300, env=env, check=True).stdout)
    env["LD_PRELOAD"] = str(ROOT / "opencv-opencl_amd" / "lib" / "libmi_cv_interpose.so")
    env["MI_CV_CLAHE_FP_CONTRACT"] = "1" if rec["clahe_arithmetic_mode_matched"].startswith("fused") else "0"
    pre = subprocess.run([str(CXX_TESTS / "interpose_probe")], capture_output=True, text=True, timeout=300, env=env)
    taken = json.loads(pre.stdout) if pre.returncode == 0 else {"error": pre.stderr[-2000:]}
    rec["interposer"] = {"plain": plain, "under_LD_PRELOAD": taken}
    checks = rec["checks"]
    checks.append({"name": "unmodified program: the interposer is not loaded in a plain run", "pass": plain["interposer_loaded"] is False, "note": ""})
    checks.append({"name": "unmodified program under LD_PRELOAD: cv::equalizeHist and cv::createCLAHE were taken by the interposer",
                   "pass": taken.get("interposer_loaded") is True and taken.get("equalizeHist_taken", 0) >= 1 and taken.get("createCLAHE_taken", 0) >= 1, "note": ""})
    checks.append({"name": "unmodified program under LD_PRELOAD: same bytes as with OpenCV's own functions",
                   "pass": taken.get("equalize_fnv") == plain["equalize_fnv"] and taken.get("clahe_fnv") == plain["clahe_fnv"], "note": ""})
    rec["failed"] = sum(1 for c in checks if not c["pass"])
    pin.write_text(json.dumps(rec, indent=1))
    print(run.stdout[-4000:])
    assert rec["failed"] == 0, [c for c in checks if not c["pass"]]
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-2000:]


@pytest.mark.gpu
def test_bench_opencv_cross_check_plumbing_with_a_stand_in(monkeypatch):
    """bench.py's `opencv_cross_check` only runs where cv2 is importable -- never on this pool -- so its plumbing (shapes, the NV12 / I420
    index arithmetic of config 5's literal reading, the option it flips and restores) would meet its first input on somebody else's
    machine.  Here it runs against a STAND-IN module named cv2 whose five functions are the oracle's: this exercises the code path
    and pins nothing (the oracle agreeing with the library is what every other GPU test shows)."""
    import types
    import oracle
    import mi_lumaeq
    import bench
    fake = types.ModuleType("cv2")
    fake.__version__ = "stand-in (oracle): pins nothing"
    fake.COLOR_YUV2BGR_NV12, fake.COLOR_BGR2YUV_I420 = 91, 128
    fake.equalizeHist = lambda a: oracle.equalize_hist(np.ascontiguousarray(a))

    class _Clahe:
        def __init__(self, clip, tiles): self.clip, self.tiles = clip, tiles
        def apply(self, a): return oracle.clahe(np.ascontiguousarray(a), self.clip, self.tiles[0], self.tiles[1])
    fake.createCLAHE = lambda clip, tiles: _Clahe(clip, tiles)

    def cvt(a, code):
        if code == fake.COLOR_YUV2BGR_NV12:
            return oracle.nv12_to_bgr(np.ascontiguousarray(a), a.shape[1], a.shape[0] * 2 // 3)
        assert code == fake.COLOR_BGR2YUV_I420
        return oracle.bgr_to_i420(np.ascontiguousarray(a))
    fake.cvtColor = cvt
    fake.split = lambda a: [np.ascontiguousarray(a[:, :, k]) for k in range(a.shape[2])]
    fake.merge = lambda planes: np.ascontiguousarray(np.stack(planes, axis=2))
    monkeypatch.setitem(sys.modules, "cv2", fake)
    with mi_lumaeq.Context(0) as ctx:
        res = bench.opencv_cross_check(ctx, 3840, 2160, "D2")
    assert "error" not in res, res
    assert set(res["shapes"]) == {"1920x1080", "3840x2160"}
    for shape, r in res["shapes"].items():
        assert r["equalizeHist_bit_exact"] and r["clahe_2.0_8x8_bit_exact"], (shape, r)
        assert r["config5_y_equalize_uv_passthrough_bit_exact"] and r["config5_literal_bgr_channels_bit_exact"], (shape, r)
    assert res["equalizeHist_bit_exact"] is True
