"""CPU checks of the C ABI: the library loads, exports every symbol include/mi_lumaeq.h declares,
and FAILS LOUDLY without a GPU (no compute calls are made here)."""
import ctypes
import re
from pathlib import Path

import pytest

import mi_lumaeq

ROOT = Path(__file__).resolve().parents[1]


def _declared_in_header():
    """every function any header under include/ declares (mi_lumaeq_tuning.h declares none: option names only)"""
    names = set()
    for hdr in sorted((ROOT / "include").glob("*.h")):
        txt = re.sub(r"/\*.*?\*/", "", hdr.read_text(), flags=re.S)
        names |= set(re.findall(r"\b(mi_[a-z0-9_]+)\s*\(", txt))
    return sorted(names)


def test_header_and_binding_agree():
    assert _declared_in_header() == sorted(mi_lumaeq.DECLARED_SYMBOLS)


def test_library_exports_every_declared_symbol(built_lib):
    for s in _declared_in_header():
        assert hasattr(built_lib, s), f"libmi_lumaeq.so does not export {s}"


TEST_HOOK_OPTIONS = ("fused_fault_inject", "fused_timeout_us", "hip_fail_after")


def test_product_library_contains_no_test_hook(built_lib):
    """The fault-injection / forced-failure hooks are compiled into libmi_lumaeq_test.so only (-DMI_TEST_HOOKS): the product library
    must know none of their option names (nor the retired "host_direct"), export exactly the same C ABI, and the public headers must
    not mention them."""
    prod = mi_lumaeq.lib_path().read_bytes()
    test_path = mi_lumaeq.lib_path().with_name("libmi_lumaeq_test.so")
    assert test_path.exists(), "build(): make -C opencv-opencl_amd/csrc builds both libraries"
    test = test_path.read_bytes()
    for name in TEST_HOOK_OPTIONS:
        assert name.encode() not in prod, f"product library knows the test hook {name!r}"
        assert name.encode() in test, f"test library lacks the hook {name!r}"
    assert b"host_direct" not in prod and b"host_direct" not in test
    assert b"+test-hooks" in test and b"+test-hooks" not in prod
    tl = mi_lumaeq.test_lib()
    for s in _declared_in_header():
        assert hasattr(tl, s), f"libmi_lumaeq_test.so does not export {s}"
    for hdr in (ROOT / "include").glob("*.h"):
        txt = hdr.read_text()
        for name in TEST_HOOK_OPTIONS + ("host_direct",):
            assert name not in txt, f"{hdr.name} documents {name!r}"
    # the speed-only options live in their own header, the behaviour options in the main one
    tuning = (ROOT / "include" / "mi_lumaeq_tuning.h").read_text()
    main = (ROOT / "include" / "mi_lumaeq.h").read_text()
    for name in ("fused_wgs_per_cu", "fused_vpt", "clahe_xcd_map", "clahe_hist_threads", "clahe_seg_pairs", "clahe_tiles_per_wg",
                 "clahe_float_tables", "clahe16_transposed", "clahe16_wide", "bgr_fused", "host_copy_threads", "host_copy_streams", "pipe_copy_streams",
                 "two_kernel_max_frames", "clahe16_fast12"):
        assert f'"{name}"' in tuning and f'"{name}"' not in main, name
        assert name.encode() in prod
    for name in ("fused", "fused_timeout_ms", "fused_demote_after", "fused_reprobe_ms", "clahe_fp_contract"):
        assert f'"{name}"' in main, name


def test_version_and_status_strings(built_lib):
    assert "gfx950" in mi_lumaeq.version()
    assert mi_lumaeq.status_str(0) == "MI_OK"
    assert mi_lumaeq.status_str(5) == "MI_ERR_NO_DEVICE" and mi_lumaeq.status_str(6) == "MI_ERR_BUSY"
    assert [built_lib.mi_kernel_name(k).decode() for k in range(len(mi_lumaeq.KERNEL_NAMES))] == mi_lumaeq.KERNEL_NAMES


def test_no_cpu_fallback(built_lib):
    """Without a HIP device context creation must fail (status, not a silent CPU path)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu tests")
    assert mi_lumaeq.device_count() == 0
    with pytest.raises(mi_lumaeq.MiError) as e:
        mi_lumaeq.Context(0)
    assert e.value.status == 5
    # null-context calls are rejected, they do not compute
    assert built_lib.mi_equalize_hist_u8(None, None, 0, None, 0, 4, 4) == 1


def test_product_never_imports_oracle():
    """The shipped path (package, bench GPU leg, C/C++ sources) must not reference oracle/."""
    pkg = ROOT / "opencv-opencl_amd"
    for p in pkg.rglob("*"):
        if p.is_file() and p.suffix in {".py", ".hip", ".h", ".hpp", ".cpp", ".c"}:
            txt = p.read_text(errors="ignore")
            assert "import oracle" not in txt and "from oracle" not in txt, p
            assert "liblumaeq_oracle" not in txt and not re.search(r"\borc_\w+\s*\(", txt), p


def test_header_is_plain_c():
    """The drop-in boundary is a C ABI: include/mi_lumaeq.h must compile as C99 (no C++-isms, no torch / HIP types) and as C++11."""
    import shutil
    import subprocess
    from pathlib import Path
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    for hdr in sorted((Path(__file__).resolve().parents[1] / "include").glob("*.h")):
        for cmd in (["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c", str(hdr)],
                    ["g++", "-std=c++11", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-x", "c++", str(hdr)]):
            r = subprocess.run(cmd, capture_output=True, text=True)
            assert r.returncode == 0, r.stderr
