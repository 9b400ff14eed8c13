"""CPU checks of the C ABI: the library loads, exports every symbol include/mi_lumaeq.h declares,
and FAILS LOUDLY without a GPU (no compute calls are made here)."""
import ctypes
import re
from pathlib import Path

import pytest

import mi_lumaeq

ROOT = Path(__file__).resolve().parents[1]


def _declared_in_header():
    txt = (ROOT / "include" / "mi_lumaeq.h").read_text()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mi_[a-z0-9_]+)\s*\(", txt)))


def test_header_and_binding_agree():
    assert _declared_in_header() == sorted(mi_lumaeq.DECLARED_SYMBOLS)


def test_library_exports_every_declared_symbol(built_lib):
    for s in _declared_in_header():
        assert hasattr(built_lib, s), f"libmi_lumaeq.so does not export {s}"


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
    hdr = Path(__file__).resolve().parents[1] / "include" / "mi_lumaeq.h"
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    for cmd in (["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c", str(hdr)],
                ["g++", "-std=c++11", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-x", "c++", str(hdr)]):
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
