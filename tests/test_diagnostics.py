"""The GPU session's crash diagnostics (tests/cxx/abrt_trace.c, loaded by conftest.py): a fatal signal must leave the native call
chain of the faulting thread in a FILE, because under a capturing test runner stderr may not survive the process."""
import signal
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
TRACER = ROOT / "tests" / "cxx" / "libabrt_trace.so"


@pytest.mark.skipif(not TRACER.exists(), reason="tests/cxx/libabrt_trace.so not built (__graft_entry__.build())")
def test_abrt_tracer_leaves_the_native_chain_in_a_file(tmp_path):
    log = tmp_path / "trace.log"
    code = ("import ctypes, os, threading\n"
            f"ctypes.CDLL({str(TRACER)!r})\n"
            "t = threading.Thread(target=os.abort)\n"            # the abort comes from a thread that is not the main one
            "t.start(); t.join()\n")
    r = subprocess.run([sys.executable, "-c", code], env={"MI_ABRT_TRACE_FILE": str(log), "PATH": "/usr/bin:/bin"},
                       capture_output=True, text=True, timeout=60)
    assert r.returncode == -signal.SIGABRT
    text = log.read_text()
    assert "[abrt_trace] fatal signal" in text and "abort" in text, text
    assert "[abrt_trace] fatal signal" in r.stderr                # and on stderr, when there is one
    # the locked-memory state at the moment of death (the one abort site on record pins caller pages: DESIGN.md "the abort")
    assert "RLIMIT_MEMLOCK soft:" in text and "VmLck:" in text and "VmPin:" in text, text
