"""CPU sanitizer runs ('-fsanitize=address,undefined for the host oracle build'):
the oracle's pin tests against an AddressSanitizer + UBSan build of oracle/sdr_oracle.cpp, and the
host-only header test built the same way.  Host code only -- GPU sanitizers do not exist on this pool."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _libasan():
    p = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True)
    path = p.stdout.strip()
    return path if p.returncode == 0 and os.path.isabs(path) and os.path.exists(path) else None


def test_host_header_test_under_asan_ubsan():
    if _libasan() is None:
        pytest.skip("no libasan in this image")
    subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "cxx"), "asan"], check=True, capture_output=True)
    p = subprocess.run([os.path.join(ROOT, "tests", "cxx", "host_test_asan")], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert "all host checks passed" in p.stdout and "runtime error" not in p.stderr and "AddressSanitizer" not in p.stderr


def test_oracle_pins_under_asan_ubsan():
    asan = _libasan()
    if asan is None:
        pytest.skip("no libasan in this image")
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], check=True, capture_output=True)
    lib = os.path.join(ROOT, "oracle", "_asan", "libsdr_oracle.so")
    env = dict(os.environ)
    # the child python gets the sanitizer runtime ahead of its own libraries; python itself leaks by design
    env["LD_PRELOAD"] = asan + ((":" + env["LD_PRELOAD"]) if env.get("LD_PRELOAD") else "")
    env["ASAN_OPTIONS"] = "detect_leaks=0:abort_on_error=0"
    env["UBSAN_OPTIONS"] = "print_stacktrace=1:halt_on_error=1"
    env["SDR_ORACLE_LIB"] = lib
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_oracle_pins.py"), "-x", "-q", "-p", "no:cacheprovider"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-4000:]
    assert "passed" in p.stdout and "runtime error" not in p.stderr and "AddressSanitizer" not in p.stderr
