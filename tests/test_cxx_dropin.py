"""Drop-in C++ headers (include/sdrlib/*.hpp): the reference's gtest cases restated in
tests/cxx/dropin_test.cpp, compiled with the host compiler against libsdr_amd.so."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cxx", "dropin_test")


def _build():
    if not os.path.exists(EXE):
        subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "cxx")], check=True, capture_output=True)


def test_headers_compile_and_report_missing_device():
    import torch
    _build()
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    p = subprocess.run([EXE, "--expect-no-device"], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "ok (no device)" in p.stdout


@pytest.mark.gpu
def test_reference_cases_through_dropin_headers():
    _build()
    p = subprocess.run([EXE], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert "all drop-in checks passed" in p.stdout
