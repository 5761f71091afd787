"""Drop-in C++ headers (include/sdrlib/*.hpp): the reference's gtest cases restated in
tests/cxx/dropin_test.cpp, compiled with the host compiler against libsdr_amd.so."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cxx", "dropin_test")


HOST_EXE = os.path.join(ROOT, "tests", "cxx", "host_test")
DEMO_EXE = os.path.join(ROOT, "tests", "cxx", "ofdm_demo")


def _build():
    if not all(os.path.exists(e) for e in (EXE, HOST_EXE, DEMO_EXE)):
        subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "cxx")], check=True, capture_output=True)


def test_host_only_headers():
    """utils::sliding_buffer (include/sdrlib/sliding_buffer.hpp) needs no device."""
    _build()
    p = subprocess.run([HOST_EXE], capture_output=True, text=True, timeout=60)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "all host checks passed" in p.stdout


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


@pytest.mark.gpu
def test_headless_demo_shows_the_payload(tmp_path):
    """examples/ofdm_demo.cpp = app/QFDemoWindow.cpp:201-266 without the window: after k ticks the
    50-character running text is the last 50 decoded payload characters."""
    _build()
    payload = ("Hello, world! " "I am a Software-Defined Radio Stack.          "
               "This string is a result of demultiplexing a 16-QAM " "multiplexed OFDM signal.          ")
    ticks = 60
    t_csv, c_csv = tmp_path / "time.csv", tmp_path / "constl.csv"
    p = subprocess.run([DEMO_EXE, "--ticks", str(ticks), "--time-csv", str(t_csv), "--constellation-csv", str(c_csv)],
                       capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    stream = "".join(payload[i % len(payload)] for i in range(4 * ticks))
    assert p.stdout.rstrip("\n") == stream[-50:]
    assert len(t_csv.read_text().splitlines()) == 513          # header + the 512-sample window
    pts = c_csv.read_text().splitlines()[1:]
    assert len(pts) == 8                                        # 4 bytes = 8 16-QAM points per tick
