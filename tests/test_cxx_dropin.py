"""Drop-in C++ headers (include/sdrlib/*.hpp): the reference's gtest cases restated in
tests/cxx/dropin_test.cpp, compiled with the host compiler against libsdr_amd.so."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cxx", "dropin_test")


HOST_EXE = os.path.join(ROOT, "tests", "cxx", "host_test")
DEMO_EXE = os.path.join(ROOT, "tests", "cxx", "ofdm_demo")
MULTI_EXE = os.path.join(ROOT, "tests", "cxx", "ber_sweep_multi")


def _build():
    if not all(os.path.exists(e) for e in (EXE, HOST_EXE, DEMO_EXE, MULTI_EXE)):
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


@pytest.mark.gpu
def test_native_multi_gpu_sweep_equals_python_sweep(ctx):
    """examples/ber_sweep_multi.cpp: the Eb/N0 sweep driven from C++ through libsdr_amd_multi.so (one context and one
    host thread per GPU, ncclCommInitAll, one ncclAllReduce of the counters) on the GPUs of this box.  Same frame plan
    and random streams as ber_sweep.py, so its counters equal those of the Python sweep bit for bit."""
    import json
    import numpy as np
    from sdr_amd import link
    _build()
    p = subprocess.run([MULTI_EXE, "--bits", "6e7", "--n-fft", "512", "--cp", "128", "--mods", "2", "4", "6", "--step-db", "5", "--seed", "5"],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    ebn0 = {2: [0.0, 5.0, 10.0], 4: [0.0, 5.0, 10.0, 15.0], 6: [0.0, 5.0, 10.0, 15.0, 20.0]}
    pts = link.plan_sweep([2, 4, 6], ebn0, 512, 6e7)
    want = ctx.link_sweep(512, 128, 5, link.rank_points(pts, 512, 0, 1)).cpu().numpy()
    assert line["points"] == len(pts) and line["n_gpus"] >= 1
    assert np.array_equal(np.array(line["counters"], dtype=np.int64), want)
    assert line["checksum_bit_errors"] == int(want[:, 0].sum()) > 0 and "ncclAllReduce" in line["collective"]
    # the faded link (configs[2] shape) the same way: counters equal the one-GPU sdr_link_chain_c32 run
    n_sym = 14 * 300
    p = subprocess.run([MULTI_EXE, "--chain", str(n_sym)], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    chain = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    cnt, _, _ = ctx.link_chain(2048, 512, 6, 3, 0, n_sym, ((1 << 18) // 14) * 14, 14, link.noise_sigma(2048, 6, 18.0), 6 * 10 ** 1.8, want_llr=True)
    k = cnt.cpu().tolist()
    assert (chain["bit_errors"], chain["bits"], chain["ofdm_symbols"]) == (k[0], k[1], k[4]) and k[0] > 0
