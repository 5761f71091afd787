"""CPU-only checks of the C ABI: the shared library loads, exports every symbol that
include/sdr_amd.h and include/sdr_amd_diag.h declare, and its host-side pieces (twiddle generator, constellation tables,
Philox, error text) agree with the oracle.  No GPU compute is called here."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = "".join(open(os.path.join(ROOT, "include", h)).read() for h in ("sdr_amd.h", "sdr_amd_diag.h"))
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sdr_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from sdr_amd import capi
    lib = capi.load()
    names = declared_symbols()
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), f"libsdr_amd.so does not export {n}"
    assert sorted(capi.SIGNATURES) == names, "sdr_amd/capi.py and include/sdr_amd.h disagree"


def test_struct_layouts_match_header():
    from sdr_amd import capi
    assert ctypes.sizeof(capi.DemodArgs) == 104 and capi.DemodArgs.flags.offset == 96      # 96 = the round-2 layout, still accepted
    assert ctypes.sizeof(capi.LinkArgs) == 88
    assert ctypes.sizeof(capi.PipeConfig) == 56
    assert ctypes.sizeof(capi.PipeResult) == 48


@pytest.mark.parametrize("n", [2, 8, 64, 1024, 4096])
@pytest.mark.parametrize("inverse", [0, 1])
def test_twiddle_tables_equal_oracle(oracle, n, inverse):
    from sdr_amd import capi
    lib = capi.load()
    t32 = np.zeros(n - 1, dtype=np.complex64)
    capi.check(lib.sdr_twiddle_table_c32(n, inverse, t32.ctypes.data_as(ctypes.c_void_p)))
    assert np.array_equal(t32, oracle.twiddles(n, bool(inverse), np.complex64))
    t64 = np.zeros(n - 1, dtype=np.complex128)
    capi.check(lib.sdr_twiddle_table_c64(n, inverse, t64.ctypes.data_as(ctypes.c_void_p)))
    assert np.array_equal(t64, oracle.twiddles(n, bool(inverse), np.complex128))


def test_rng_tables_equal_oracle(oracle):
    """The Gaussian generator's tables (rng.hpp) are built on the host by the library and by the oracle with the same
    formulas: bit-equal, like the twiddle tables."""
    from sdr_amd import capi
    lib = capi.load()
    t = np.zeros(576, dtype=np.float32)
    capi.check(lib.sdr_rng_tables_f32(t.ctypes.data_as(ctypes.c_void_p)))
    assert np.array_equal(t, oracle.rng_tables())


def test_non_pow2_error_text():
    from sdr_amd import capi
    lib = capi.load()
    buf = np.zeros(16, dtype=np.complex64)
    rc = lib.sdr_twiddle_table_c32(12, 0, buf.ctypes.data_as(ctypes.c_void_p))
    assert rc == capi.SDR_ERR_NOT_POW2
    assert capi.last_error() == "The sequence size must be of powers of 2"     # reference lib/inc/fft.hpp:92


@pytest.mark.parametrize("bps", [1, 2, 4, 6, 8, 10])
def test_constellation_tables_equal_oracle(oracle, bps):
    from sdr_amd import capi
    lib = capi.load()
    nrm, inrm = ctypes.c_float(), ctypes.c_float()
    tab = np.zeros(1 << bps, dtype=np.complex64)
    capi.check(lib.sdr_constellation_f32(bps, ctypes.byref(nrm), ctypes.byref(inrm), tab.ctypes.data_as(ctypes.c_void_p)))
    onrm, oinrm, otab = oracle.constellation(bps)
    assert np.float32(nrm.value) == onrm and np.float32(inrm.value) == oinrm and np.array_equal(tab, otab)


def test_host_philox_equals_oracle(oracle):
    from sdr_amd import capi
    lib = capi.load()
    rng = np.random.default_rng(3)
    for _ in range(50):
        c = rng.integers(0, 2**32, 4, dtype=np.uint64).astype(np.uint32)
        k = rng.integers(0, 2**32, 2, dtype=np.uint64).astype(np.uint32)
        o = np.zeros(4, dtype=np.uint32)
        P = ctypes.POINTER(ctypes.c_uint32)
        capi.check(lib.sdr_philox4x32_10(c.ctypes.data_as(P), k.ctypes.data_as(P), o.ctypes.data_as(P)))
        assert np.array_equal(o, oracle.philox(c, k))


def test_no_device_is_a_loud_error():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from sdr_amd import capi
    lib = capi.load()
    h = ctypes.c_void_p()
    rc = lib.sdr_create(0, ctypes.byref(h))
    assert rc != 0 and not h.value
    assert capi.last_error()
    from sdr_amd.ops import Context
    with pytest.raises(RuntimeError):
        Context(0)


def test_product_does_not_touch_the_oracle():
    # the product (sdr_amd/, include/) must never reference the oracle
    for base in ("sdr_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp", "Makefile")):
                    text = open(os.path.join(dirpath, f), errors="replace").read()
                    assert "oracle" not in text.lower(), os.path.join(dirpath, f)


def test_multi_gpu_library_exports_every_declared_symbol():
    """libsdr_amd_multi.so (the native multi-GPU driver, links RCCL) loads without a GPU and exports what
    include/sdr_amd_multi.h declares; without a device its constructor fails with an error value."""
    text = open(os.path.join(ROOT, "include", "sdr_amd_multi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = sorted(set(re.findall(r"\b(sdr_multi_[a-z0-9_]+)\s*\(", text)))
    assert len(names) >= 8
    lib = ctypes.CDLL(os.path.join(ROOT, "sdr_amd", "lib", "libsdr_amd_multi.so"))
    for n in names:
        assert hasattr(lib, n), f"libsdr_amd_multi.so does not export {n}"
    import torch
    if not torch.cuda.is_available():
        lib.sdr_multi_last_error.restype = ctypes.c_char_p
        h = ctypes.c_void_p()
        dev = (ctypes.c_int * 1)(0)
        assert lib.sdr_multi_create(dev, 1, ctypes.byref(h)) != 0 and not h.value
        assert b"GPU 0" in lib.sdr_multi_last_error()
        assert lib.sdr_multi_create(dev, 0, ctypes.byref(h)) != 0
