"""ctypes binding of oracle/libsdr_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
nothing under sdr_amd/ does.  numpy in, numpy out.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from ctypes import c_float, c_int, c_size_t, c_uint32, c_uint64, c_void_p, c_longlong, c_double

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SDR_ORACLE_LIB") or os.path.join(_HERE, "libsdr_oracle.so")   # SDR_ORACLE_LIB: the sanitizer build
POW2_MESSAGE = "The sequence size must be of powers of 2"

_lib = None


def build() -> None:
    subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = ctypes.CDLL(LIB_PATH)
        L.orc_pow2_error.restype = ctypes.c_char_p
        L.orc_ln_unit.restype = c_float
        L.orc_ln_unit.argtypes = [c_float]
        L.orc_from_constl_f32.restype = c_longlong
        L.orc_from_constl_f64.restype = c_longlong
        L.orc_tx_index.restype = ctypes.c_uint
        L.orc_count_bit_errors.restype = c_uint64
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(c_void_p)


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def _cplx_dtype(a):
    return np.complex128 if np.asarray(a).dtype == np.complex128 else np.complex64


def fft2(x, inverse=False):
    """fft::fft2 / ifft2 on the last axis; returns (rc, out)."""
    dt = _cplx_dtype(x)
    out = _c(x, dt).copy()
    n = out.shape[-1] if out.ndim else 1
    batch = out.size // n if n else 0
    fn = lib().orc_fft2_batch_c32 if dt == np.complex64 else lib().orc_fft2_batch_c64
    rc = fn(_p(out), c_size_t(n), c_size_t(batch), c_int(int(inverse)))
    return rc, out


def twiddles(n, inverse=False, dtype=np.complex64):
    out = np.zeros(max(n - 1, 0), dtype=dtype)
    fn = lib().orc_twiddles_c32 if dtype == np.complex64 else lib().orc_twiddles_c64
    fn(c_size_t(n), c_int(int(inverse)), _p(out))
    return out


def ofdm_rx(x, n_fft, cp):
    dt = _cplx_dtype(x)
    x = _c(x, dt).reshape(-1, n_fft + cp)
    out = np.empty((x.shape[0], n_fft), dtype=dt)
    fn = lib().orc_ofdm_rx_c32 if dt == np.complex64 else lib().orc_ofdm_rx_c64
    for s in range(x.shape[0]):
        rc = fn(_p(x[s]), c_size_t(n_fft + cp), c_size_t(cp), _p(out[s]))
        if rc:
            return rc, out
    return 0, out


def ofdm_tx(x, n_fft, cp):
    dt = _cplx_dtype(x)
    x = _c(x, dt).reshape(-1, n_fft)
    out = np.zeros((x.shape[0], n_fft + cp), dtype=dt)
    fn = lib().orc_ofdm_tx_c32 if dt == np.complex64 else lib().orc_ofdm_tx_c64
    for s in range(x.shape[0]):
        rc = fn(_p(x[s]), c_size_t(n_fft), c_size_t(cp), _p(out[s]))
        if rc:
            return rc, out
    return 0, out


def constellation(bps, dtype=np.float32):
    tab = np.zeros(1 << bps, dtype=np.complex64 if dtype == np.float32 else np.complex128)
    if dtype == np.float32:
        nrm, inrm = c_float(), c_float()
        rc = lib().orc_constellation_f32(c_int(bps), ctypes.byref(nrm), ctypes.byref(inrm), _p(tab))
    else:
        nrm, inrm = c_double(), c_double()
        rc = lib().orc_constellation_f64(c_int(bps), ctypes.byref(nrm), ctypes.byref(inrm), _p(tab))
    assert rc == 0
    return dtype(nrm.value), dtype(inrm.value), tab


def nearest(bps, sym):
    dt = _cplx_dtype(sym)
    sym = _c(sym, dt).ravel()
    out = np.zeros(sym.size, dtype=np.uint16)
    fn = lib().orc_nearest_f32 if dt == np.complex64 else lib().orc_nearest_f64
    assert fn(c_int(bps), _p(sym), c_size_t(sym.size), _p(out)) == 0
    return out


def to_constl(bps, data, n_sym=None, dtype=np.complex64):
    data = _c(data, np.uint8).ravel()
    if n_sym is None:
        n_sym = data.size * 8 // bps
    out = np.zeros(n_sym, dtype=dtype)
    fn = lib().orc_to_constl_f32 if dtype == np.complex64 else lib().orc_to_constl_f64
    assert fn(c_int(bps), _p(data), c_size_t(n_sym), _p(out)) == 0
    return out


def from_constl(bps, sym):
    dt = _cplx_dtype(sym)
    sym = _c(sym, dt).ravel()
    out = np.zeros(sym.size * bps // 8, dtype=np.uint8)
    fn = lib().orc_from_constl_f32 if dt == np.complex64 else lib().orc_from_constl_f64
    n = fn(c_int(bps), _p(sym), c_size_t(sym.size), _p(out))
    assert n == out.size
    return out


def soft_demap(bps, sym, weight=1.0):
    sym = _c(sym, np.complex64).ravel()
    out = np.zeros((sym.size, bps), dtype=np.float32)
    wt = _c(weight, np.float32).ravel() if isinstance(weight, np.ndarray) else None
    w0 = 0.0 if wt is not None else float(weight)
    assert lib().orc_soft_demap_f32(c_int(bps), _p(sym), c_size_t(sym.size), _p(wt), c_float(w0), _p(out)) == 0
    return out


def zf_taps(bps, chan, inv_n0):
    chan = _c(chan, np.complex64).ravel()
    taps = np.zeros_like(chan)
    wgt = np.zeros(chan.size, dtype=np.float32)
    assert lib().orc_zf_taps_f32(c_int(bps), _p(chan), c_size_t(chan.size), c_float(inv_n0), _p(taps), _p(wgt)) == 0
    return taps, wgt


def demod(rx, n_fft, cp, bps, eq=None, weight=1.0, sym_per_set=0, want_fft=True, want_llr=True, want_hard=True, threads=1, weight_from_taps=False):
    """Fused receive path on the CPU.  Returns dict(fft, llr, hard).  threads > 1 splits the symbols
    over host threads (same results; the reference itself runs on one)."""
    rx = _c(rx, np.complex64).ravel()
    n_sym = rx.size // (n_fft + cp)
    fft = np.zeros((n_sym, n_fft), dtype=np.complex64) if want_fft else None
    llr = np.zeros((n_sym, n_fft, bps), dtype=np.float32) if want_llr else None
    hard = np.zeros(n_sym * n_fft * bps // 8, dtype=np.uint8) if want_hard else None
    eqa = _c(eq, np.complex64).ravel() if eq is not None else None
    wt = _c(weight, np.float32).ravel() if isinstance(weight, np.ndarray) else None
    w0 = 0.0 if wt is not None else float(weight)
    rc = lib().orc_demod_c32_threads(ctypes.c_uint(threads), c_uint32(n_fft), c_uint32(cp), c_int(bps), c_size_t(n_sym), c_uint32(sym_per_set),
                                     _p(rx), _p(eqa), _p(wt), c_float(w0), _p(fft), _p(llr), _p(hard), ctypes.c_uint(1 if weight_from_taps else 0))
    assert rc == 0, rc
    return {"fft": fft, "llr": llr, "hard": hard}


def philox(ctr, key):
    c = np.asarray(ctr, dtype=np.uint32)
    k = np.asarray(key, dtype=np.uint32)
    o = np.zeros(4, dtype=np.uint32)
    lib().orc_philox4x32_10(_p(c), _p(k), _p(o))
    return o


def ln_unit(x):
    return float(lib().orc_ln_unit(c_float(float(x))))


def sincos_turn(k24):
    o = np.zeros(2, dtype=np.float32)
    lib().orc_sincos_turn(c_uint32(int(k24)), _p(o))
    return float(o[0]), float(o[1])


def gaussian_pair(x1, x2, sigma=1.0):
    """Box-Muller pair of the generator from two 32-bit Philox words: (re, im)."""
    o = np.zeros(2, dtype=np.float32)
    lib().orc_gaussian_pair(c_uint32(int(x1)), c_uint32(int(x2)), c_float(float(sigma)), _p(o))
    return float(o[0]), float(o[1])


def rng_tables():
    """The generator's tables: 32 (r_i, L_i) pairs, then 256 (sin, cos) pairs; 576 floats."""
    o = np.zeros(576, dtype=np.float32)
    lib().orc_rng_tables(_p(o))
    return o


def link_generate(n_fft, cp, bps, seed, g0, n_sym, sigma, chan=None, sym_per_set=0):
    rx = np.zeros((n_sym, n_fft + cp), dtype=np.complex64)
    bits = np.zeros(n_sym * n_fft * bps // 8, dtype=np.uint8)
    ch = _c(chan, np.complex64).ravel() if chan is not None else None
    rc = lib().orc_link_generate_c32(c_uint32(n_fft), c_uint32(cp), c_int(bps), c_uint64(seed), c_uint64(g0), c_size_t(n_sym),
                                     c_uint32(sym_per_set), c_float(sigma), _p(ch), _p(rx), _p(bits))
    assert rc == 0, rc
    return rx, bits


def link_ber(n_fft, cp, bps, seed, g0, n_sym, sigma, w0=1.0, counters=None):
    if counters is None:
        counters = np.zeros(8, dtype=np.uint64)
    rc = lib().orc_link_ber_c32(c_uint32(n_fft), c_uint32(cp), c_int(bps), c_uint64(seed), c_uint64(g0), c_size_t(n_sym),
                                c_float(sigma), c_float(w0), _p(counters))
    assert rc == 0, rc
    return counters


def link_channel(n_fft, seed, first_set, n_sets):
    chan = np.zeros((n_sets, n_fft), dtype=np.complex64)
    rc = lib().orc_link_channel_c32(c_uint32(n_fft), c_uint64(seed), c_uint64(first_set), c_size_t(n_sets), _p(chan))
    assert rc == 0, rc
    return chan


def link_chain(n_fft, cp, bps, seed, g0, n_sym, sym_per_set, sigma, inv_n0, counters=None, want_llr=False):
    if counters is None:
        counters = np.zeros(8, dtype=np.uint64)
    llr = np.zeros((n_sym, n_fft, bps), dtype=np.float32) if want_llr else None
    rc = lib().orc_link_chain_c32(c_uint32(n_fft), c_uint32(cp), c_int(bps), c_uint64(seed), c_uint64(g0), c_size_t(n_sym), c_uint32(sym_per_set),
                                  c_float(sigma), c_float(inv_n0), _p(counters), _p(llr))
    assert rc == 0, rc
    return (counters, llr) if want_llr else counters


def count_bit_errors(a, b):
    a = _c(a, np.uint8).ravel()
    b = _c(b, np.uint8).ravel()
    assert a.size == b.size
    return int(lib().orc_count_bit_errors(_p(a), _p(b), c_size_t(a.size)))


def llr_abs_units(llr):
    """Counter 5 per LLR: round(min(|L|, 64) / 64 * 65535), NaN -> 0 (numpy restatement of llr_stats in sdr_oracle.cpp;
    the product is exact in float64 and never lies on a tie)."""
    x = np.abs(np.asarray(llr, dtype=np.float32)) * np.float32(0.015625)
    x = np.where(np.isnan(x), np.float32(0.0), np.minimum(x, np.float32(1.0)))
    return np.floor(x.astype(np.float64) * 65535.0 + 0.5).astype(np.int64)

