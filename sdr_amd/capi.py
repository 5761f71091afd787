"""ctypes binding of libsdr_amd.so (include/sdr_amd.h).

The library is the product; this module is plumbing.  There is no CPU fallback: if the shared
object is missing, importing any compute entry point raises, and every call on a machine
without a HIP device fails with the library's own error text.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_size_t, c_uint8, c_uint16, c_uint32, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SDR_AMD_LIB") or os.path.join(_HERE, "lib", "libsdr_amd.so")   # SDR_AMD_LIB: A/B builds while tuning

SDR_OK = 0
SDR_ERR_NOT_POW2 = -1
SDR_ERR_INVALID = -2
SDR_ERR_HIP = -3
SDR_ERR_UNSUPPORTED = -4

POW2_MESSAGE = "The sequence size must be of powers of 2"  # reference lib/inc/fft.hpp:92


class SdrError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"libsdr_amd error {code}: {message}")
        self.code = code
        self.message = message


class DemodArgs(Structure):
    _fields_ = [
        ("struct_size", c_uint32),
        ("n_fft", c_uint32),
        ("cp", c_uint32),
        ("mod", c_uint32),
        ("n_sym", c_uint64),
        ("sym_per_set", c_uint32),
        ("llr_weight0", c_float),
        ("d_rx", c_void_p),
        ("d_eq", c_void_p),
        ("d_llr_weight", c_void_p),
        ("d_llr", c_void_p),
        ("d_hard", c_void_p),
        ("d_fft", c_void_p),
        ("d_ref_hard", c_void_p),
        ("d_counters", c_void_p),
        ("flags", c_uint32),
        ("reserved", c_uint32),
    ]


class LinkArgs(Structure):
    _fields_ = [
        ("struct_size", c_uint32),
        ("n_fft", c_uint32),
        ("cp", c_uint32),
        ("mod", c_uint32),
        ("seed", c_uint64),
        ("first_symbol", c_uint64),
        ("n_sym", c_uint64),
        ("sym_per_set", c_uint32),
        ("sigma", c_float),
        ("llr_weight0", c_float),
        ("reserved", c_uint32),
        ("d_chan", c_void_p),
        ("d_rx", c_void_p),
        ("d_tx_bits", c_void_p),
        ("d_counters", c_void_p),
    ]


POINT_HARD_ONLY = 1     # sdr_sweep_point.flags: hard decisions only, counters 5 and 6 untouched
DEMOD_WEIGHT_FROM_TAPS = 1  # sdr_demod_args.flags: w_k = llr_weight0 / |G_k|^2 from the taps, no weight array
SWEEP_ZERO_COUNTERS = 1     # sdr_sweep_args.flags: the call clears the counters ahead of the launch


class SweepPoint(Structure):
    _fields_ = [
        ("mod", c_uint32),
        ("sigma", c_float),
        ("llr_weight0", c_float),
        ("flags", c_uint32),
        ("first_symbol", c_uint64),
        ("n_sym", c_uint64),
    ]


class SweepArgs(Structure):
    _fields_ = [
        ("struct_size", c_uint32),
        ("n_fft", c_uint32),
        ("cp", c_uint32),
        ("n_points", c_uint32),
        ("seed", c_uint64),
        ("points", POINTER(SweepPoint)),
        ("d_counters", c_void_p),
        ("flags", c_uint32),
        ("reserved", c_uint32),
    ]


class ChainArgs(Structure):
    _fields_ = [
        ("struct_size", c_uint32),
        ("n_fft", c_uint32),
        ("cp", c_uint32),
        ("mod", c_uint32),
        ("seed", c_uint64),
        ("first_symbol", c_uint64),
        ("n_sym", c_uint64),
        ("chunk_symbols", c_uint64),
        ("sym_per_set", c_uint32),
        ("no_llr", c_uint32),
        ("sigma", c_float),
        ("inv_n0", c_float),
        ("d_workspace", c_void_p),
        ("workspace_bytes", c_uint64),
        ("d_counters", c_void_p),
        ("d_llr_chunk", POINTER(c_void_p)),
    ]


class PipeConfig(Structure):
    _fields_ = [
        ("struct_size", c_uint32),
        ("n_fft", c_uint32),
        ("cp", c_uint32),
        ("mod", c_uint32),
        ("chunk_symbols", c_uint64),
        ("depth", c_uint32),
        ("flags", c_uint32),
        ("sym_per_set", c_uint32),
        ("llr_weight0", c_float),
        ("d_eq", c_void_p),
        ("d_llr_weight", c_void_p),
    ]


class PipeResult(Structure):
    _fields_ = [
        ("sequence", c_uint64),
        ("n_sym", c_uint64),
        ("h_llr", c_void_p),
        ("h_hard", c_void_p),
        ("d_llr", c_void_p),
        ("d_hard", c_void_p),
    ]


PIPE_LLR_TO_HOST, PIPE_HARD_TO_HOST, PIPE_WANT_LLR, PIPE_WANT_HARD = 1, 2, 4, 8

# name -> (restype, argtypes); every symbol include/sdr_amd.h and include/sdr_amd_diag.h declare
SIGNATURES = {
    "sdr_version": (c_char_p, []),
    "sdr_last_error": (c_char_p, []),
    "sdr_device_count": (c_int, [POINTER(c_int)]),
    "sdr_create": (c_int, [c_int, POINTER(c_void_p)]),
    "sdr_create_on_stream": (c_int, [c_int, c_void_p, POINTER(c_void_p)]),
    "sdr_destroy": (None, [c_void_p]),
    "sdr_synchronize": (c_int, [c_void_p]),
    "sdr_stream": (c_void_p, [c_void_p]),
    "sdr_fft2_c32": (c_int, [c_void_p, c_void_p, c_size_t, c_size_t, c_int]),
    "sdr_fft2_c64": (c_int, [c_void_p, c_void_p, c_size_t, c_size_t, c_int]),
    "sdr_fft2_host_c32": (c_int, [c_void_p, c_void_p, c_size_t, c_size_t, c_int]),
    "sdr_fft2_host_c64": (c_int, [c_void_p, c_void_p, c_size_t, c_size_t, c_int]),
    "sdr_twiddle_table_c32": (c_int, [c_size_t, c_int, c_void_p]),
    "sdr_twiddle_table_c64": (c_int, [c_size_t, c_int, c_void_p]),
    "sdr_ofdm_rx_c32": (c_int, [c_void_p, c_void_p, c_size_t, c_size_t, c_size_t, c_void_p]),
    "sdr_ofdm_rx_c64": (c_int, [c_void_p, c_void_p, c_size_t, c_size_t, c_size_t, c_void_p]),
    "sdr_ofdm_tx_c32": (c_int, [c_void_p, c_void_p, c_size_t, c_size_t, c_size_t, c_void_p]),
    "sdr_ofdm_tx_c64": (c_int, [c_void_p, c_void_p, c_size_t, c_size_t, c_size_t, c_void_p]),
    "sdr_ofdm_rx_host_c32": (c_int, [c_void_p, c_void_p, c_size_t, c_size_t, c_size_t, c_void_p]),
    "sdr_ofdm_rx_host_c64": (c_int, [c_void_p, c_void_p, c_size_t, c_size_t, c_size_t, c_void_p]),
    "sdr_ofdm_tx_host_c32": (c_int, [c_void_p, c_void_p, c_size_t, c_size_t, c_size_t, c_void_p]),
    "sdr_ofdm_tx_host_c64": (c_int, [c_void_p, c_void_p, c_size_t, c_size_t, c_size_t, c_void_p]),
    "sdr_to_constl_c32": (c_int, [c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "sdr_to_constl_c64": (c_int, [c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "sdr_from_constl_c32": (c_int, [c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "sdr_from_constl_c64": (c_int, [c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "sdr_to_constl_host_c32": (c_int, [c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "sdr_to_constl_host_c64": (c_int, [c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "sdr_from_constl_host_c32": (c_int, [c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "sdr_from_constl_host_c64": (c_int, [c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "sdr_nearest_c32": (c_int, [c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "sdr_nearest_c64": (c_int, [c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "sdr_nearest_host_c32": (c_int, [c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "sdr_nearest_host_c64": (c_int, [c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "sdr_constellation_f32": (c_int, [c_int, POINTER(c_float), POINTER(c_float), c_void_p]),
    "sdr_ofdm_demod_c32": (c_int, [c_void_p, POINTER(DemodArgs)]),
    "sdr_ofdm_demod_host_c32": (c_int, [c_void_p, POINTER(DemodArgs), c_size_t]),
    "sdr_ofdm_demod_noalloc_c32": (c_int, [c_void_p, POINTER(DemodArgs)]),
    "sdr_placement_search_c32": (c_int, [c_void_p, c_void_p, c_uint64, c_uint32, c_uint32, c_uint32, c_uint64, c_uint64, POINTER(c_uint64), POINTER(c_uint64),
                                 POINTER(c_float), POINTER(c_float)]),
    "sdr_diag_copy_shape_c32": (c_int, [c_void_p, c_void_p, c_uint32, c_uint32, c_uint32, c_uint64, c_void_p]),
    "sdr_diag_set_tuning": (c_int, [c_char_p, ctypes.c_long]),
    "sdr_zf_taps_c32": (c_int, [c_void_p, c_int, c_void_p, c_size_t, c_float, c_void_p, c_void_p]),
    "sdr_zf_taps_host_c32": (c_int, [c_void_p, c_int, c_void_p, c_size_t, c_float, c_void_p, c_void_p]),
    "sdr_soft_demap_c32": (c_int, [c_void_p, c_int, c_void_p, c_size_t, c_void_p, c_float, c_void_p]),
    "sdr_soft_demap_host_c32": (c_int, [c_void_p, c_int, c_void_p, c_size_t, c_void_p, c_float, c_void_p]),
    "sdr_link_generate_c32": (c_int, [c_void_p, POINTER(LinkArgs)]),
    "sdr_link_ber_c32": (c_int, [c_void_p, POINTER(LinkArgs)]),
    "sdr_link_sweep_c32": (c_int, [c_void_p, POINTER(SweepArgs)]),
    "sdr_link_channel_c32": (c_int, [c_void_p, c_uint32, c_uint64, c_uint64, c_size_t, c_void_p]),
    "sdr_link_chain_workspace": (c_int, [POINTER(ChainArgs), POINTER(c_uint64)]),
    "sdr_link_chain_c32": (c_int, [c_void_p, POINTER(ChainArgs)]),
    "sdr_philox4x32_10": (c_int, [POINTER(c_uint32), POINTER(c_uint32), POINTER(c_uint32)]),
    "sdr_rng_tables_f32": (c_int, [c_void_p]),
    "sdr_pipe_create": (c_int, [c_int, POINTER(PipeConfig), POINTER(c_void_p)]),
    "sdr_pipe_acquire": (c_int, [c_void_p, POINTER(c_void_p)]),
    "sdr_pipe_submit": (c_int, [c_void_p, c_uint64]),
    "sdr_pipe_collect": (c_int, [c_void_p, POINTER(PipeResult)]),
    "sdr_pipe_release": (c_int, [c_void_p]),
    "sdr_pipe_in_flight": (c_int, [c_void_p, POINTER(c_uint32)]),
    "sdr_pipe_destroy": (None, [c_void_p]),
}

_lib = None


OLDER_BUILD = False      # set by tools/abbench.py only: the library about to be loaded may lack the newer entry points


def load() -> ctypes.CDLL:
    """Loads libsdr_amd.so (built in-tree by __graft_entry__.build() / sdr_amd/csrc/Makefile)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C sdr_amd/csrc` (hipcc, gfx950). "
            "sdr_amd has no CPU fallback."
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        if OLDER_BUILD and not hasattr(lib, name):
            continue             # tools/abbench.py timing an earlier round's library against this one
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def last_error() -> str:
    return load().sdr_last_error().decode("utf-8", "replace")


def check(rc: int) -> None:
    if rc != SDR_OK:
        raise SdrError(rc, last_error())
