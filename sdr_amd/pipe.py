"""Streaming ingest over the C ABI's sdr_pipe_* (include/sdr_amd.h, "Streaming ingest").

The reference feeds its receive chain one OFDM symbol per timer tick from host memory and keeps
the outputs in host-side rings (utils::sliding_buffer, lib/inc/sliding_buffer.hpp:14-104; filled
at app/QFDemoWindow.cpp:213-229).  Here the producer writes chunks of received symbols into
pinned host buffers and the upload, the fused receive kernel and the download of the results
overlap on three streams.

    with Pipe(0, n_fft=1024, cp=256, mod=4, chunk_symbols=4096, depth=3, hard_to_host=True) as pipe:
        for hard in pipe.run(chunks):          # chunks: iterable of complex64 arrays
            ...
"""
import ctypes
from typing import Iterable, Iterator, Optional

import numpy as np

from . import capi
from .capi import PipeConfig, PipeResult, check


class Chunk:
    """Results of one submitted chunk; the arrays alias pinned memory until release()."""

    def __init__(self, sequence, n_sym, llr, hard, d_llr, d_hard):
        self.sequence, self.n_sym, self.llr, self.hard, self.d_llr, self.d_hard = sequence, n_sym, llr, hard, d_llr, d_hard


class Pipe:
    def __init__(self, device: int = 0, *, n_fft: int, cp: int, mod: int, chunk_symbols: int, depth: int = 3,
                 llr_to_host: bool = False, hard_to_host: bool = False, want_llr: bool = False, want_hard: bool = False,
                 llr_weight: float = 1.0, sym_per_set: int = 0, d_eq: int = 0, d_llr_weight: int = 0):
        self.lib = capi.load()
        self.n_fft, self.cp, self.mod, self.chunk_symbols, self.depth = n_fft, cp, mod, chunk_symbols, depth
        c = PipeConfig()
        c.struct_size = ctypes.sizeof(PipeConfig)
        c.n_fft, c.cp, c.mod, c.chunk_symbols, c.depth = n_fft, cp, mod, chunk_symbols, depth
        c.flags = ((capi.PIPE_LLR_TO_HOST if llr_to_host else 0) | (capi.PIPE_HARD_TO_HOST if hard_to_host else 0)
                   | (capi.PIPE_WANT_LLR if want_llr else 0) | (capi.PIPE_WANT_HARD if want_hard else 0))
        c.sym_per_set, c.llr_weight0 = sym_per_set, llr_weight
        c.d_eq, c.d_llr_weight = d_eq or None, d_llr_weight or None
        h = ctypes.c_void_p()
        check(self.lib.sdr_pipe_create(device, ctypes.byref(c), ctypes.byref(h)))
        self.handle = h

    # ---- producer ----
    def acquire(self) -> np.ndarray:
        """Pinned input buffer of the next slot as a complex64 array (chunk_symbols, n_fft + cp)."""
        p = ctypes.c_void_p()
        check(self.lib.sdr_pipe_acquire(self.handle, ctypes.byref(p)))
        n = self.chunk_symbols * (self.n_fft + self.cp)
        buf = (ctypes.c_float * (2 * n)).from_address(p.value)
        return np.frombuffer(buf, dtype=np.complex64).reshape(self.chunk_symbols, self.n_fft + self.cp)

    def submit(self, n_sym: int) -> None:
        check(self.lib.sdr_pipe_submit(self.handle, n_sym))

    # ---- consumer ----
    def collect(self) -> Chunk:
        r = PipeResult()
        check(self.lib.sdr_pipe_collect(self.handle, ctypes.byref(r)))
        llr = hard = None
        if r.h_llr:
            n = r.n_sym * self.n_fft * self.mod
            llr = np.frombuffer((ctypes.c_float * n).from_address(r.h_llr), dtype=np.float32).reshape(r.n_sym, self.n_fft, self.mod)
        if r.h_hard:
            n = r.n_sym * self.n_fft * self.mod // 8
            hard = np.frombuffer((ctypes.c_uint8 * n).from_address(r.h_hard), dtype=np.uint8)
        return Chunk(r.sequence, r.n_sym, llr, hard, r.d_llr, r.d_hard)

    def release(self) -> None:
        check(self.lib.sdr_pipe_release(self.handle))

    def in_flight(self) -> int:
        n = ctypes.c_uint32()
        check(self.lib.sdr_pipe_in_flight(self.handle, ctypes.byref(n)))
        return n.value

    def run(self, chunks: Iterable[np.ndarray]) -> Iterator[Chunk]:
        """Feeds every chunk (complex64, whole symbols) and yields results in order, keeping up to
        `depth` chunks in flight.  A yielded Chunk is released when the generator is advanced."""
        def drain_one():
            c = self.collect()
            try:
                yield c
            finally:
                self.release()
        for x in chunks:
            x = np.ascontiguousarray(x, dtype=np.complex64).reshape(-1, self.n_fft + self.cp)
            if self.in_flight() == self.depth:
                yield from drain_one()
            buf = self.acquire()
            buf[: x.shape[0]] = x
            self.submit(x.shape[0])
        while self.in_flight():
            yield from drain_one()

    def close(self) -> None:
        if getattr(self, "handle", None):
            self.lib.sdr_pipe_destroy(self.handle)
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
