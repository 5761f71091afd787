"""Host-side mirror of the reference's fft / ofdm / modulation interface on device tensors.

Names and argument meaning follow the reference headers (lib/inc/fft.hpp, ofdm.hpp,
modulation.hpp); each method calls the C ABI of libsdr_amd.so with torch tensors used only as
device memory.  Complex tensors are torch.complex64 / complex128, contiguous, on the GPU.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import Optional

import torch

from . import capi
from .capi import DEMOD_WEIGHT_FROM_TAPS, POINT_HARD_ONLY, SWEEP_ZERO_COUNTERS, ChainArgs, DemodArgs, LinkArgs, SweepArgs, SweepPoint, check

MODS = {"bpsk": 1, "qpsk": 2, "16qam": 4, "64qam": 6, "256qam": 8, "1024qam": 10}

COUNTER_NAMES = ("bit_errors", "bits", "symbol_errors", "symbols", "ofdm_symbols", "llr_abs_q10", "llr_sign_errors", "reserved")


def mod_bits(mod) -> int:
    if isinstance(mod, str):
        return MODS[mod.lower()]
    return int(mod)


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _need(t: torch.Tensor, dtype, name: str, numel: Optional[int] = None):
    if not t.is_cuda:
        raise ValueError(f"{name} must be a GPU tensor")
    if t.dtype != dtype:
        raise ValueError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    if numel is not None and t.numel() != numel:
        raise ValueError(f"{name} must have {numel} elements, got {t.numel()}")


@dataclass
class DemodResult:
    llr: Optional[torch.Tensor]
    hard: Optional[torch.Tensor]
    fft: Optional[torch.Tensor]


class Context:
    """One device + one stream (sdr_ctx).  Enqueues on torch's current stream of `device`."""

    def __init__(self, device: int = 0):
        self.lib = capi.load()
        self.device = int(device)
        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device visible; sdr_amd has no CPU fallback")
        torch.cuda.set_device(self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        h = ctypes.c_void_p()
        check(self.lib.sdr_create_on_stream(self.device, ctypes.c_void_p(stream), ctypes.byref(h)))
        self._h = h
        self.torch_device = torch.device("cuda", self.device)

    def close(self):
        if getattr(self, "_h", None):
            self.lib.sdr_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        check(self.lib.sdr_synchronize(self._h))

    # ---- fft::fft2 / fft::ifft2 (lib/inc/fft.hpp:176-205), in place, batched over rows ----
    def fft2_(self, x: torch.Tensor, inverse: bool = False) -> torch.Tensor:
        if x.dtype not in (torch.complex64, torch.complex128):
            raise ValueError("fft2_ needs a complex tensor")
        _need(x, x.dtype, "x")
        n = x.shape[-1] if x.dim() else 1
        batch = x.numel() // n if n else 0
        fn = self.lib.sdr_fft2_c32 if x.dtype == torch.complex64 else self.lib.sdr_fft2_c64
        check(fn(self._h, _ptr(x), n, batch, int(inverse)))
        return x

    def ifft2_(self, x: torch.Tensor) -> torch.Tensor:
        return self.fft2_(x, inverse=True)

    # ---- ofdm::rx / ofdm::tx (lib/inc/ofdm.hpp:13-54), batched ----
    def ofdm_rx(self, rx: torch.Tensor, n_fft: int, cp: int) -> torch.Tensor:
        if rx.dtype not in (torch.complex64, torch.complex128):
            raise ValueError("ofdm_rx needs a complex tensor")
        _need(rx, rx.dtype, "rx")
        n_sym = rx.numel() // (n_fft + cp) if (n_fft + cp) else 0
        if rx.numel() != n_sym * (n_fft + cp):
            raise ValueError("rx length is not a multiple of n_fft+cp")
        out = torch.empty((n_sym, n_fft), dtype=rx.dtype, device=rx.device)
        fn = self.lib.sdr_ofdm_rx_c32 if rx.dtype == torch.complex64 else self.lib.sdr_ofdm_rx_c64
        check(fn(self._h, _ptr(rx), n_fft, cp, n_sym, _ptr(out)))
        return out

    def ofdm_tx(self, freq: torch.Tensor, n_fft: int, cp: int) -> torch.Tensor:
        if freq.dtype not in (torch.complex64, torch.complex128):
            raise ValueError("ofdm_tx needs a complex tensor")
        _need(freq, freq.dtype, "freq")
        n_sym = freq.numel() // n_fft if n_fft else 0
        if freq.numel() != n_sym * n_fft:
            raise ValueError("freq length is not a multiple of n_fft")
        out = torch.empty((n_sym, n_fft + cp), dtype=freq.dtype, device=freq.device)
        fn = self.lib.sdr_ofdm_tx_c32 if freq.dtype == torch.complex64 else self.lib.sdr_ofdm_tx_c64
        check(fn(self._h, _ptr(freq), n_fft, cp, n_sym, _ptr(out)))
        return out

    # ---- modulation::to_constl / from_constl / nearest (lib/inc/modulation.hpp:49-112) ----
    def to_constl(self, data: torch.Tensor, mod, n_sym: Optional[int] = None, dtype=torch.complex64) -> torch.Tensor:
        bps = mod_bits(mod)
        _need(data, torch.uint8, "data")
        if n_sym is None:
            n_sym = data.numel() * 8 // bps
        out = torch.empty(n_sym, dtype=dtype, device=data.device)
        fn = self.lib.sdr_to_constl_c32 if dtype == torch.complex64 else self.lib.sdr_to_constl_c64
        check(fn(self._h, bps, _ptr(data), n_sym, _ptr(out)))
        return out

    def from_constl(self, sym: torch.Tensor, mod) -> torch.Tensor:
        bps = mod_bits(mod)
        _need(sym, sym.dtype, "sym")
        n_sym = sym.numel()
        out = torch.empty(n_sym * bps // 8, dtype=torch.uint8, device=sym.device)
        fn = self.lib.sdr_from_constl_c32 if sym.dtype == torch.complex64 else self.lib.sdr_from_constl_c64
        check(fn(self._h, bps, _ptr(sym), n_sym, _ptr(out)))
        return out

    def nearest(self, sym: torch.Tensor, mod) -> torch.Tensor:
        bps = mod_bits(mod)
        if sym.dtype not in (torch.complex64, torch.complex128):
            raise ValueError("nearest needs a complex tensor")
        _need(sym, sym.dtype, "sym")
        out = torch.empty(sym.numel(), dtype=torch.int16, device=sym.device)
        fn = self.lib.sdr_nearest_c32 if sym.dtype == torch.complex64 else self.lib.sdr_nearest_c64
        check(fn(self._h, bps, _ptr(sym), sym.numel(), _ptr(out)))
        return out

    def soft_demap(self, sym: torch.Tensor, mod, weight=1.0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        bps = mod_bits(mod)
        _need(sym, torch.complex64, "sym")
        if out is None:
            out = torch.empty((sym.numel(), bps), dtype=torch.float32, device=sym.device)
        else:
            _need(out, torch.float32, "out", sym.numel() * bps)
        wt = weight if isinstance(weight, torch.Tensor) else None
        w0 = 0.0 if wt is not None else float(weight)
        if wt is not None:
            _need(wt, torch.float32, "weight", sym.numel())
        check(self.lib.sdr_soft_demap_c32(self._h, bps, _ptr(sym), sym.numel(), _ptr(wt), w0, _ptr(out)))
        return out

    def zf_taps(self, chan: torch.Tensor, mod, inv_n0: float):
        bps = mod_bits(mod)
        _need(chan, torch.complex64, "chan")
        taps = torch.empty_like(chan)
        wgt = torch.empty(chan.shape, dtype=torch.float32, device=chan.device)
        check(self.lib.sdr_zf_taps_c32(self._h, bps, _ptr(chan), chan.numel(), float(inv_n0), _ptr(taps), _ptr(wgt)))
        return taps, wgt

    # ---- fused receive path: CP removal -> FFT -> equalise -> LLR (+ hard bits, counters) ----
    def demod(self, rx: torch.Tensor, n_fft: int, cp: int, mod, *, eq: Optional[torch.Tensor] = None,
              llr_weight=1.0, sym_per_set: int = 0, want_llr: bool = True, want_hard: bool = False,
              want_fft: bool = False, ref_hard: Optional[torch.Tensor] = None, counters: Optional[torch.Tensor] = None,
              out_llr: Optional[torch.Tensor] = None, out_hard: Optional[torch.Tensor] = None,
              out_fft: Optional[torch.Tensor] = None, noalloc: bool = False, weight_from_taps: bool = False) -> DemodResult:
        """weight_from_taps=True (taps given, llr_weight a scalar): SDR_DEMOD_WEIGHT_FROM_TAPS, w_k = llr_weight / |G_k|^2
        formed in the kernel.  noalloc=True calls sdr_ofdm_demod_noalloc_c32 (static device scratch, no hipMalloc in the
        library; pass every out_* tensor so that torch allocates nothing either)."""
        bps = mod_bits(mod)
        _need(rx, torch.complex64, "rx")
        n_sym = rx.numel() // (n_fft + cp)
        if rx.numel() != n_sym * (n_fft + cp):
            raise ValueError("rx length is not a multiple of n_fft+cp")
        dev = rx.device
        llr = out_llr if out_llr is not None else (torch.empty((n_sym, n_fft, bps), dtype=torch.float32, device=dev) if want_llr else None)
        hard = out_hard if out_hard is not None else (torch.empty(n_sym * n_fft * bps // 8, dtype=torch.uint8, device=dev) if want_hard else None)
        fft = out_fft if out_fft is not None else (torch.empty((n_sym, n_fft), dtype=torch.complex64, device=dev) if want_fft else None)
        wt = llr_weight if isinstance(llr_weight, torch.Tensor) else None
        a = DemodArgs()
        a.struct_size = getattr(self, "demod_struct_size", None) or ctypes.sizeof(DemodArgs)      # (tools/abbench.py: an older build's layout)
        a.n_fft, a.cp, a.mod, a.n_sym = n_fft, cp, bps, n_sym
        a.sym_per_set = sym_per_set
        a.llr_weight0 = 0.0 if wt is not None else float(llr_weight)
        a.d_rx = rx.data_ptr()
        # the kernel indexes tap set sym // sym_per_set (set 0 alone when sym_per_set is 0)
        need = n_fft * ((n_sym + sym_per_set - 1) // sym_per_set if sym_per_set else 1)
        if eq is not None:
            _need(eq, torch.complex64, "eq")
            if eq.numel() < need:
                raise ValueError(f"eq must hold at least {need} taps ({need // max(n_fft, 1)} sets of {n_fft}), got {eq.numel()}")
            a.d_eq = eq.data_ptr()
        if wt is not None:
            _need(wt, torch.float32, "llr_weight")
            if wt.numel() < need:
                raise ValueError(f"llr_weight must hold at least {need} weights ({need // max(n_fft, 1)} sets of {n_fft}), got {wt.numel()}")
            a.d_llr_weight = wt.data_ptr()
        if llr is not None:
            _need(llr, torch.float32, "llr", n_sym * n_fft * bps)
            a.d_llr = llr.data_ptr()
        if hard is not None:
            _need(hard, torch.uint8, "hard", n_sym * n_fft * bps // 8)
            a.d_hard = hard.data_ptr()
        if fft is not None:
            _need(fft, torch.complex64, "fft", n_sym * n_fft)
            a.d_fft = fft.data_ptr()
        if ref_hard is not None:
            _need(ref_hard, torch.uint8, "ref_hard", n_sym * n_fft * bps // 8)
            _need(counters, torch.int64, "counters", 8)
            a.d_ref_hard = ref_hard.data_ptr()
            a.d_counters = counters.data_ptr()
        if weight_from_taps:
            a.flags = DEMOD_WEIGHT_FROM_TAPS
        fn = self.lib.sdr_ofdm_demod_noalloc_c32 if noalloc else self.lib.sdr_ofdm_demod_c32
        check(fn(self._h, ctypes.byref(a)))
        return DemodResult(llr, hard, fft)

    # ---- where the buffers lie (sdr_placement_search_c32; no counterpart in the reference) ----
    def placement_search(self, arena: torch.Tensor, n_fft: int, cp: int, mod, n_sym: int, step_bytes: int = 0):
        """Fastest (in_offset, out_offset) in bytes for demod()'s sample and LLR buffers inside `arena` (a contiguous uint8 GPU
        tensor), and the traffic model's rate in GB/s at the best and the worst placement tried.  Overwrites parts of the arena."""
        _need(arena, torch.uint8, "arena")
        i, o = ctypes.c_uint64(), ctypes.c_uint64()
        best, worst = ctypes.c_float(), ctypes.c_float()
        check(self.lib.sdr_placement_search_c32(self._h, _ptr(arena), arena.numel(), n_fft, cp, mod_bits(mod), n_sym, step_bytes,
                                                ctypes.byref(i), ctypes.byref(o), ctypes.byref(best), ctypes.byref(worst)))
        return i.value, o.value, best.value, worst.value

    # ---- device frame generator / AWGN / Monte-Carlo link ----
    def copy_shape(self, rx: torch.Tensor, n_fft: int, cp: int, mod, out_llr: torch.Tensor) -> None:
        """Diagnostic: the bytes of demod(LLR only) with no arithmetic (sdr_diag_copy_shape_c32)."""
        bps = mod_bits(mod)
        n_sym = rx.numel() // (n_fft + cp)
        _need(rx, torch.complex64, "rx", n_sym * (n_fft + cp))
        _need(out_llr, torch.float32, "llr", n_sym * n_fft * bps)
        check(self.lib.sdr_diag_copy_shape_c32(self._h, _ptr(rx), n_fft, cp, bps, n_sym, _ptr(out_llr)))

    def _link_args(self, n_fft, cp, mod, seed, first_symbol, n_sym, sigma, sym_per_set=0, llr_weight=1.0) -> LinkArgs:
        a = LinkArgs()
        a.struct_size = ctypes.sizeof(LinkArgs)
        a.n_fft, a.cp, a.mod = n_fft, cp, mod_bits(mod)
        a.seed, a.first_symbol, a.n_sym = seed, first_symbol, n_sym
        a.sym_per_set = sym_per_set
        a.sigma = float(sigma)
        a.llr_weight0 = float(llr_weight)
        return a

    def link_generate(self, n_fft: int, cp: int, mod, seed: int, first_symbol: int, n_sym: int, sigma: float, *,
                      chan: Optional[torch.Tensor] = None, sym_per_set: int = 0, want_bits: bool = True,
                      out_rx: Optional[torch.Tensor] = None):
        bps = mod_bits(mod)
        rx = out_rx if out_rx is not None else torch.empty((n_sym, n_fft + cp), dtype=torch.complex64, device=self.torch_device)
        _need(rx, torch.complex64, "rx", n_sym * (n_fft + cp))
        bits = torch.empty(n_sym * n_fft * bps // 8, dtype=torch.uint8, device=self.torch_device) if want_bits else None
        a = self._link_args(n_fft, cp, mod, seed, first_symbol, n_sym, sigma, sym_per_set)
        if chan is not None:
            _need(chan, torch.complex64, "chan")
            a.d_chan = chan.data_ptr()
        a.d_rx = rx.data_ptr()
        if bits is not None:
            a.d_tx_bits = bits.data_ptr()
        check(self.lib.sdr_link_generate_c32(self._h, ctypes.byref(a)))
        return rx, bits

    def link_ber(self, n_fft: int, cp: int, mod, seed: int, first_symbol: int, n_sym: int, sigma: float,
                 llr_weight: float = 1.0, counters: Optional[torch.Tensor] = None) -> torch.Tensor:
        if counters is None:
            counters = torch.zeros(8, dtype=torch.int64, device=self.torch_device)
        _need(counters, torch.int64, "counters", 8)
        a = self._link_args(n_fft, cp, mod, seed, first_symbol, n_sym, sigma, 0, llr_weight)
        a.d_counters = counters.data_ptr()
        check(self.lib.sdr_link_ber_c32(self._h, ctypes.byref(a)))
        return counters

    def link_sweep(self, n_fft: int, cp: int, seed: int, points, counters: Optional[torch.Tensor] = None, hard_only: bool = False,
                   zero_counters: bool = False) -> torch.Tensor:
        """sdr_link_sweep_c32: every (modulation, noise level) point of a sweep in ONE launch.
        points = sequence of (mod, sigma, llr_weight0, first_symbol, n_sym[, flags]); counters = int64
        tensor (n_points, 8) on the device, accumulated into (created zeroed when None).
        hard_only: SDR_POINT_HARD_ONLY on every point (bit / symbol errors from hard decisions, no LLRs:
        counters 5 and 6 are left alone).  zero_counters: SDR_SWEEP_ZERO_COUNTERS, the library clears `counters` on its
        stream ahead of the launch instead of accumulating into them."""
        n = len(points)
        if counters is None:
            counters = torch.zeros((n, 8), dtype=torch.int64, device=self.torch_device)
        _need(counters, torch.int64, "counters", n * 8)
        table = (SweepPoint * max(n, 1))()
        for i, pt in enumerate(points):
            mod, sigma, w0, first, count = pt[:5]
            table[i].mod, table[i].sigma, table[i].llr_weight0 = mod_bits(mod), float(sigma), float(w0)
            table[i].first_symbol, table[i].n_sym = int(first), int(count)
            table[i].flags = (int(pt[5]) if len(pt) > 5 else 0) | (POINT_HARD_ONLY if hard_only else 0)
        a = SweepArgs()
        a.struct_size = ctypes.sizeof(SweepArgs)
        a.n_fft, a.cp, a.n_points, a.seed = n_fft, cp, n, seed
        a.points = ctypes.cast(table, ctypes.POINTER(SweepPoint))
        a.d_counters = counters.data_ptr()
        a.flags = SWEEP_ZERO_COUNTERS if zero_counters else 0
        check(self.lib.sdr_link_sweep_c32(self._h, ctypes.byref(a)))
        return counters

    def link_channel(self, n_fft: int, seed: int, first_set: int, n_sets: int) -> torch.Tensor:
        """sdr_link_channel_c32: unit-variance complex Gaussian channel realisations, (n_sets, n_fft)."""
        chan = torch.empty((n_sets, n_fft), dtype=torch.complex64, device=self.torch_device)
        check(self.lib.sdr_link_channel_c32(self._h, n_fft, seed, first_set, n_sets, _ptr(chan)))
        return chan

    def _chain_args(self, n_fft, cp, mod, seed, first_symbol, n_sym, chunk_symbols, sym_per_set, sigma, inv_n0, want_llr) -> ChainArgs:
        a = ChainArgs()
        a.struct_size = ctypes.sizeof(ChainArgs)
        a.n_fft, a.cp, a.mod = n_fft, cp, mod_bits(mod)
        a.seed, a.first_symbol, a.n_sym, a.chunk_symbols = seed, first_symbol, n_sym, chunk_symbols
        a.sym_per_set, a.no_llr = sym_per_set, 0 if want_llr else 1
        a.sigma, a.inv_n0 = float(sigma), float(inv_n0)
        return a

    def link_chain_workspace(self, n_fft, cp, mod, n_sym, chunk_symbols, sym_per_set, want_llr=True) -> int:
        a = self._chain_args(n_fft, cp, mod, 0, 0, n_sym, chunk_symbols, sym_per_set, 0.0, 0.0, want_llr)
        n = ctypes.c_uint64()
        check(self.lib.sdr_link_chain_workspace(ctypes.byref(a), ctypes.byref(n)))
        return int(n.value)

    def link_chain(self, n_fft: int, cp: int, mod, seed: int, first_symbol: int, n_sym: int, chunk_symbols: int, sym_per_set: int,
                   sigma: float, inv_n0: float, *, workspace: Optional[torch.Tensor] = None, counters: Optional[torch.Tensor] = None, want_llr: bool = True):
        """sdr_link_chain_c32: channel -> taps -> generate -> equalised demodulation -> counters over
        n_sym symbols in resident chunks that reuse `workspace` (uint8 tensor; allocated when None).
        Returns (counters, llr view of the LAST chunk or None, workspace)."""
        need = self.link_chain_workspace(n_fft, cp, mod, n_sym, chunk_symbols, sym_per_set, want_llr)
        if workspace is None:
            workspace = torch.empty(need, dtype=torch.uint8, device=self.torch_device)
        _need(workspace, torch.uint8, "workspace")
        if counters is None:
            counters = torch.zeros(8, dtype=torch.int64, device=self.torch_device)
        _need(counters, torch.int64, "counters", 8)
        a = self._chain_args(n_fft, cp, mod, seed, first_symbol, n_sym, chunk_symbols, sym_per_set, sigma, inv_n0, want_llr)
        a.d_workspace, a.workspace_bytes, a.d_counters = workspace.data_ptr(), workspace.numel(), counters.data_ptr()
        where = ctypes.c_void_p()
        a.d_llr_chunk = ctypes.pointer(where)
        check(self.lib.sdr_link_chain_c32(self._h, ctypes.byref(a)))
        llr = None
        if want_llr and where.value:
            bps = mod_bits(mod)
            chunk = min(chunk_symbols, -(-n_sym // sym_per_set) * sym_per_set)
            last = n_sym - ((n_sym - 1) // chunk) * chunk
            off = where.value - workspace.data_ptr()
            llr = workspace[off: off + last * n_fft * bps * 4].view(torch.float32).view(last, n_fft, bps)
        return counters, llr, workspace

