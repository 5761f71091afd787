"""One arena, buffers placed inside it where the receive kernel runs fastest (sdr_placement_search_c32).

On MI355X the rate of sdr_ofdm_demod_c32 depends on where its sample and LLR buffers lie in device memory
(include/sdr_amd.h): 0.74 / 0.77 / 0.80-0.83 of the HBM peak for the same kernel and data.  A
plain allocation leaves that to chance; an application that owns one large arena can choose.

    arena = Arena(ctx, 192 << 30)
    rx, llr = arena.place_demod(n_fft, cp, mod, n_sym)      # views of the arena, fastest placement found
    rx.copy_(samples); ctx.demod(rx, n_fft, cp, mod, out_llr=llr)
    taps = arena.alloc((n_sets, n_fft), torch.complex64)    # anything else: first free gap
    arena.reset()                                           # all views are dead from here on
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch

from .ops import Context, mod_bits

GRAIN = 2 << 20          # offsets handed out are multiples of 2 MiB, like the search's


def _bytes(shape, dtype) -> int:
    n = 1
    for s in shape:
        n *= int(s)
    return n * torch.empty((), dtype=dtype).element_size()


class Arena:
    """A contiguous block of device memory, the views into it and a first-fit record of what is taken."""

    def __init__(self, ctx: Context, n_bytes: int, step_bytes: int = 0):
        self.ctx = ctx
        self.step_bytes = int(step_bytes)
        self.mem = torch.empty((int(n_bytes) // GRAIN * GRAIN,), dtype=torch.uint8, device=ctx.torch_device)
        self.taken: List[Tuple[int, int]] = []          # (offset, end), sorted
        self.last: Optional[dict] = None

    @property
    def n_bytes(self) -> int:
        return self.mem.numel()

    def reset(self) -> None:
        self.taken = []

    def close(self) -> None:
        """Gives the memory back to torch's allocator; every view handed out must be dead."""
        self.taken = []
        self.mem = None

    def view(self, offset: int, shape, dtype) -> torch.Tensor:
        n_bytes = _bytes(shape, dtype)
        if offset < 0 or offset + n_bytes > self.n_bytes:
            raise ValueError(f"view of {n_bytes} bytes at {offset} leaves the arena of {self.n_bytes}")
        return self.mem[offset:offset + n_bytes].view(dtype).view(*shape)

    def take(self, offset: int, shape, dtype) -> torch.Tensor:
        """The view at `offset`, recorded as taken; ValueError if it overlaps what is taken already."""
        end = offset + _bytes(shape, dtype)
        for a, b in self.taken:
            if offset < b and a < end:
                raise ValueError(f"[{offset}, {end}) overlaps [{a}, {b})")
        t = self.view(offset, shape, dtype)
        self.taken.append((offset, (end + GRAIN - 1) // GRAIN * GRAIN))
        self.taken.sort()
        return t

    def alloc(self, shape, dtype) -> torch.Tensor:
        """First free gap that holds the tensor (2 MiB grain); MemoryError if none does."""
        need = _bytes(shape, dtype)
        at = 0
        for a, b in self.taken + [(self.n_bytes, self.n_bytes)]:
            if a - at >= need:
                return self.take(at, shape, dtype)
            at = max(at, b)
        raise MemoryError(f"no gap of {need} bytes left in the arena of {self.n_bytes}")

    def search(self, n_fft: int, cp: int, mod, n_sym: int) -> dict:
        i, o, best, worst = self.ctx.placement_search(self.mem, n_fft, cp, mod_bits(mod), n_sym, self.step_bytes)
        self.last = {"in_offset": i, "out_offset": o, "model_best_gbs": best, "model_worst_gbs": worst}
        return self.last

    def place_demod(self, n_fft: int, cp: int, mod, n_sym: int):
        """(rx, llr): complex64 [n_sym, n_fft + cp] and float32 [n_sym, n_fft, bps] views at the fastest placement found,
        recorded as taken.  Call it on an arena with nothing else taken (the search writes where it tries outputs), and
        fill rx afterwards."""
        if self.taken:
            raise ValueError("place_demod on an arena with live views: the search would overwrite them; reset() first")
        p = self.search(n_fft, cp, mod, n_sym)
        return (self.take(p["in_offset"], (n_sym, n_fft + cp), torch.complex64),
                self.take(p["out_offset"], (n_sym, n_fft, mod_bits(mod)), torch.float32))
