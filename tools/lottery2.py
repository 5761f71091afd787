#!/usr/bin/env python3
"""Which buffer's placement moves the equalised kernel's rate?  FFT-1024 / 64-QAM / 1 Mi symbols, taps per
14 symbols.  Phase A: samples and LLRs stay, the taps + weights move (fresh allocations, junk in between).
Phase B: taps stay, the LLR buffer moves.  Phase C: taps stay, the sample buffer moves."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sdr_amd.ops import Context
ctx = Context(0)
n, cp, bps, n_sym, per = 1024, 256, 6, 1 << 20, 14
g = torch.Generator(device="cuda").manual_seed(1)
n_sets = (n_sym + per - 1) // per
B = (8 * n + 4 * n * bps) * n_sym


def mk_rx():
    return torch.randn((n_sym, (n + cp) * 2), dtype=torch.float32, device="cuda", generator=g).mul_((2.0 * n) ** -0.5).view(torch.complex64)


def mk_taps():
    eq = torch.randn((n_sets, n * 2), dtype=torch.float32, device="cuda", generator=g).view(torch.complex64)
    w = torch.rand((n_sets, n), dtype=torch.float32, device="cuda", generator=g)
    return eq, w


def rate(rx, eq, w, llr):
    for _ in range(4):
        ctx.demod(rx, n, cp, bps, eq=eq, llr_weight=w, sym_per_set=per, out_llr=llr)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(8):
        ctx.demod(rx, n, cp, bps, eq=eq, llr_weight=w, sym_per_set=per, out_llr=llr)
    e1.record()
    torch.cuda.synchronize()
    return B / (e0.elapsed_time(e1) / 8 / 1e3) / 8e12 * 100


rx = mk_rx()
llr = torch.empty((n_sym, n, bps), dtype=torch.float32, device="cuda")
eq, w = mk_taps()
junk = []
out = []
for i in range(6):
    out.append(f"{rate(rx, eq, w, llr):.1f} (eq@{eq.data_ptr() >> 21 & 0xfff:03x} w@{w.data_ptr() >> 21 & 0xfff:03x})")
    junk.append(torch.empty(((37 + 11 * i) << 20,), dtype=torch.uint8, device="cuda"))
    eq, w = mk_taps()
print("A taps move:", "  ".join(out), flush=True)
out = []
for i in range(5):
    out.append(f"{rate(rx, eq, w, llr):.1f} (llr@{llr.data_ptr() >> 21:x})")
    junk.append(torch.empty(((53 + 17 * i) << 20,), dtype=torch.uint8, device="cuda"))
    llr = torch.empty((n_sym, n, bps), dtype=torch.float32, device="cuda")
print("B llr moves:", "  ".join(out), flush=True)
out = []
for i in range(5):
    out.append(f"{rate(rx, eq, w, llr):.1f} (rx@{rx.data_ptr() >> 21:x})")
    junk.append(torch.empty(((29 + 13 * i) << 20,), dtype=torch.uint8, device="cuda"))
    rx = mk_rx()
print("C rx moves: ", "  ".join(out), flush=True)
out = []
for i in range(5):
    out.append(f"{rate(rx, None, 1.5, llr):.1f}")
    junk.append(torch.empty(((29 + 13 * i) << 20,), dtype=torch.uint8, device="cuda"))
    llr = torch.empty((n_sym, n, bps), dtype=torch.float32, device="cuda")
print("D no taps, llr moves:", "  ".join(out), flush=True)
