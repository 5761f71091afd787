#!/usr/bin/env python3
"""Find a FAST draw of the equalised workload (all buffers re-allocated until frac > 0.71), then move one
buffer at a time to a fresh allocation and time again: whose physical placement carries the effect?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sdr_amd.ops import Context
ctx = Context(0)
n_fft, cp, bps, n_sym, ebn0, with_eq = bench.WORKLOADS["fft1024_64qam_eq"]
B = bench.algorithmic_bytes_per_symbol(n_fft, bps) * n_sym


def rate(rx, eq, wgt, llr, per):
    for _ in range(4):
        ctx.demod(rx, n_fft, cp, bps, eq=eq, llr_weight=wgt, sym_per_set=per, out_llr=llr)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(8):
        ctx.demod(rx, n_fft, cp, bps, eq=eq, llr_weight=wgt, sym_per_set=per, out_llr=llr)
    e1.record()
    torch.cuda.synchronize()
    return B / (e0.elapsed_time(e1) / 8 / 1e3) / 8e12


for attempt in range(12):
    rx, eq, wgt, w0, per = bench.make_inputs(ctx, torch, n_fft, cp, bps, n_sym, ebn0, True, seed=2, first_symbol=0)
    llr = torch.zeros((n_sym, n_fft, bps), dtype=torch.float32, device="cuda")
    f = rate(rx, eq, wgt, llr, per)
    print(f"draw {attempt}: {f:.4f}", flush=True)
    if f > 0.71:
        break
    del rx, eq, wgt, llr
    torch.cuda.empty_cache()
else:
    sys.exit(0)
print(f"again: {rate(rx, eq, wgt, llr, per):.4f}")
eq2, wgt2 = eq.clone(), wgt.clone()
print(f"taps+weights moved: {rate(rx, eq2, wgt2, llr, per):.4f}   back: {rate(rx, eq, wgt, llr, per):.4f}")
print(f"only taps moved: {rate(rx, eq2, wgt, llr, per):.4f}   only weights moved: {rate(rx, eq, wgt2, llr, per):.4f}")
llr2 = torch.zeros_like(llr)
print(f"llr moved: {rate(rx, eq, wgt, llr2, per):.4f}   back: {rate(rx, eq, wgt, llr, per):.4f}")
del llr2
rx2 = rx.clone()
print(f"rx moved: {rate(rx2, eq, wgt, llr, per):.4f}   back: {rate(rx, eq, wgt, llr, per):.4f}")
print(f"no taps at all on these buffers: {B and 0 or 0}", end="")
for _ in range(4):
    ctx.demod(rx, n_fft, cp, bps, llr_weight=w0, out_llr=llr)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(8):
    ctx.demod(rx, n_fft, cp, bps, llr_weight=w0, out_llr=llr)
e1.record(); torch.cuda.synchronize()
print(f" {B / (e0.elapsed_time(e1) / 8 / 1e3) / 8e12:.4f}")
