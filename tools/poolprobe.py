#!/usr/bin/env python3
"""Finer cuts through tools/poolmap.py's map: where exactly the fast zone of the LLR buffer begins and ends, whether the
sample buffer has one too, and whether the tap-reading kernel sees the same map."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sdr_amd.ops import Context

ctx = Context(0)
n, cp, per = 1024, 256, 14
GiB = 1 << 30
POOL = 230
pool = torch.empty((POOL * GiB,), dtype=torch.uint8, device="cuda")
f32 = pool.view(torch.float32)
for i in range(0, f32.numel(), 1 << 30):
    f32[i:i + (1 << 30)].uniform_(-0.05, 0.05)
g = torch.Generator(device="cuda").manual_seed(1)


def rate(rx_off, llr_off, bps=6, n_sym=1 << 20, eq=None, wgt=None, steps=6):
    rx_bytes, llr_bytes = n_sym * (n + cp) * 8, n_sym * n * bps * 4
    rxc = pool[rx_off:rx_off + rx_bytes].view(torch.float32).view(n_sym, (n + cp) * 2).view(torch.complex64)
    llr = pool[llr_off:llr_off + llr_bytes].view(torch.float32).view(n_sym, n, bps)
    kw = dict(eq=eq, llr_weight=wgt, sym_per_set=per) if eq is not None else dict(llr_weight=1.5)
    for _ in range(3):
        ctx.demod(rxc, n, cp, bps, out_llr=llr, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        ctx.demod(rxc, n, cp, bps, out_llr=llr, **kw)
    e1.record()
    torch.cuda.synchronize()
    return (8 * n + 4 * n * bps) * n_sym / (e0.elapsed_time(e1) / steps * 1e-3) / 8e12


print("(a) rx at 0, llr start 26 ... 76 GiB in steps of 2:", " ".join(f"{c}:{100 * rate(0, c * GiB):.0f}" for c in range(26, 78, 2)), flush=True)
print("(a2) rx at 200, llr start 100 ... 136 GiB in steps of 2:", " ".join(f"{c}:{100 * rate(200 * GiB, c * GiB):.0f}" for c in range(100, 138, 2)), flush=True)
print("(b) llr at 112 (fast), rx start 40 ... 70 in steps of 2:", " ".join(f"{r}:{100 * rate(r * GiB, 112 * GiB):.0f}" for r in range(40, 72, 2)), flush=True)
print("(b2) llr at 80 (inside one region), rx start 40 ... 66 in steps of 2:", " ".join(f"{r}:{100 * rate(r * GiB, 80 * GiB):.0f}" for r in range(40, 68, 2)), flush=True)
n_sets = ((1 << 20) + per - 1) // per
eq = torch.randn((n_sets, n * 2), dtype=torch.float32, device="cuda", generator=g).view(torch.complex64)
wgt = torch.rand((n_sets, n), dtype=torch.float32, device="cuda", generator=g)
print("(c) taps + weights, rx at 0, llr start 16 ... 200 in steps of 8:", " ".join(f"{c}:{100 * rate(0, c * GiB, eq=eq, wgt=wgt):.0f}" for c in range(16, 202, 8)), flush=True)
print("(c2) taps + weights, llr at 112, rx start 0 ... 96 in steps of 8 (and 58):", " ".join(f"{r}:{100 * rate(r * GiB, 112 * GiB, eq=eq, wgt=wgt):.0f}" for r in list(range(0, 100, 8)) + [58]), flush=True)
print("(d) 16-QAM (17.2 GB of LLRs), rx at 0, llr start 16 ... 200 in steps of 8:", " ".join(f"{c}:{100 * rate(0, c * GiB, bps=4):.0f}" for c in range(16, 202, 8)), flush=True)
print("(e) small batch (64 Ki symbols: 0.6 + 1.5 GiB), rx at 0, llr start 1 ... 70 (GiB):", " ".join(f"{c}:{100 * rate(0, c * GiB, n_sym=1 << 16, steps=20):.0f}" for c in (1, 2, 8, 32, 60, 62, 63, 64, 66, 70)), flush=True)
print("(e2) small batch, llr start 62.5 ... 63.75 GiB in steps of 0.25:", " ".join(f"{c}:{100 * rate(0, int(c * GiB), n_sym=1 << 16, steps=20):.0f}" for c in (62.5, 62.75, 63.0, 63.25, 63.5, 63.75)), flush=True)
ctx.close()
