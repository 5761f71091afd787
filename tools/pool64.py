#!/usr/bin/env python3
"""FFT-64 / QPSK (configs[0]'s shape, 4 Mi symbols: 2.5 GiB of samples, 2 GiB of LLRs) on views of one 200 GiB allocation: is the small
kernel as sensitive to where its LLR buffer lies as the large ones, and does the traffic model see it?"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sdr_amd.ops import Context

ctx = Context(0)
n, cp, bps, n_sym = 64, 16, 2, 1 << 22
GiB = 1 << 30
pool = torch.empty((200 * GiB,), dtype=torch.uint8, device="cuda")
rx_bytes, llr_bytes = n_sym * (n + cp) * 8, n_sym * n * bps * 4
pool[:rx_bytes].view(torch.float32).uniform_(-0.05, 0.05)
rxc = pool[:rx_bytes].view(torch.float32).view(n_sym, (n + cp) * 2).view(torch.complex64)


def rates(llr_off, steps=20):
    llr = pool[llr_off:llr_off + llr_bytes].view(torch.float32).view(n_sym, n, bps)
    out = []
    for fn in (lambda: ctx.demod(rxc, n, cp, bps, out_llr=llr, llr_weight=1.5), lambda: ctx.copy_shape(rxc, n, cp, bps, llr)):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out.append((8 * n + 4 * n * bps) * n_sym / (e0.elapsed_time(e1) / steps * 1e-3) / 8e12 * 100)
    return out


for off in (8, 32, 60, 62, 62.5, 63, 63.5, 64, 80, 100, 126, 127, 127.5, 128, 140, 158, 159, 159.5, 160, 170, 190.5, 191, 191.5, 192):
    k, m = rates(int(off * GiB))
    print(f"llr at {off:6.1f} GiB: kernel {k:5.1f}  model {m:5.1f}", flush=True)
ctx.close()
