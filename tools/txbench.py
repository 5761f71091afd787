#!/usr/bin/env python3
"""ofdm::tx batch (bins in, time samples with prefix out) over n_fft: symbols/s and % of 8 TB/s on 8N + 8(N + cp) bytes per symbol."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sdr_amd.ops import Context

ctx = Context(0)
for n in [int(a) for a in sys.argv[1:]] or [64, 128, 256, 512, 1024, 2048, 4096, 8192]:
    cp = n // 4
    n_sym = min(1 << 22, (6 << 30) // (8 * n + 8 * (n + cp)))
    g = torch.Generator(device="cuda").manual_seed(1)
    freq = torch.randn((n_sym, n * 2), dtype=torch.float32, device="cuda", generator=g).view(torch.complex64)
    for _ in range(6):
        out = ctx.ofdm_tx(freq, n, cp)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        out = ctx.ofdm_tx(freq, n, cp)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    b = (8 * n + 8 * (n + cp)) * n_sym
    print(f"ofdm::tx n_fft {n:5d} n_sym {n_sym:8d} {ms:7.3f} ms  {n_sym / ms / 1e3:9.1f} Msym/s  {b / ms / 1e6:7.1f} GB/s  {b / ms / 1e6 / 80:5.1f} %", flush=True)
    del freq, out
    torch.cuda.empty_cache()
ctx.close()
