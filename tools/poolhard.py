#!/usr/bin/env python3
"""Read-dominated kernels on views of one 230 GiB allocation: hard bits only (8 KiB read, 768 B written per FFT-1024 / 64-QAM symbol) and
bins out (ofdm::rx, 8 KiB read and written), with the sample buffer at different offsets.  Does the placement map of the LLR kernel
(profiles/r3_placement_map.txt) hold for them?"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sdr_amd import capi
from sdr_amd.ops import Context

ctx = Context(0)
n, cp, bps, n_sym = 1024, 256, 6, 1 << 20
GiB = 1 << 30
POOL = 230
pool = torch.empty((POOL * GiB,), dtype=torch.uint8, device="cuda")
f32 = pool.view(torch.float32)
for i in range(0, f32.numel(), 1 << 30):
    f32[i:i + (1 << 30)].uniform_(-0.05, 0.05)
rx_bytes = n_sym * (n + cp) * 8


def rx_at(off):
    return pool[off:off + rx_bytes].view(torch.float32).view(n_sym, (n + cp) * 2).view(torch.complex64)


def timed(fn, bytes_, steps=6):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return bytes_ / (e0.elapsed_time(e1) / steps * 1e-3) / 8e12 * 100


hard_bytes = n_sym * n * bps // 8
bins_bytes = n_sym * n * 8
print("hard bits only, hard buffer at 220 GiB; rx start (GiB): % of 8 TB/s")
hard = pool[220 * GiB:220 * GiB + hard_bytes]
print(" ".join(f"{r}:{timed(lambda: ctx.demod(rx_at(r * GiB), n, cp, bps, want_llr=False, out_hard=hard), (8 * n + n * bps // 8) * n_sym):.1f}" for r in range(0, 208, 4)), flush=True)
print("hard bits only, rx at 100 GiB; hard buffer start (GiB):")
rx = rx_at(100 * GiB)
print(" ".join(f"{h}:{timed(lambda: ctx.demod(rx, n, cp, bps, want_llr=False, out_hard=pool[h * GiB:h * GiB + hard_bytes]), (8 * n + n * bps // 8) * n_sym):.1f}" for h in list(range(0, 96, 8)) + [63, 63.5, 64] + list(range(112, 224, 8)) if True for h in [int(h * 2) / 2] if True for h in [h] ), flush=True) if False else None
for hh in list(range(0, 96, 8)) + [120, 127, 128, 150, 159, 160, 200]:
    hb = pool[int(hh * GiB):int(hh * GiB) + hard_bytes]
    print(f"  hard {hh}: {timed(lambda: ctx.demod(rx, n, cp, bps, want_llr=False, out_hard=hb), (8 * n + n * bps // 8) * n_sym):.1f}", flush=True)
print("bins out (sdr_ofdm_rx_c32), rx at 0; bins start (GiB):")
rx = rx_at(0)
for b in range(12, 208, 6):
    bins = pool[b * GiB:b * GiB + bins_bytes].view(torch.float32).view(n_sym, n * 2).view(torch.complex64)
    r = timed(lambda: capi.check(ctx.lib.sdr_ofdm_rx_c32(ctx._h, rx.data_ptr(), n, cp, n_sym, bins.data_ptr())), 16 * n * n_sym)
    print(f"  bins {b}: {r:.1f}", flush=True)
ctx.close()
