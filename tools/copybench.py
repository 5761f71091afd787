#!/usr/bin/env python3
"""Memory-system ceilings on this GPU for the access patterns of the receive path (torch copies)."""
import torch
def t(fn, steps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps
n, cp, s = 1024, 256, 400000
a = torch.randn((s, (n + cp) * 2), device="cuda")
b = torch.empty((s, n * 2), device="cuda")
ms = t(lambda: b.copy_(a[:, cp * 2:]))
print(f"strided copy (skip CP) 8K of 10K -> 8K: {2 * b.numel() * 4 / ms / 1e6:.0f} GB/s")
c = torch.empty_like(b)
ms = t(lambda: c.copy_(b))
print(f"contiguous copy: {2 * b.numel() * 4 / ms / 1e6:.0f} GB/s")
d = torch.empty((s, n * 6), device="cuda")
ms = t(lambda: d.fill_(1.0))
print(f"fill (write only): {d.numel() * 4 / ms / 1e6:.0f} GB/s")
ms = t(lambda: b.sum())
print(f"sum (read only): {b.numel() * 4 / ms / 1e6:.0f} GB/s")
big = torch.empty((s, n * 8), device="cuda")
ms = t(lambda: big[:, : n * 6].copy_(d))   # read 24K write 24K
print(f"copy 24K rows: {2 * d.numel() * 4 / ms / 1e6:.0f} GB/s")
