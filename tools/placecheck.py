#!/usr/bin/env python3
"""Does the traffic model see the placement map the receive kernel sees?  One 230 GiB pool; for a list of (rx, llr)
offsets print the kernel's and the model's fraction of 8 TB/s, then what sdr_placement_search_c32 picks."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sdr_amd.ops import Context

ctx = Context(0)
n, cp, bps, n_sym = 1024, 256, 6, 1 << 20
GiB = 1 << 30
POOL = int(os.environ.get("POOL_GIB", "230"))
pool = torch.empty((POOL * GiB,), dtype=torch.uint8, device="cuda")
f32 = pool.view(torch.float32)
for i in range(0, f32.numel(), 1 << 30):
    f32[i:i + (1 << 30)].uniform_(-0.05, 0.05)


def rates(rx_off, llr_off, steps=6):
    rx_bytes, llr_bytes = n_sym * (n + cp) * 8, n_sym * n * bps * 4
    rxc = pool[rx_off:rx_off + rx_bytes].view(torch.float32).view(n_sym, (n + cp) * 2).view(torch.complex64)
    llr = pool[llr_off:llr_off + llr_bytes].view(torch.float32).view(n_sym, n, bps)
    out = []
    for fn in (lambda: ctx.demod(rxc, n, cp, bps, out_llr=llr, llr_weight=1.5), lambda: ctx.copy_shape(rxc, n, cp, bps, llr)):
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out.append((8 * n + 4 * n * bps) * n_sym / (e0.elapsed_time(e1) / steps * 1e-3) / 8e12 * 100)
    return out


print("rx@0, llr start (GiB): kernel / model")
for c in range(16, POOL - 24, 8):
    k, m = rates(0, c * GiB)
    print(f"  llr {c:3d}: {k:5.1f} / {m:5.1f}", flush=True)
print("llr@112, rx start: kernel / model")
for r in list(range(0, 104, 8)) + [144, 176]:
    k, m = rates(r * GiB, 112 * GiB)
    print(f"  rx {r:3d}: {k:5.1f} / {m:5.1f}", flush=True)
for step in (0, 4 * GiB):
    t0 = time.time()
    i, o, best, worst = ctx.placement_search(pool, n, cp, bps, n_sym, step)
    torch.cuda.synchronize()
    dt = time.time() - t0
    f32[i // 4:i // 4 + n_sym * (n + cp) * 2].uniform_(-0.05, 0.05)
    k, m = rates(i, o)
    print(f"search step {step / GiB:.0f} GiB: in {i / GiB:.3f} out {o / GiB:.3f} GiB, model best {best / 80:.1f} worst {worst / 80:.1f} %, {dt:.2f} s; kernel there {k:.1f}, model {m:.1f}", flush=True)
ctx.close()
