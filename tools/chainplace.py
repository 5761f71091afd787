#!/usr/bin/env python3
"""The chunked whole link (sdr_link_chain_c32, BASELINE configs[2] at 4 Mi symbols) with its workspace at different offsets of one
200 GiB allocation: does the placement map of the receive kernel (profiles/r3_placement_map.txt) show in the chain's rate?"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import CHAIN
from sdr_amd.link import noise_sigma
from sdr_amd.ops import Context

ctx = Context(0)
c = dict(CHAIN["fft2048_64qam_eq_4M_chain"])
GiB = 1 << 30
POOL = int(os.environ.get("POOL_GIB", "200"))
STEP = float(os.environ.get("STEP_GIB", "4"))
pool = torch.empty((POOL * GiB,), dtype=torch.uint8, device="cuda")
sigma = noise_sigma(c["n_fft"], c["bps"], c["ebn0"])
inv_n0 = c["bps"] * 10 ** (c["ebn0"] / 10)
need = ctx.link_chain_workspace(c["n_fft"], c["cp"], c["bps"], c["n_sym"], c["chunk"], c["per"])
llr_bytes = (c["chunk"] * c["n_fft"] * c["bps"] * 4 + 255) // 256 * 256
print(f"workspace {need / GiB:.2f} GiB, LLR buffer = its last {llr_bytes / GiB:.2f} GiB (starts {(need - llr_bytes) / GiB:.2f} GiB in)", flush=True)
cnt = torch.zeros(8, dtype=torch.int64, device="cuda")


def rate(off):
    ws = pool[off:off + need]
    ctx.link_chain(c["n_fft"], c["cp"], c["bps"], c["seed"], 0, c["n_sym"], c["chunk"], c["per"], sigma, inv_n0, workspace=ws, counters=cnt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2):
        ctx.link_chain(c["n_fft"], c["cp"], c["bps"], c["seed"], 0, c["n_sym"], c["chunk"], c["per"], sigma, inv_n0, workspace=ws, counters=cnt)
    torch.cuda.synchronize()
    return c["n_sym"] / ((time.perf_counter() - t0) / 2)


off = 0.0
while off * GiB + need <= POOL * GiB:
    o = int(off * GiB) // (2 << 20) * (2 << 20)
    r = rate(o)
    print(f"workspace at {off:6.1f} GiB (LLR at {(o + need - llr_bytes) / GiB:6.1f} ... {(o + need) / GiB:6.1f}): {r / 1e6:6.2f} M symbols/s", flush=True)
    off += STEP
ctx.close()
