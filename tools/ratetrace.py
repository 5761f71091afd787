#!/usr/bin/env python3
"""Rate of the receive kernel over time under continuous load: does the GPU change state (clocks, power)?
usage: ratetrace.py [eq] [launches]   -- prints the HBM fraction of every block of 20 launches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sdr_amd.ops import Context
ctx = Context(0)
# optional: "hold=GiB" keeps a dummy allocation of that size ahead of the real buffers (which physical region do they get?)
hold = [int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("hold=")]
dummy = torch.empty((hold[0] << 30,), dtype=torch.uint8, device="cuda") if hold and hold[0] else None
use_eq = "eq" in sys.argv[1:]
launches = int([a for a in sys.argv[1:] if a.isdigit()][0]) if any(a.isdigit() for a in sys.argv[1:]) else 800
name = "fft1024_64qam_eq" if use_eq else "fft1024_64qam_1M"
n_fft, cp, bps, n_sym, ebn0, with_eq = bench.WORKLOADS[name]
B = bench.algorithmic_bytes_per_symbol(n_fft, bps) * n_sym
rx, eq, wgt, w0, per = bench.make_inputs(ctx, torch, n_fft, cp, bps, n_sym, ebn0, with_eq, seed=2, first_symbol=0)
llr = torch.zeros((n_sym, n_fft, bps), dtype=torch.float32, device="cuda")
torch.cuda.synchronize()
t_start = time.perf_counter()
blk = 20
evs = [torch.cuda.Event(enable_timing=True) for _ in range(launches // blk + 1)]
evs[0].record()
for i in range(launches):
    ctx.demod(rx, n_fft, cp, bps, eq=eq, llr_weight=(wgt if wgt is not None else w0), sym_per_set=per, out_llr=llr)
    if (i + 1) % blk == 0:
        evs[(i + 1) // blk].record()
torch.cuda.synchronize()
out = []
t = 0.0
for j in range(launches // blk):
    ms = evs[j].elapsed_time(evs[j + 1])
    t += ms
    out.append(f"{t / 1e3:5.2f}s {B * blk / (ms / 1e3) / 8e12:.3f}")
print(name, f"hold {hold[0] if hold else 0} GiB", " | ".join(out[:3] + out[-2:]))
