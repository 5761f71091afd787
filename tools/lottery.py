#!/usr/bin/env python3
"""Does a workload's rate depend on what ran (and was freed) before it in the same process?
usage: lottery.py name[,name...]   -- runs the listed bench.py workloads in order, prints frac of each."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sdr_amd.ops import Context
ctx = Context(0)
for name in sys.argv[1].split(","):
    r = bench.time_workload(ctx, torch, None, name, 8, 5, 1, 0)
    st = torch.cuda.memory_stats()
    print(f"{name:20s} frac {r['roofline']['frac']:.4f}  reserved {st['reserved_bytes.all.current'] / 2**30:.1f} GiB  segments {st['segment.all.current']}  {r['buffers']}", flush=True)
