#!/usr/bin/env python3
"""from_constl: word stores (4-byte aligned output) against the byte-store path (output misaligned by one byte), same process, same input."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sdr_amd import capi
from sdr_amd.ops import Context

ctx = Context(0)
n_sym = 1 << 28
for bps in (1, 2, 4, 6, 8, 10):
    data = torch.randint(0, 256, (n_sym * bps // 8,), dtype=torch.uint8, device="cuda")
    sym = ctx.to_constl(data, bps, n_sym)
    buf = torch.empty(n_sym * bps // 8 + 8, dtype=torch.uint8, device="cuda")
    res = []
    for off in (0, 1, 0, 1):
        out = buf[off: off + n_sym * bps // 8]
        fn = lambda: capi.check(ctx.lib.sdr_from_constl_c32(ctx._h, bps, sym.data_ptr(), n_sym, out.data_ptr()))
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 8
        res.append(n_sym * (8 + bps / 8) / ms / 1e6)
        assert torch.equal(out, data)
    print(f"bps={bps:2d} from_constl GB/s: words {res[0]:7.1f} {res[2]:7.1f} | bytes {res[1]:7.1f} {res[3]:7.1f}", flush=True)
    del data, sym, buf
ctx.close()
