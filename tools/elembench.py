#!/usr/bin/env python3
"""Rates of the element-wise constellation kernels (to_constl / from_constl / nearest / soft_demap)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timeit(torch, fn, reps=8):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    import torch
    from sdr_amd.ops import Context
    ctx = Context(0)
    n_sym = 1 << 28
    for bps in (2, 4, 6, 8):
        data = torch.randint(0, 256, (n_sym * bps // 8,), dtype=torch.uint8, device="cuda")
        sym = ctx.to_constl(data, bps, n_sym)
        rows = [
            ("to_constl", lambda: ctx.to_constl(data, bps, n_sym), n_sym * (bps / 8 + 8)),
            ("from_constl", lambda: ctx.from_constl(sym, bps), n_sym * (8 + bps / 8)),
            ("nearest", lambda: ctx.nearest(sym, bps), n_sym * (8 + 2)),
            ("soft_demap", lambda: ctx.soft_demap(sym, bps, 1.5), n_sym * (8 + 4 * bps)),
        ]
        for name, fn, nbytes in rows:
            ms = timeit(torch, fn)
            print(f"bps={bps} {name:>12s}: {n_sym / ms / 1e6:8.2f} Gsym/s  {nbytes / ms / 1e6:8.1f} GB/s", flush=True)
        del data, sym
        torch.cuda.empty_cache()
    ctx.close()


if __name__ == "__main__":
    main()
