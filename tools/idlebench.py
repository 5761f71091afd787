#!/usr/bin/env python3
"""Does the rate depend on how long the GPU idled before the launches?  One set of buffers (FFT-1024 / 64-QAM, with or
without taps, 1 Mi symbols); after each idle period a train of launches is timed in groups, so that a decay from a
cool / power-rich start towards the sustained rate shows as a trajectory.
usage: idlebench.py [--eq] [--idle 0 5 20] [--groups 12] [--per-group 10]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--eq", action="store_true")
    ap.add_argument("--idle", type=float, nargs="+", default=[0.0, 5.0, 20.0])
    ap.add_argument("--groups", type=int, default=12)
    ap.add_argument("--per-group", type=int, default=10)
    args = ap.parse_args()
    import torch
    from sdr_amd.ops import Context
    ctx = Context(0)
    n, cp, bps, n_sym, per = 1024, 256, 6, 1 << 20, 14
    g = torch.Generator(device="cuda").manual_seed(1)
    rx = torch.randn((n_sym, (n + cp) * 2), dtype=torch.float32, device="cuda", generator=g).mul_((2.0 * n) ** -0.5).view(torch.complex64)
    eq = wgt = None
    if args.eq:
        n_sets = (n_sym + per - 1) // per
        eq = torch.randn((n_sets, n * 2), dtype=torch.float32, device="cuda", generator=g).view(torch.complex64)
        wgt = torch.rand((n_sets, n), dtype=torch.float32, device="cuda", generator=g)
    out = torch.empty((n_sym, n, bps), dtype=torch.float32, device="cuda")
    step = lambda: ctx.demod(rx, n, cp, bps, eq=eq, llr_weight=(wgt if wgt is not None else 1.5), sym_per_set=per if args.eq else 0, out_llr=out)
    bytes_launch = (8 * n + 4 * n * bps) * n_sym
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    for idle in args.idle:
        time.sleep(idle)
        fr = []
        for _ in range(args.groups):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.per_group):
                step()
            e1.record()
            torch.cuda.synchronize()
            fr.append(bytes_launch / (e0.elapsed_time(e1) / args.per_group * 1e-3) / 8e12)
        print(f"idle {idle:5.1f} s, then {args.groups} groups of {args.per_group} launches ({'taps + weights' if args.eq else 'no taps'}): " + " ".join(f"{f:.3f}" for f in fr), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
