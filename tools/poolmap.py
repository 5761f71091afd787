#!/usr/bin/env python3
"""Map of the receive kernel's rate over the placement of its two big buffers inside ONE allocation (see poolbench.py):
rows = offset of the sample buffer, columns = offset of the LLR buffer, both in GiB; cells = fraction of 8 TB/s.
usage: poolmap.py [--pool 230] [--step 8] [--bps 6] [--eq]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pool", type=int, default=230)
    ap.add_argument("--step", type=int, default=8)
    ap.add_argument("--rx-step", type=int, default=16)
    ap.add_argument("--bps", type=int, default=6)
    ap.add_argument("--eq", action="store_true")
    ap.add_argument("--steps", type=int, default=6)
    args = ap.parse_args()
    import torch
    from sdr_amd.ops import Context
    ctx = Context(0)
    n, cp, bps, n_sym, per = 1024, 256, args.bps, 1 << 20, 14
    GiB = 1 << 30
    pool = torch.empty((args.pool * GiB,), dtype=torch.uint8, device="cuda")
    f32 = pool.view(torch.float32)
    chunk = 1 << 30
    for i in range(0, f32.numel(), chunk):                        # finite, noise-like floats everywhere
        f32[i:i + chunk].uniform_(-0.05, 0.05)
    rx_bytes = n_sym * (n + cp) * 8
    llr_bytes = n_sym * n * bps * 4
    bytes_launch = (8 * n + 4 * n * bps) * n_sym
    eq = wgt = None
    if args.eq:
        g = torch.Generator(device="cuda").manual_seed(1)
        n_sets = (n_sym + per - 1) // per
        eq = torch.randn((n_sets, n * 2), dtype=torch.float32, device="cuda", generator=g).view(torch.complex64)
        wgt = torch.rand((n_sets, n), dtype=torch.float32, device="cuda", generator=g)
    print(f"pool {args.pool} GiB at {pool.data_ptr():#x}; rx {rx_bytes / GiB:.2f} GiB, llr {llr_bytes / GiB:.2f} GiB, {'taps + weights' if args.eq else 'no taps'}", flush=True)

    def rate(rx_off, llr_off):
        rxc = pool[rx_off:rx_off + rx_bytes].view(torch.float32).view(n_sym, (n + cp) * 2).view(torch.complex64)
        llr = pool[llr_off:llr_off + llr_bytes].view(torch.float32).view(n_sym, n, bps)
        kw = dict(eq=eq, llr_weight=wgt, sym_per_set=per) if args.eq else dict(llr_weight=1.5)
        for _ in range(3):
            ctx.demod(rxc, n, cp, bps, out_llr=llr, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.steps):
            ctx.demod(rxc, n, cp, bps, out_llr=llr, **kw)
        e1.record()
        torch.cuda.synchronize()
        return bytes_launch / (e0.elapsed_time(e1) / args.steps * 1e-3) / 8e12

    cols = [c for c in range(0, args.pool, args.step) if c * GiB + llr_bytes <= args.pool * GiB]
    print("rx\\llr " + " ".join(f"{c:4d}" for c in cols), flush=True)
    for r in range(0, args.pool, args.rx_step):
        if r * GiB + rx_bytes > args.pool * GiB:
            break
        cells = []
        for c in cols:
            lo, hi = c * GiB, c * GiB + llr_bytes
            if not (hi <= r * GiB or lo >= r * GiB + rx_bytes):
                cells.append("   .")
                continue
            cells.append(f"{100 * rate(r * GiB, c * GiB):4.0f}")
        print(f"{r:6d} " + " ".join(cells), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
