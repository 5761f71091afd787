#!/usr/bin/env python3
"""Where inside device memory do the buffers have to lie?  ONE allocation of --pool GiB; the sample and LLR buffers of the
FFT-1024 / 64-QAM workload (1 Mi symbols: 10.7 GB in, 25.8 GB out) are views at chosen offsets inside it, so that their
placement is the only thing that changes between measurements.
usage: poolbench.py [--pool 200] [--eq]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pool", type=int, default=200)
    ap.add_argument("--bps", type=int, default=6)
    ap.add_argument("--steps", type=int, default=10)
    args = ap.parse_args()
    import torch
    from sdr_amd.ops import Context
    ctx = Context(0)
    n, cp, bps, n_sym = 1024, 256, args.bps, 1 << 20
    GiB = 1 << 30
    pool = torch.empty((args.pool * GiB,), dtype=torch.uint8, device="cuda")
    rx_bytes = n_sym * (n + cp) * 8
    llr_bytes = n_sym * n * bps * 4
    bytes_launch = (8 * n + 4 * n * bps) * n_sym
    print(f"pool {args.pool} GiB at {pool.data_ptr():#x}; rx {rx_bytes / GiB:.2f} GiB, llr {llr_bytes / GiB:.2f} GiB", flush=True)
    g = torch.Generator(device="cuda").manual_seed(1)

    def views(rx_off, llr_off):
        rx = pool[rx_off:rx_off + rx_bytes].view(torch.float32).view(n_sym, (n + cp) * 2)
        llr = pool[llr_off:llr_off + llr_bytes].view(torch.float32).view(n_sym, n, bps)
        return rx, llr

    def rate(rx_off, llr_off):
        rx, llr = views(rx_off, llr_off)
        rx.normal_(generator=g).mul_((2.0 * n) ** -0.5)
        rxc = rx.view(torch.complex64)
        for _ in range(6):
            ctx.demod(rxc, n, cp, bps, llr_weight=1.5, out_llr=llr)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.steps):
            ctx.demod(rxc, n, cp, bps, llr_weight=1.5, out_llr=llr)
        e1.record()
        torch.cuda.synchronize()
        return bytes_launch / (e0.elapsed_time(e1) / args.steps * 1e-3) / 8e12

    top = args.pool * GiB
    print("rx at 0, llr moves (GiB):", flush=True)
    for off in (11, 12, 16, 24, 32, 48, 64, 96, 128, 160):
        if off * GiB + llr_bytes <= top:
            print(f"   llr@{off:4d} GiB: {rate(0, off * GiB):.3f}", flush=True)
    print("llr at 0, rx moves (GiB):", flush=True)
    for off in (25, 26, 32, 48, 64, 96, 128, 160, 180):
        if off * GiB + rx_bytes <= top:
            print(f"   rx@{off:4d} GiB: {rate(off * GiB, 0):.3f}", flush=True)
    print("both move together (rx at base, llr right behind it):", flush=True)
    for base in (0, 8, 16, 32, 40, 64, 80, 100, 120, 140, 160):
        if (base + 11) * GiB + llr_bytes <= top:
            print(f"   base {base:4d} GiB: {rate(base * GiB, (base + 11) * GiB):.3f}", flush=True)
    print("fine offsets of llr behind rx at 0 (MiB beyond 11 GiB):", flush=True)
    for mib in (0, 2, 4, 8, 16, 64, 256, 512):
        print(f"   +{mib:4d} MiB: {rate(0, 11 * GiB + mib * (1 << 20)):.3f}", flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
