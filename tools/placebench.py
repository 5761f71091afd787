#!/usr/bin/env python3
"""Does the rate depend on where the buffers sit?  One pool, the receive kernel's input at a fixed
offset and its LLR output at a swept offset behind it; every placement is timed several times.

  python tools/placebench.py --case 2048:6 --symbols 98304
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", default="2048:6")
    ap.add_argument("--symbols", type=int, default=98304)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--deltas", type=int, nargs="+", default=[0, 256, 1024, 4096, 16384, 65536, 262144, 1 << 20, (1 << 20) + 4096, 3 << 19, 2 << 20])
    ap.add_argument("--fresh", type=int, default=0, help="instead of sweeping offsets: this many fresh allocations (empty_cache between)")
    args = ap.parse_args()
    import torch
    from sdr_amd.ops import Context
    ctx = Context(0)
    n, bps = (int(v) for v in args.case.split(":")[:2])
    cp = n // 4
    n_sym = args.symbols
    in_bytes = n_sym * (n + cp) * 8
    out_bytes = n_sym * n * bps * 4
    alg = n_sym * (8 * n + 4 * n * bps)

    def time_it(rx, out):
        for _ in range(2):
            ctx.demod(rx, n, cp, bps, llr_weight=1.5, out_llr=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.steps):
            ctx.demod(rx, n, cp, bps, llr_weight=1.5, out_llr=out)
        e1.record()
        torch.cuda.synchronize()
        return alg / (e0.elapsed_time(e1) / args.steps) / 1e6

    if args.fresh:
        for i in range(args.fresh):
            rx = torch.randn((n_sym, (n + cp) * 2), dtype=torch.float32, device="cuda").mul_(0.05).view(torch.complex64)
            out = torch.empty((n_sym, n, bps), dtype=torch.float32, device="cuda")
            r = [time_it(rx, out) for _ in range(3)]
            print(f"fresh {i}: rx {rx.data_ptr():#x} out {out.data_ptr():#x}  " + " ".join(f"{v:7.1f}" for v in r) + " GB/s", flush=True)
            del rx, out
            torch.cuda.empty_cache()
    else:
        pool = torch.empty(in_bytes + out_bytes + (8 << 20), dtype=torch.uint8, device="cuda")
        base = (-pool.data_ptr()) % (2 << 20)          # 2 MiB aligned start
        rx = pool[base: base + in_bytes].view(torch.float32)
        rx.normal_().mul_(0.05)
        rx = rx.view(torch.complex64).view(n_sym, n + cp)
        print(f"pool {pool.data_ptr():#x} rx at +{base:#x}")
        for d in args.deltas:
            o = base + ((in_bytes + (2 << 20) - 1) // (2 << 20)) * (2 << 20) + d
            out = pool[o: o + out_bytes].view(torch.float32).view(n_sym, n, bps)
            r = [time_it(rx, out) for _ in range(3)]
            print(f"delta {d:9d}: " + " ".join(f"{v:7.1f}" for v in r) + " GB/s", flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
