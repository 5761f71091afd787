#!/usr/bin/env python3
"""Is the 5.7 / 6.3 TB/s split between runs a matter of WHICH XCD gets which eighth of the batch
(workgroup ids are dealt round-robin over the XCDs from wherever the previous dispatch ended) or of
the physical placement of the buffers?  For each of a few fresh allocations: time the receive kernel
after a rotator dispatch of r = 0..7 workgroups (which shifts the round-robin start by r)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", default="1024:6")
    ap.add_argument("--symbols", type=int, default=1 << 20)
    ap.add_argument("--fresh", type=int, default=3)
    ap.add_argument("--steps", type=int, default=5)
    args = ap.parse_args()
    import torch
    from sdr_amd.ops import Context
    ctx = Context(0)
    n, bps = (int(v) for v in args.case.split(":")[:2])
    cp = n // 4
    n_sym = args.symbols
    alg = n_sym * (8 * n + 4 * n * bps)
    tiny_rx = torch.zeros((8, n + cp), dtype=torch.complex64, device="cuda")
    tiny_out = torch.zeros((8, n, bps), dtype=torch.float32, device="cuda")
    for i in range(args.fresh):
        rx = torch.randn((n_sym, (n + cp) * 2), dtype=torch.float32, device="cuda").mul_(0.05).view(torch.complex64)
        out = torch.empty((n_sym, n, bps), dtype=torch.float32, device="cuda")
        for _ in range(3):
            ctx.demod(rx, n, cp, bps, llr_weight=1.5, out_llr=out)
        torch.cuda.synchronize()
        row = []
        for r in list(range(8)) + [0, 1]:
            if r:
                ctx.copy_shape(tiny_rx[:r], n, cp, bps, tiny_out[:r])          # r workgroups: shifts the round-robin start
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.steps):
                ctx.demod(rx, n, cp, bps, llr_weight=1.5, out_llr=out)
            e1.record()
            torch.cuda.synchronize()
            row.append(alg / (e0.elapsed_time(e1) / args.steps) / 1e6)
        print(f"fresh {i}: " + " ".join(f"{v:7.0f}" for v in row), flush=True)
        del rx, out
        torch.cuda.empty_cache()
    ctx.close()


if __name__ == "__main__":
    main()
