#!/usr/bin/env python3
"""Occupancy sweep of the receive kernels on ONE set of buffers per case (fresh allocations differ
by up to 10 % in write rate on this GPU, see DESIGN.md: sweeping on fresh buffers measures the
allocation lottery, not the knob).  For every case: allocate once, then time each SDR lds_pad value
(sdr_diag_set_tuning) `--rounds` times, interleaved.

  python tools/occbench.py --cases 1024:6 1024:6:eq --pads 0 3000 6656 10752
"""
import argparse
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", nargs="+", default=["1024:6"])
    ap.add_argument("--pads", type=int, nargs="+", default=[0, 3000, 6656, 8476, 10752, 13677])
    ap.add_argument("--knob", default="lds_pad", help="sdr_diag_set_tuning knob swept by --pads (lds_pad, xcd_window, xcd_stagger)")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--symbols", type=int, default=0)
    ap.add_argument("--placed", action="store_true", help="sample and LLR buffers placed by sdr_placement_search_c32 in a 180 GiB arena")
    args = ap.parse_args()
    import torch
    from sdr_amd.ops import Context
    ctx = Context(0)
    arena = None
    if args.placed:
        from sdr_amd.placement import Arena
        arena = Arena(ctx, 180 << 30)
    for case in args.cases:
        parts = case.split(":")
        n, bps = int(parts[0]), int(parts[1])
        with_eq, with_hard, no_llr = "eq" in parts[2:], "hard" in parts[2:], "nollr" in parts[2:]
        cp = n // 4
        bytes_sym = 8 * n + (0 if no_llr else 4 * n * bps) + (n * bps // 8 if with_hard else 0)
        n_sym = args.symbols or max(1024, min(1 << 20, (8 << 30) // (bytes_sym + 8 * cp)))
        g = torch.Generator(device="cuda").manual_seed(1)
        placed_out = None
        if arena is not None and not no_llr:
            arena.reset()
            rx, placed_out = arena.place_demod(n, cp, bps, n_sym)
            f = rx.view(torch.float32).view(-1)
            for i in range(0, f.numel(), 1 << 28):
                f[i:i + (1 << 28)].normal_(generator=g).mul_((2.0 * n) ** -0.5)
        else:
            rx = torch.randn((n_sym, (n + cp) * 2), dtype=torch.float32, device="cuda", generator=g).mul_((2.0 * n) ** -0.5).view(torch.complex64)
        eq = wgt = None
        per = 0
        if with_eq:
            per = 14
            n_sets = (n_sym + per - 1) // per
            eq = torch.randn((n_sets, n * 2), dtype=torch.float32, device="cuda", generator=g).view(torch.complex64)
            wgt = torch.rand((n_sets, n), dtype=torch.float32, device="cuda", generator=g)
        out = None if no_llr else (placed_out if placed_out is not None else torch.empty((n_sym, n, bps), dtype=torch.float32, device="cuda"))
        hard = torch.empty(n_sym * n * bps // 8, dtype=torch.uint8, device="cuda") if with_hard else None
        step = lambda: ctx.demod(rx, n, cp, bps, eq=eq, llr_weight=(wgt if wgt is not None else 1.5), sym_per_set=per, out_llr=out, want_llr=not no_llr, out_hard=hard)
        for _ in range(6):
            step()
        torch.cuda.synchronize()
        res = {p: [] for p in args.pads}
        for _ in range(args.rounds):
            for p in args.pads:
                ctx.lib.sdr_diag_set_tuning(args.knob.encode(), p)
                step()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.steps):
                    step()
                e1.record()
                torch.cuda.synchronize()
                res[p].append(bytes_sym * n_sym / (e0.elapsed_time(e1) / args.steps) / 1e6)
        ctx.lib.sdr_diag_set_tuning(args.knob.encode(), (1 << 30) if args.knob == "xcd_window" else 0)
        print(f"case {case:>14s} n_sym {n_sym}: " + "  ".join(f"{args.knob} {p}: {statistics.median(v) / 80:5.1f}%" for p, v in res.items()), flush=True)
        del rx, out, eq, wgt, hard
        torch.cuda.empty_cache()
    ctx.close()


if __name__ == "__main__":
    main()
