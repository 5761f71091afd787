#!/usr/bin/env python3
"""Kernel micro-benchmark for tuning: times sdr_ofdm_demod_c32 / sdr_ofdm_rx_c32 on resident data.

  python tools/kbench.py --cases 1024:4 1024:6 1024:0 4096:8 --symbols 262144 --steps 10

case = n_fft:bps[:eq][:hard][:cnt]   (bps 0 = FFT only / ofdm::rx batch; bps 'fft' / 'ifft' = in-place fft::fft2 / ifft2 batch; eq = per-subcarrier
equaliser sets; hard = also packed hard bits; cnt = error counters against reference bits; nollr = no LLR
output, i.e. with hard: the reference's own chain ofdm::rx -> from_constl; gen = the frame generator with
noise, GB/s = samples written; mc = the Monte-Carlo link kernel, GB/s = what the receive kernel would move)
Inputs are random complex noise (values do not change the instruction stream), except for hard:nollr, which runs on generated frames.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", nargs="+", default=["1024:4", "1024:6"])
    ap.add_argument("--symbols", type=int, default=0, help="symbols per case (default: ~8 GiB of input+output)")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=8, help="fresh buffers run slow for their first 2-3 launches (first touch)")
    ap.add_argument("--repeat", type=int, default=1, help="time every case this many times (interleaved) and report the median")
    ap.add_argument("--placed", action="store_true", help="sample and LLR buffers of the LLR-writing cases as views of one 180 GiB arena, placed by sdr_placement_search_c32 (DESIGN.md 5.1)")
    args = ap.parse_args()
    import torch
    from sdr_amd.ops import Context
    ctx = Context(0)
    arena = None
    if args.placed:
        from sdr_amd.placement import Arena
        arena = Arena(ctx, 180 << 30)
    import statistics
    results = {}
    for case in args.cases * args.repeat:
        parts = case.split(":")
        fft_only = parts[1] in ("fft", "ifft", "fft64")
        n, bps = int(parts[0]), (0 if fft_only else int(parts[1]))
        with_eq = "eq" in parts[2:]
        with_hard = "hard" in parts[2:]
        cp = n // 4
        bytes_sym = 8 * n + (4 * n * bps if bps else 8 * n)
        n_sym = args.symbols or max(1024, min(1 << 20, (8 << 30) // (bytes_sym + 8 * cp)))
        g = torch.Generator(device="cuda").manual_seed(1)
        placed_out = None
        if arena is not None and bps and "nollr" not in parts[2:] and not ("gen" in parts[2:] or "mc" in parts[2:] or "mch" in parts[2:]):
            arena.reset()
            rx, placed_out = arena.place_demod(n, cp, bps, n_sym)
            f = rx.view(torch.float32).view(-1)
            for i in range(0, f.numel(), 1 << 28):
                f[i:i + (1 << 28)].normal_(generator=g).mul_((2.0 * n) ** -0.5)
        elif bps and "nollr" in parts[2:] and "hard" in parts[2:]:
            # Hard decisions only: the margin slicer's fix-up path depends on how many bins lie near a decision boundary, so this
            # case is timed on received frames (generator + AWGN at a working Eb/N0), not on Gaussian noise
            import bench
            ebn0 = {1: 8.0, 2: 10.0, 4: 12.0, 6: 16.0, 8: 22.0, 10: 28.0}[bps]
            rx = bench.make_inputs(ctx, torch, n, cp, bps, n_sym, ebn0, False, seed=2, first_symbol=0)[0]
        else:
            rx = torch.randn((n_sym, (n + cp) * 2), dtype=torch.float32, device="cuda", generator=g).mul_((2.0 * n) ** -0.5).view(torch.complex64)      # bins of unit average energy, like a received constellation
        eq = wgt = None
        per = 0
        if with_eq:
            per = 14
            n_sets = (n_sym + per - 1) // per
            eq = torch.randn((n_sets, n * 2), dtype=torch.float32, device="cuda", generator=g).view(torch.complex64)
            wgt = torch.rand((n_sets, n), dtype=torch.float32, device="cuda", generator=g)
        hard = ref = cnt = None
        if bps and with_hard:
            hard = torch.empty(n_sym * n * bps // 8, dtype=torch.uint8, device="cuda")
            bytes_sym += n * bps // 8
        if bps and "cnt" in parts[2:]:
            ref = torch.randint(0, 256, (n_sym * n * bps // 8,), dtype=torch.uint8, device="cuda", generator=g)
            cnt = torch.zeros(8, dtype=torch.int64, device="cuda")
            bytes_sym += n * bps // 8
        no_llr = "nollr" in parts[2:]
        if no_llr:
            bytes_sym -= 4 * n * bps
        if "gen" in parts[2:] or "mc" in parts[2:] or "mch" in parts[2:]:
            # frame generator with noise (bytes = samples written) / Monte-Carlo link (no HBM traffic: reported as LLR-equivalent bytes)
            del rx
            rx = None
            if "gen" in parts[2:]:
                n_sym = args.symbols or min(1 << 18, (4 << 30) // ((n + cp) * 8))
                gout = torch.empty((n_sym, n + cp), dtype=torch.complex64, device="cuda")
                bytes_sym = (n + cp) * 8
                step = lambda: ctx.link_generate(n, cp, bps, 1, 0, n_sym, 0.01, out_rx=gout, want_bits=False)
            else:
                n_sym = args.symbols or (1 << 18)
                mcnt = torch.zeros(8, dtype=torch.int64, device="cuda")
                bytes_sym = 8 * n + 4 * n * bps
                if "mch" in parts[2:]:      # hard-only Monte-Carlo body, through the sweep entry point
                    step = lambda: ctx.link_sweep(n, cp, 1, [(bps, 0.01, 1.0, 0, n_sym)], mcnt.view(1, 8), hard_only=True)
                else:
                    step = lambda: ctx.link_ber(n, cp, bps, 1, 0, n_sym, 0.01, 1.0, counters=mcnt)
        elif bps:
            out = None if no_llr else (placed_out if placed_out is not None else torch.empty((n_sym, n, bps), dtype=torch.float32, device="cuda"))
            step = lambda: ctx.demod(rx, n, cp, bps, eq=eq, llr_weight=(wgt if wgt is not None else 1.5), sym_per_set=per, out_llr=out, want_llr=not no_llr,
                                     out_hard=hard, ref_hard=ref, counters=cnt)
        elif fft_only:
            buf = rx.view(-1)[: n_sym * n].view(n_sym, n)
            if parts[1] == "fft64":                                          # fp64 generic engine: 16 bytes per point each way
                buf = buf[: n_sym // 2].to(torch.complex128)
                bytes_sym = 32 * n
                n_sym = n_sym // 2
            step = lambda: ctx.fft2_(buf, inverse=(parts[1] == "ifft"))      # in place; values drift, timing does not care
        else:
            step = lambda: ctx.ofdm_rx(rx, n, cp)
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.steps):
            step()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.steps
        gbs = bytes_sym * n_sym / ms / 1e6
        results.setdefault(case, []).append(gbs)
        if args.repeat == 1:
            print(f"case {case:>12s} n_sym {n_sym:8d}  {ms:8.3f} ms/launch  {n_sym / ms / 1e3:9.2f} Msym/s  {gbs:8.1f} GB/s  {gbs / 80:5.1f} % of 8 TB/s", flush=True)
        del rx, eq, wgt, hard, ref, cnt
        out = gout = mcnt = None
        torch.cuda.empty_cache()
    if args.repeat > 1:
        for case, v in results.items():
            print(f"case {case:>12s} median {statistics.median(v):8.1f} GB/s ({statistics.median(v) / 80:5.1f} %)  min {min(v):8.1f}  max {max(v):8.1f}  n={len(v)}", flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
