#!/usr/bin/env python3
"""Same-buffer A/B of two builds of libsdr_amd.so: both libraries are loaded into ONE process and time the
receive kernel alternately on the SAME tensors, so that neither the box nor the placement of the buffers
(profiles/r2_eq_placement_lottery.txt) can tell them apart.
usage: abbench.py libA.so libB.so --cases n_fft:bps[:eq|:taps][:hard][:cnt][:nollr][:fft] | n_fft:bins ...   (prints % of 8 TB/s for A, B, A - B;
taps = equaliser taps without a weight array (scalar weight), wft = the same with SDR_DEMOD_WEIGHT_FROM_TAPS, cnt = counters against reference bits)"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("lib_a")
    ap.add_argument("lib_b")
    ap.add_argument("--cases", nargs="+", required=True)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--b-round2", action="store_true", help="library B is a round-2 build: pass it the 96-byte sdr_demod_args")
    args = ap.parse_args()
    import torch
    from sdr_amd import capi
    from sdr_amd.ops import Context

    def ctx_of(path):
        capi._lib = None
        capi.LIB_PATH = os.path.abspath(path)
        capi.OLDER_BUILD = args.b_round2 and path == args.lib_b
        c = Context(0)          # binds the freshly loaded library
        return c
    ca, cb = ctx_of(args.lib_a), ctx_of(args.lib_b)
    if args.b_round2:
        cb.demod_struct_size = 96      # the round-2 sdr_demod_args (no `flags`)
    assert ca.lib is not cb.lib
    g = torch.Generator(device="cuda").manual_seed(1)
    for case in args.cases:
        parts = case.split(":")
        n = int(parts[0])
        bins_only = parts[1] == "bins"                      # ofdm::rx: strip the prefix, transform, bins out
        bps = 2 if bins_only else int(parts[1])
        wft = "wft" in parts[2:]
        with_eq, with_hard, no_llr, with_fft = "eq" in parts[2:] or "taps" in parts[2:] or wft, "hard" in parts[2:], "nollr" in parts[2:] or bins_only, "fft" in parts[2:]
        taps_only, with_cnt = "taps" in parts[2:] or wft, "cnt" in parts[2:]
        cp = n // 4
        bytes_sym = 8 * n + (0 if no_llr else 4 * n * bps) + (n * bps // 8 if with_hard else 0) + (8 * n if with_fft or bins_only else 0)
        n_sym = max(1024, min(1 << 20, (8 << 30) // (bytes_sym + 8 * cp)))
        rx = torch.randn((n_sym, (n + cp) * 2), dtype=torch.float32, device="cuda", generator=g).mul_((2.0 * n) ** -0.5).view(torch.complex64)
        eq = wgt = None
        per = 0
        if with_eq:
            per = 14
            n_sets = (n_sym + per - 1) // per
            eq = torch.randn((n_sets, n * 2), dtype=torch.float32, device="cuda", generator=g).view(torch.complex64)
            wgt = None if taps_only else torch.rand((n_sets, n), dtype=torch.float32, device="cuda", generator=g)
        ref = cnt = None
        if with_cnt:
            ref = torch.randint(0, 256, (n_sym * n * bps // 8,), dtype=torch.uint8, device="cuda", generator=g)
            cnt = torch.zeros(8, dtype=torch.int64, device="cuda")
        out = None if no_llr else torch.empty((n_sym, n, bps), dtype=torch.float32, device="cuda")
        hard = torch.empty(n_sym * n * bps // 8, dtype=torch.uint8, device="cuda") if with_hard else None
        bins = torch.empty((n_sym, n), dtype=torch.complex64, device="cuda") if with_fft or bins_only else None

        def rate(c):
            if bins_only:
                step = lambda: capi.check(c.lib.sdr_ofdm_rx_c32(c._h, rx.data_ptr(), n, cp, n_sym, bins.data_ptr()))
            else:
                step = lambda: c.demod(rx, n, cp, bps, eq=eq, llr_weight=(wgt if wgt is not None else 1.5), sym_per_set=per, out_llr=out, want_llr=not no_llr, out_hard=hard, out_fft=bins,
                                       ref_hard=ref, counters=cnt, weight_from_taps=wft)
            for _ in range(3):
                step()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.steps):
                step()
            e1.record()
            torch.cuda.synchronize()
            return bytes_sym * n_sym / (e0.elapsed_time(e1) / args.steps / 1e3) / 8e12 * 100
        for c in (ca, cb):           # first touch of fresh buffers
            rate(c)
        ra, rb = [], []
        for _ in range(args.rounds):
            ra.append(rate(ca))
            rb.append(rate(cb))
        ma, mb = sorted(ra)[len(ra) // 2], sorted(rb)[len(rb) // 2]
        print(f"case {case:>14s}  A {ma:5.1f}  B {mb:5.1f}  A-B {ma - mb:+5.1f}   (A {' '.join(f'{v:.1f}' for v in ra)} | B {' '.join(f'{v:.1f}' for v in rb)})", flush=True)
        del rx, eq, wgt, out, hard, bins, ref, cnt
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
