#!/usr/bin/env python3
"""Times the device frame generator (sdr_link_generate_c32) and the batched ofdm::tx on resident buffers."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from sdr_amd.ops import Context
    ctx = Context(0)
    for n, bps in [(1024, 4), (1024, 6), (2048, 6), (4096, 8), (8192, 6), (512, 4), (64, 2)]:
        cp = n // 4
        n_sym = min(1 << 18, (4 << 30) // ((n + cp) * 8))
        out = torch.empty((n_sym, n + cp), dtype=torch.complex64, device="cuda")
        freq = torch.randn((n_sym, n * 2), dtype=torch.float32, device="cuda").view(torch.complex64)
        cases = {
            "generate+noise": lambda: ctx.link_generate(n, cp, bps, 1, 0, n_sym, 0.01, out_rx=out, want_bits=False),
            "generate": lambda: ctx.link_generate(n, cp, bps, 1, 0, n_sym, 0.0, out_rx=out, want_bits=False),
            "ofdm_tx": lambda: ctx.ofdm_tx(freq, n, cp),
        }
        for name, fn in cases.items():
            for _ in range(6):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(12):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 12
            wr = n_sym * (n + cp) * 8 / ms / 1e6
            print(f"N={n:5d} bps={bps} {name:>15s}: {n_sym / ms / 1e3:8.2f} Msym/s  written {wr:7.1f} GB/s", flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
