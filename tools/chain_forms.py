#!/usr/bin/env python3
"""The chunked faded link (sdr_link_chain_c32) as ONE kernel per chunk (chain_fused_kernel: samples in LDS only) against the
earlier form (generator + receive kernel, samples through the workspace), same process, same workspace; counters must agree.
  python tools/chain_forms.py [n_fft:bps[:nollr] ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sdr_amd.link import noise_sigma
from sdr_amd.ops import Context

ctx = Context(0)
cases = sys.argv[1:] or ["2048:6", "2048:6:nollr", "1024:4", "1024:6", "4096:8", "256:6", "64:2", "512:4", "8192:6", "2048:2", "2048:10"]
EBN0 = {1: 6.0, 2: 8.0, 4: 12.0, 6: 18.0, 8: 24.0, 10: 30.0}
for case in cases:
    parts = case.split(":")
    n, bps = int(parts[0]), int(parts[1])
    nollr = "nollr" in parts[2:]
    cp, per = n // 4, 14
    n_sym = (1 << 33) // (n * 8 * 14) * 14                      # 8 GiB of payload samples per pass
    chunk = max(per, (1 << 29) // (n * 14) * 14)                # 4 GiB of payload samples per chunk
    sigma, inv_n0 = noise_sigma(n, bps, EBN0[bps]), bps * 10 ** (EBN0[bps] / 10)
    need = ctx.link_chain_workspace(n, cp, bps, n_sym, chunk, per, want_llr=not nollr)
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    out = {}
    for form in (0, 1, 0, 1):
        ctx.lib.sdr_diag_set_tuning(b"chain_form", 1 + form)
        cnt = torch.zeros(8, dtype=torch.int64, device="cuda")
        kw = dict(workspace=ws, counters=cnt, want_llr=not nollr)
        ctx.link_chain(n, cp, bps, 3, 0, n_sym, chunk, per, sigma, inv_n0, **kw)
        torch.cuda.synchronize()
        cnt.zero_()
        t0 = time.perf_counter()
        for _ in range(2):
            ctx.link_chain(n, cp, bps, 3, 0, n_sym, chunk, per, sigma, inv_n0, **kw)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 2
        out.setdefault(form, []).append((n_sym / dt, cnt.cpu().tolist()))
    ctx.lib.sdr_diag_set_tuning(b"chain_form", 0)
    one, two = max(r for r, _ in out[0]), max(r for r, _ in out[1])
    same = all(c == out[0][0][1] for f in out for _, c in out[f])
    print(f"case {case:>14s} {n_sym:9d} symbols in chunks of {chunk:7d}: one kernel {one / 1e6:8.2f}  two kernels {two / 1e6:8.2f} M symbols/s  ({one / two - 1:+.1%})  counters {'equal' if same else 'DIFFER'}", flush=True)
    del ws
    torch.cuda.empty_cache()
ctx.close()
