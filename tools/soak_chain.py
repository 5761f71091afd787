#!/usr/bin/env python3
"""Random shapes of the chunked faded link (sdr_link_chain_c32) in both forms -- one kernel per chunk, generator + receive kernel --
against the oracle's chain: counters and the last chunk's LLRs, bit for bit.  usage: soak_chain.py [cases] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from oracle import binding as oracle
from sdr_amd.link import noise_sigma
from sdr_amd.ops import Context

count = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = Context(0)
EBN0 = {1: 5.0, 2: 8.0, 4: 12.0, 6: 18.0, 8: 24.0, 10: 30.0}
bad = 0
for i in range(count):
    n = int(rng.choice([64, 128, 256, 512, 1024, 2048, 4096, 8192]))
    bps = int(rng.choice([1, 2, 4, 6, 8, 10]))
    cp = 2 * int(rng.integers(0, n // 2 + 1))
    per = int(rng.integers(1, 15))
    n_sym = int(rng.integers(1, (60 if n <= 1024 else 16)))
    chunk = per * int(rng.integers(1, 6))
    g0 = per * int(rng.integers(0, 1 << 20))
    seed = int(rng.integers(0, 1 << 40))
    sigma = np.float32(noise_sigma(n, bps, EBN0[bps] - float(rng.uniform(0, 6))))
    inv_n0 = np.float32(bps * 10 ** (EBN0[bps] / 10))
    want, want_llr = oracle.link_chain(n, cp, bps, seed, g0, n_sym, per, sigma, inv_n0, want_llr=True)
    for form in (1, 2):
        ctx.lib.sdr_diag_set_tuning(b"chain_form", form)
        cnt, llr, _ = ctx.link_chain(n, cp, bps, seed, g0, n_sym, chunk, per, float(sigma), float(inv_n0))
        torch.cuda.synchronize()
        eff = min(chunk, ((n_sym + per - 1) // per) * per)
        last = n_sym - ((n_sym - 1) // eff) * eff
        ok = np.array_equal(cnt.cpu().numpy().astype(np.uint64), want) and llr.shape[0] == last and np.array_equal(llr.cpu().numpy().view(np.uint32), want_llr[n_sym - last:].view(np.uint32))
        if not ok:
            bad += 1
            print(f"MISMATCH case {i} form {form}: n {n} cp {cp} bps {bps} per {per} n_sym {n_sym} chunk {chunk} g0 {g0} seed {seed}", flush=True)
    if i % 20 == 19:
        print(f"{i + 1} cases, {bad} mismatches", flush=True)
ctx.lib.sdr_diag_set_tuning(b"chain_form", 0)
ctx.close()
print(f"done: {count} cases x 2 forms, {bad} mismatches")
sys.exit(1 if bad else 0)
