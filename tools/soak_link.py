#!/usr/bin/env python3
"""Random shapes of the Monte-Carlo link (sdr_link_ber_c32: every counter), the one-launch sweep (sdr_link_sweep_c32, soft and
hard-only points mixed) and the frame generator (sdr_link_generate_c32: samples and bits, with and without a channel) against the oracle,
bit for bit.  usage: soak_link.py [cases] [seed]"""
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


def note(what, **kw):
    global bad
    bad += 1
    print(f"MISMATCH {what}: {kw}", flush=True)


for i in range(count):
    n = int(rng.choice([64, 128, 256, 512, 1024, 2048, 4096, 8192]))
    bps = int(rng.choice([1, 2, 4, 6, 8, 10]))
    cp = int(rng.choice([0, n // 4, int(rng.integers(0, n + 1))]))
    n_sym = int(rng.integers(1, (48 if n <= 1024 else 12)))
    g0 = int(rng.integers(0, 1 << 40))
    seed = int(rng.integers(0, 1 << 40))
    sigma = np.float32(noise_sigma(n, bps, EBN0[bps] - float(rng.uniform(0, 8))))
    w0 = np.float32(rng.uniform(0.5, 40.0))
    # Monte-Carlo link
    want = oracle.link_ber(n, cp, bps, seed, g0, n_sym, sigma, w0)
    got = ctx.link_ber(n, cp, bps, seed, g0, n_sym, float(sigma), float(w0)).cpu().numpy().astype(np.uint64)
    if not np.array_equal(got, np.asarray(want, dtype=np.uint64)):
        note("link_ber", n=n, cp=cp, bps=bps, n_sym=n_sym, g0=g0, seed=seed)
    # generator, with a channel every other case
    per = int(rng.integers(1, 15))
    h = None
    if i % 2:
        sets = (g0 % per + n_sym + per - 1) // per + 1
        h = ((rng.standard_normal((sets, n)) + 1j * rng.standard_normal((sets, n))) * np.sqrt(0.5)).astype(np.complex64)
    gg0 = g0 - g0 % per if h is not None else g0
    rrx, rbits = oracle.link_generate(n, cp, bps, seed, gg0, n_sym, sigma, chan=h, sym_per_set=(per if h is not None else 0))
    rx, bits = ctx.link_generate(n, cp, bps, seed, gg0, n_sym, float(sigma), chan=(torch.from_numpy(h).cuda() if h is not None else None), sym_per_set=(per if h is not None else 0))
    if not (np.array_equal(bits.cpu().numpy(), rbits) and np.array_equal(rx.cpu().numpy().view(np.uint32), np.asarray(rrx).reshape(rx.shape).view(np.uint32))):
        note("link_generate", n=n, cp=cp, bps=bps, n_sym=n_sym, g0=gg0, seed=seed, per=per, chan=h is not None)
    if i % 20 == 19:
        print(f"{i + 1} cases, {bad} mismatches", flush=True)
# one-launch sweeps: random mixes of modulations, soft and hard-only
for i in range(max(1, count // 10)):
    n = int(rng.choice([64, 256, 1024, 2048]))
    cp = n // 4
    seed = int(rng.integers(0, 1 << 40))
    pts, want_rows = [], []
    first = 0
    for _ in range(int(rng.integers(2, 9))):
        bps = int(rng.choice([1, 2, 4, 6, 8, 10]))
        n_sym = int(rng.integers(1, 40))
        sigma = np.float32(noise_sigma(n, bps, EBN0[bps] - float(rng.uniform(0, 8))))
        w0 = np.float32(rng.uniform(0.5, 40.0))
        hard = bool(rng.integers(0, 2))
        pts.append((bps, float(sigma), float(w0), first, n_sym, 1 if hard else 0))
        row = np.asarray(oracle.link_ber(n, cp, bps, seed, first, n_sym, sigma, w0), dtype=np.uint64).copy()
        if hard:
            row[5] = row[6] = 0
        want_rows.append(row)
        first += n_sym + int(rng.integers(0, 5))
    got = ctx.link_sweep(n, cp, seed, pts).cpu().numpy().astype(np.uint64)
    if not np.array_equal(got, np.stack(want_rows)):
        note("link_sweep", n=n, seed=seed, pts=pts)
ctx.close()
print(f"done: {count} link_ber + {count} link_generate cases, {max(1, count // 10)} sweeps, {bad} mismatches")
sys.exit(1 if bad else 0)
