#!/usr/bin/env python3
"""PCIe-inclusive throughput of the streaming ring (sdr_pipe_*) against the one-shot _host call.

  python tools/pipebench.py --n-fft 1024 --mod 4 --chunk 8192 --chunks 24 --depth 3
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n-fft", type=int, default=1024)
    ap.add_argument("--mod", type=int, default=4)
    ap.add_argument("--chunk", type=int, default=8192)
    ap.add_argument("--chunks", type=int, default=24)
    ap.add_argument("--depth", type=int, default=3)
    args = ap.parse_args()
    from sdr_amd.pipe import Pipe
    n, cp, bps = args.n_fft, args.n_fft // 4, args.mod
    rng = np.random.default_rng(1)
    src = (rng.standard_normal((args.chunk, (n + cp) * 2), dtype=np.float32) * np.float32(0.05)).view(np.complex64)
    for label, kw in [("hard bits to host", dict(hard_to_host=True)), ("LLRs to host", dict(llr_to_host=True)),
                      ("LLRs stay on device", dict(want_llr=True))]:
        with Pipe(0, n_fft=n, cp=cp, mod=bps, chunk_symbols=args.chunk, depth=args.depth, **kw) as pipe:
            def feed(count):
                done = 0
                for _ in range(count):
                    if pipe.in_flight() == args.depth:
                        pipe.collect(); pipe.release(); done += 1
                    pipe.acquire()          # the producer's own fill is not timed: the samples are already there
                    pipe.submit(args.chunk)
                while pipe.in_flight():
                    pipe.collect(); pipe.release(); done += 1
                return done
            pipe.acquire()[:] = src
            pipe.submit(args.chunk); pipe.collect(); pipe.release()
            for _ in range(args.depth - 1):          # touch every slot once
                pipe.acquire()[:] = src
                pipe.submit(args.chunk)
            while pipe.in_flight():
                pipe.collect(); pipe.release()
            t0 = time.perf_counter()
            feed(args.chunks)
            dt = time.perf_counter() - t0
            sym = args.chunk * args.chunks
            up = sym * (n + cp) * 8 / dt / 1e9
            print(f"{label:>22s}: {sym / dt / 1e6:8.3f} Msym/s  upload {up:6.1f} GB/s  ({args.chunks} chunks of {args.chunk}, depth {args.depth})", flush=True)


if __name__ == "__main__":
    main()
