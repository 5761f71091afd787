#!/usr/bin/env python3
"""Rate matrices of the receive kernel (profiles/r2_rate_matrices.txt): runs tools/kbench.py over n_fft x bits per
subcarrier for the LLR-only, tap-reading, every-output and hard-only variants and prints % of 8 TB/s on SURVEY 8(d) bytes.
usage: python3 tools/ratematrix.py > profiles/rN_rate_matrices.txt"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIZES = [64, 128, 256, 512, 1024, 2048, 4096, 8192]
FEW = [64, 256, 1024, 2048, 4096, 8192]


PLACED = "--placed" in sys.argv[1:]


def run(cases):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kbench.py"), *(["--placed"] if PLACED else []), "--cases", *cases], capture_output=True, text=True, check=True).stdout
    res = {}
    for m in re.finditer(r"case\s+(\S+)\s+n_sym\s+\d+\s+[\d.]+ ms/launch\s+([\d.]+) Msym/s\s+[\d.]+ GB/s\s+([\d.]+) %", out):
        res[m.group(1)] = (float(m.group(3)), float(m.group(2)))
    return res


def matrix(title, sizes, mods, suffix, rates=False):
    res = run([f"{n}:{b}{suffix}" for n in sizes for b in mods])
    print(title)
    w = 16 if rates else 7
    print("N\\bps " + "".join(f"{b:>{w}d}" for b in mods))
    for n in sizes:
        cells = []
        for b in mods:
            pct, msym = res[f"{n}:{b}{suffix}"]
            cells.append(f"{pct:7.1f}({msym:7.1f})" if rates else f"{pct:7.1f}")
        print(f"{n:6d}" + "".join(cells))
    print(flush=True)


def main():
    if PLACED:
        print("# tools/ratematrix.py --placed: as below, but the sample and LLR buffers of every cell are views of one 180 GiB arena at the offsets sdr_placement_search_c32 picks (DESIGN.md 5.1).")
        print("# The hard-bits-only variant and ofdm::rx keep plain allocations (the first does not depend on placement, tools/poolhard.py).")
    print("# tools/ratematrix.py (tools/kbench.py, one sample per cell: 20 timed launches after 8 warm-up launches, fresh buffers per cell, one box), % of 8 TB/s on SURVEY 8(d) bytes")
    print("# rows: n_fft (cp = n_fft / 4); columns: bits per subcarrier (1 = BPSK ... 10 = 1024-QAM).  Cells move by +-2 to 3 points with the box and the placement of the buffers.\n")
    matrix("fused CP removal + FFT + max-log LLR", SIZES, [1, 2, 4, 6, 8, 10], "")
    matrix("fused CP removal + FFT + max-log LLR WITH per-subcarrier taps and weights (14 symbols per set)", SIZES, [1, 2, 4, 6, 8, 10], ":eq")
    matrix("LLRs + packed hard bits (every-output variant), % of 8 TB/s (Msym/s)", FEW, [2, 4, 6, 8, 10], ":hard", rates=True)
    matrix("LLRs + packed hard bits WITH taps and weights (every-output variant), % of 8 TB/s (Msym/s)", FEW, [2, 4, 6, 8, 10], ":eq:hard", rates=True)
    matrix("packed hard bits ONLY (the reference's ofdm::rx -> from_constl chain; VALU bound), % of 8 TB/s (Msym/s)", FEW, [2, 4, 6, 8, 10], ":hard:nollr", rates=True)
    res = run([f"{n}:0" for n in SIZES])
    print("ofdm::rx batch (CP removal + FFT, bins out)")
    print("     N " + "".join(f"{n:7d}" for n in SIZES))
    print("       " + "".join(f"{res[f'{n}:0'][0]:7.1f}" for n in SIZES))


if __name__ == "__main__":
    main()
