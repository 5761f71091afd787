#!/usr/bin/env python3
"""Fold the FETCH_SIZE / WRITE_SIZE passes of tools/profile_bench.sh into profiles/<tag>_traffic.json.

usage: traffic_json.py gpurun_out/prof profiles/r2_traffic.json
Per receive kernel of the default bench run: median counter value per dispatch (KiB), gfx950 correction
(MI355X_MICROARCH.md, HBM section: FETCH_SIZE counts half the bytes of wide coalesced reads), bytes per
launch and per symbol.  bench.py reads the result for roofline.traffic.
"""
import csv
import glob
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS, kernel_source_fingerprint  # noqa: E402

MODS = {1: (1, 0), 2: (1, 1), 4: (2, 2), 6: (3, 3), 8: (4, 4), 10: (5, 5)}


def kernel_of(name):
    n_fft, _, bps, _, _, with_eq = WORKLOADS[name]
    mi, mq = MODS[bps]
    return f"{n_fft.bit_length() - 1},{mi},{mq},false,{'true' if with_eq else 'false'},false,false,false"      # <LOG2N, MI, MQ, FULL, EQ, HARDONLY, NOWGT, LC>


def collect(root, counter):
    vals = {}
    for p in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(p)):
            if r["Counter_Name"] != counter:
                continue
            m = re.search(r"rx_fused_kernel<([^>]*)>", r["Kernel_Name"])
            if m:
                vals.setdefault(m.group(1).replace(" ", ""), []).append(float(r["Counter_Value"]))
    return vals


def main():
    root, out = sys.argv[1], sys.argv[2]
    fetch, write = collect(root, "FETCH_SIZE"), collect(root, "WRITE_SIZE")
    kernels = {}
    for name, (n_fft, cp, bps, n_sym, _, with_eq) in WORKLOADS.items():
        k = kernel_of(name)
        if k not in fetch or k not in write:
            continue
        # median, not mean: the same instantiation also serves other launches of the run (the 4 Mi-symbol pass ends with a chunk of
        # 128 symbols), and the workload's own dispatches are the majority
        f = sorted(fetch[k])[len(fetch[k]) // 2]
        w = sorted(write[k])[len(write[k]) // 2]
        total = (2.0 * f + w) * 1024.0
        alg = (8 * n_fft + 4 * n_fft * bps) * n_sym
        kernels[name] = {"kernel": f"sdr::rx_fused_kernel<{k}>", "dispatches": [len(fetch[k]), len(write[k])], "FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w,
                         "hbm_bytes_per_launch": total, "symbols_per_launch": n_sym, "hbm_bytes_per_symbol": total / n_sym,
                         "algorithmic_bytes_per_symbol": alg / n_sym, "traffic_over_algorithmic": total / alg}
    note = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python3 bench.py --steps 2 --warmup 1 --no-cpu` (tools/profile_bench.sh). "
            "Units: KiB. gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts half the bytes of wide coalesced reads, so bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024. "
            "Workloads with the equaliser read 12 bytes of taps and weights per bin and 14-symbol set on top of the algorithmic bytes.")
    json.dump({"note": note, "kernel_source_fingerprint": kernel_source_fingerprint(), "kernels": kernels}, open(out, "w"), indent=1)
    for name, k in kernels.items():
        print(f"{name:22s} {k['hbm_bytes_per_symbol']:10.1f} B/symbol measured, {k['algorithmic_bytes_per_symbol']:8.0f} algorithmic, x{k['traffic_over_algorithmic']:.4f}")


if __name__ == "__main__":
    main()
