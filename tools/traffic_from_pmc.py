#!/usr/bin/env python3
"""Fold two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; counter_collection.csv each) into
profiles/r1_traffic.json, the per-launch HBM traffic bench.py reports as roofline.traffic.

  python tools/traffic_from_pmc.py profiles/r1_pmc_fetch_size.csv profiles/r1_pmc_write_size.csv > profiles/r1_traffic.json

Units and the gfx950 correction follow MI355X_MICROARCH.md (HBM / rocprofv3): both counters are in
KiB and FETCH_SIZE counts half the bytes of wide coalesced reads, so
bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.
"""
import csv
import json
import sys

WORKLOADS = {"fft1024_16qam_1M": "rx_fused_kernel<10, 2, 2, false, false, false>", "fft1024_64qam_1M": "rx_fused_kernel<10, 3, 3, false, false, false>"}
SYMBOLS = 1 << 20


def mean_counter(path, counter, kernel):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter and kernel in r["Kernel_Name"]]
    if not vals:
        raise SystemExit(f"{path}: no {counter} rows for {kernel}")
    return sum(vals) / len(vals)


def main():
    fetch_csv, write_csv = sys.argv[1], sys.argv[2]
    out = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python3 bench.py --steps 2 --warmup 1 --no-cpu` "
                   "(1 Mi symbols per launch). Units: KiB. gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts half the bytes "
                   "of wide coalesced reads, so bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024.", "kernels": {}}
    for name, kernel in WORKLOADS.items():
        f = mean_counter(fetch_csv, "FETCH_SIZE", kernel)
        w = mean_counter(write_csv, "WRITE_SIZE", kernel)
        total = (2 * f + w) * 1024
        out["kernels"][name] = {"kernel": "void sdr::" + kernel, "FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "hbm_bytes_per_launch": total,
                                "symbols_per_launch": SYMBOLS, "hbm_bytes_per_symbol": total / SYMBOLS}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
