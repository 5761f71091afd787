#!/usr/bin/env python3
"""Prints the figures DESIGN.md section 7 quotes from the round's committed profiles (profiles/r3_*)."""
import csv
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
BYTES = {"10, 2, 2, false, false": (24576, 1 << 20), "10, 3, 3, false, false": (32768, 1 << 20), "10, 3, 3, false, true": (32768, 1 << 20),
         "11, 3, 3, false, true": (65536, None), "12, 4, 4, false, false": (163840, 1 << 17), "6, 1, 1, false, false": (1024, 1 << 22)}


def line(name):
    d = json.load(open(os.path.join(P, name)))
    ex = d.get("extra", {})
    print(f"{name}: value {d['value']:.4g} {d['unit']}, ms/step {d['ms_per_step']:.3f}, roofline.frac {d['roofline']['frac']:.4f}, traffic on this build: {d['roofline'].get('traffic_measured_on_this_build')}, "
          f"cpu {(d.get('cpu_baseline') or {}).get('value')}")
    for k, v in ex.items():
        if isinstance(v, dict):
            keep = {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in v.items() if kk in ("hbm_frac", "sym_per_s", "bits_per_s", "kernel_over_copy", "seconds_per_pass", "seconds_per_sweep", "value")}
            pl = v.get("per_launch")
            if pl:
                keep["frac best/median/worst"] = (round(pl["hbm_frac_best"], 4), round(pl["hbm_frac_median"], 4), round(pl["hbm_frac_worst"], 4))
            print("   ", k, keep)


def main():
    for n in ("r3_bench_line.json", "r3_bench_line_under_rocprof.json"):
        if os.path.exists(os.path.join(P, n)):
            line(n)
    print("rocprofv3 kernel stats of the default bench (profiles/r3_bench_kernel_stats.csv):")
    for r in csv.DictReader(open(os.path.join(P, "r3_bench_kernel_stats.csv"))):
        m = re.search(r"rx_fused_kernel<([^>]*)>", r["Name"])
        if not m:
            continue
        key = ", ".join(m.group(1).split(", ")[:5])
        if key in BYTES and BYTES[key][1]:
            b, n = BYTES[key]
            print(f"    rx_fused_kernel<{m.group(1)}>: {r['Calls']} launches, average {float(r['AverageNs']) / 1e6:.4f} ms -> {b * n / float(r['AverageNs']) / 8000:.4f} of 8 TB/s")
        else:
            print(f"    rx_fused_kernel<{m.group(1)}>: {r['Calls']} launches, average {float(r['AverageNs']) / 1e6:.4f} ms")
    t = json.load(open(os.path.join(P, "r3_traffic.json")))
    for k, v in t["kernels"].items():
        print(f"    traffic {k}: {v['hbm_bytes_per_symbol']:.1f} B/symbol, x{v['traffic_over_algorithmic']:.4f}")
    for n in ("r3_ber_sweep.json", "r3_ber_sweep_hard_only.json"):
        s = json.load(open(os.path.join(P, n)))
        print(f"{n}: {s['seconds_per_sweep'] * 1e3:.2f} ms, {s['bits_per_s']:.3g} bits/s, fixed overhead {s['fixed_overhead_us']:.1f} us, roofline.frac {s['roofline']['frac']:.3f}, issue utilisation {s['valu_issue_utilisation']['frac']:.3f}")


if __name__ == "__main__":
    main()
