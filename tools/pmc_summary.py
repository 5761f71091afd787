#!/usr/bin/env python3
"""Folds the passes of tools/pmc_kernels.sh into one table per kernel (mean per dispatch):

  python tools/pmc_summary.py gpurun_out/pmc_<tag> > profiles/r2_pmc_<tag>.txt

Derived columns (MI355X_MICROARCH.md: SQ_* cycle counters are in quad-cycles and summed over waves):
  valu/lds/wait shares = SQ_ACTIVE_INST_VALU, SQ_ACTIVE_INST_LDS, SQ_WAIT_ANY, SQ_WAIT_INST_ANY over SQ_WAVE_CYCLES;
  lane-instructions per unit = SQ_INSTS_VALU * 64 / units; L2 hit rate = TCC_HIT / (TCC_HIT + TCC_MISS);
  LDS conflict share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"(\w+_kernel)<([^>]*)>", name)
    return f"{m.group(1)}<{m.group(2).replace(' ', '')}>" if m else name[:60]


def main():
    root = sys.argv[1]
    vals = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(os.path.join(root, "p*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            vals[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    regs = {}
    dur = defaultdict(list)
    for path in glob.glob(os.path.join(root, "trace", "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            k = short(r["Kernel_Name"])
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
            regs[k] = (r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"), r.get("Workgroup_Size"), r.get("Grid_Size"))
    for k in sorted(vals):
        c = {n: sum(v) / len(v) for n, v in vals[k].items()}
        if not any(t in k for t in ("rx_fused", "tx_fused", "mc_fused", "sweep", "demap", "constl", "nearest")):
            continue
        print(f"== {k}")
        if k in regs:
            d = sorted(dur[k])
            print(f"   trace: median {d[len(d) // 2]:.1f} us over {len(d)} dispatches; VGPR {regs[k][0]} AGPR {regs[k][1]} SGPR {regs[k][2]} LDS {regs[k][3]} B, workgroup {regs[k][4]}, grid {regs[k][5]}")
        for n in sorted(c):
            print(f"   {n:32s} {c[n]:18.1f}")
        wc = c.get("SQ_WAVE_CYCLES")
        if wc:
            parts = []
            for n, label in (("SQ_ACTIVE_INST_VALU", "VALU active"), ("SQ_ACTIVE_INST_LDS", "LDS active"), ("SQ_ACTIVE_INST_VMEM", "VMEM active"), ("SQ_ACTIVE_INST_SCA", "scalar active"),
                             ("SQ_ACTIVE_INST_ANY", "any active"), ("SQ_WAIT_ANY", "waiting (waitcnt/barrier)"), ("SQ_WAIT_INST_ANY", "issue stall"), ("SQ_WAIT_INST_LDS", "LDS issue stall")):
                if n in c:
                    parts.append(f"{label} {100 * c[n] / wc:.1f} %")
            print("   share of wave-cycles: " + ", ".join(parts))
        if "SQ_BUSY_CYCLES" in c and "SQ_ACTIVE_INST_VALU" in c:
            # per SIMD: a wave64 VALU instruction occupies the SIMD for one quad-cycle... reported as is
            pass
        if "SQ_WAVES" in c and c["SQ_WAVES"]:
            w = c["SQ_WAVES"]
            per = [f"{n[9:]} {c[n] / w:.1f}" for n in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM", "SQ_INSTS_VALU_TRANS_F32") if n in c]
            print(f"   per wave: " + ", ".join(per) + (f", wave-cycles x4 {4 * wc / w:.0f}" if wc else ""))
        if "TCC_HIT_sum" in c:
            h, m = c["TCC_HIT_sum"], c.get("TCC_MISS_sum", 0.0)
            print(f"   L2 hit rate {100 * h / max(h + m, 1):.1f} % ({h:.0f} hits, {m:.0f} misses)")
        if c.get("SQ_LDS_IDX_ACTIVE"):
            print(f"   LDS bank-conflict cycles / LDS active cycles {100 * c.get('SQ_LDS_BANK_CONFLICT', 0) / c['SQ_LDS_IDX_ACTIVE']:.1f} %")


if __name__ == "__main__":
    main()
