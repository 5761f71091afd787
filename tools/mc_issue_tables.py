#!/usr/bin/env python3
"""ber_sweep.py's issue-cost tables from the PMC summaries of tools/r3_mc_pmc.sh (gpurun_out/mc_pmc/full.txt, hard.txt).
Per OFDM symbol at n_fft = 1024 (one wave per symbol, 65536 symbols per dispatch):
  lane instructions = 64 x SQ_INSTS_VALU / symbols
  issue cycles      = (2.15 x (VALU - INT64 - TRANS) + 4.6 x INT64 + 10 x TRANS) / symbols
with the per-class issue costs tools/valubench measured (profiles/r2_valubench_issue_cost.txt); VOP3-encoded single-word
instructions (v_fma, v_med3, compares into scalar pairs: 4.1-4.6 cycles) cannot be told apart by the counters and are
priced as VOP2, so the cycle figure is a lower bound."""
import re
import sys

SYMBOLS = 65536
MODS = {"1,0": 1, "1,1": 2, "2,2": 4, "3,3": 6, "4,4": 8, "5,5": 10}


def blocks(path, header):
    cur, out = None, {}
    for line in open(path):
        m = re.match(header, line)
        if m:
            cur = m.group(1)
            out[cur] = {}
            continue
        m = re.match(r"\s+(SQ_\w+)\s+([\d.]+)\s*$", line)
        if m and cur is not None:
            out[cur].setdefault(m.group(1), float(m.group(2)))
    return out


def figures(c):
    valu, i64, trans = c["SQ_INSTS_VALU"], c["SQ_INSTS_VALU_INT64"], c["SQ_INSTS_VALU_TRANS_F32"]
    return round(64 * valu / SYMBOLS), round((2.15 * (valu - i64 - trans) + 4.6 * i64 + 10 * trans) / SYMBOLS)


def main():
    root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/mc_pmc"
    full = {MODS[k]: figures(v) for k, v in blocks(root + "/full.txt", r"== mc_fused_kernel<10,(\d,\d)>").items()}
    hard = {int(k): figures(v) for k, v in blocks(root + "/hard.txt", r"######## 1024:(\d+):mch").items()}
    for name, tab, idx in (("VALU_ISSUE_CYCLES_PER_SYMBOL_1024", full, 1), ("VALU_LANE_INSTR_PER_SYMBOL_1024", full, 0),
                           ("VALU_ISSUE_CYCLES_PER_SYMBOL_1024_HARD", hard, 1), ("VALU_LANE_INSTR_PER_SYMBOL_1024_HARD", hard, 0)):
        print(f"{name} = {{" + ", ".join(f"{b}: {tab[b][idx]}" for b in sorted(tab)) + "}")


if __name__ == "__main__":
    main()
