#!/usr/bin/env python3
"""Instruction histogram per kernel of a gfx950 assembly listing (hipcc -S --cuda-device-only).

usage: isa_hist.py file.s [substring-of-kernel-name ...]
Static counts (each instruction once, loops not weighted): what the straight-line receive / link
kernels execute per wave, to first order.
"""
import collections
import re
import subprocess
import sys


def demangle(name):
    try:
        return subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip() or name
    except OSError:
        return name


def main():
    path, pats = sys.argv[1], sys.argv[2:]
    cur, hist = None, {}
    for line in open(path):
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", line)
        if m:
            cur = m.group(1)
            hist[cur] = collections.Counter()
            continue
        if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
            cur = None
        if cur is None:
            continue
        m = re.match(r"^\t([a-z_0-9]+)\s", line)
        if m and not m.group(1).startswith("."):
            hist[cur][m.group(1)] += 1
    for k, h in hist.items():
        d = demangle(k)
        if pats and not all(p in d or p in k for p in pats):
            continue
        tot = sum(h.values())
        valu = sum(v for i, v in h.items() if i.startswith("v_"))
        pk = sum(v for i, v in h.items() if i.startswith("v_pk_"))
        print(f"== {d}\n   total {tot}  VALU {valu}  packed {pk}  SALU {sum(v for i, v in h.items() if i.startswith('s_'))}  DS {sum(v for i, v in h.items() if i.startswith('ds_'))}")
        print("   " + "  ".join(f"{i}:{v}" for i, v in h.most_common(28)))


if __name__ == "__main__":
    main()
