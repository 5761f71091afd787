#!/bin/bash
# HBM traffic of kbench cases: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes (gfx950: FETCH_SIZE
# counts half the bytes of wide coalesced reads; both in KiB).  tools/pmc_traffic.sh <tag> <cases...>
set -e
cd "${GRAFT_REPO_ROOT:-$PWD}"
export TMPDIR=/tmp
TAG=$1; shift
OUT=gpurun_out/traffic_$TAG
rm -rf $OUT && mkdir -p $OUT
RUN="python3 tools/kbench.py --steps 2 --warmup 1 --cases $*"
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o c -- $RUN > $OUT/fetch.out 2> $OUT/fetch.err
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o c -- $RUN > $OUT/write.out 2> $OUT/write.err
python3 - "$OUT" <<'PY'
import csv, glob, sys, re
from collections import defaultdict
root = sys.argv[1]
v = defaultdict(lambda: defaultdict(list))
for p in glob.glob(root + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        m = re.search(r"(\w+_kernel)<([^>]*)>", r["Kernel_Name"])
        if m:
            v[m.group(1) + "<" + m.group(2).replace(" ", "") + ">"][r["Counter_Name"]].append((float(r["Counter_Value"]), int(r["Grid_Size"])))
for k in sorted(v):
    f = v[k].get("FETCH_SIZE"); w = v[k].get("WRITE_SIZE")
    if f and w:
        fb = 2 * 1024 * sum(x for x, _ in f) / len(f); wb = 1024 * sum(x for x, _ in w) / len(w)
        print(f"{k}: grid {f[0][1]} threads, read {fb / 1e9:.3f} GB (2 x FETCH_SIZE), written {wb / 1e9:.3f} GB per launch")
PY
