#!/bin/bash
# run-length sweep of the tap-holding variants (symbol groups per workgroup)
set -e
for G in 1 2 3 4 7 14; do
  echo "G=$G"
  SDR_GROUPS_PER_WG=$G timeout -k 10 150 python -u tools/kbench.py --cases 1024:4:eq 2048:6:eq 4096:8:eq --steps 8 --warmup 2 --repeat 3 2>&1 | grep median
done
