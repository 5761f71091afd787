#!/bin/bash
# resident workgroups per CU (LDS padding) against rate, two-wave workgroups (N = 2048)
set -e
for PAD in 0 6500 14000 28000; do
  echo "PAD=$PAD"
  SDR_LDS_PAD=$PAD timeout -k 10 150 python -u tools/kbench.py --cases 2048:6 2048:6:eq 4096:8 --steps 8 --warmup 2 --repeat 3 2>&1 | grep median
done
