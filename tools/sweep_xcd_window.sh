#!/bin/bash
# window of the XCD-aware workgroup-id permutation (0 = natural order; clipped to the grid) against
# rate, on large and on small batches
set -e
for W in 0 4096 32768 131072 1073741824; do
  echo "WINDOW=$W"
  SDR_XCD_WINDOW=$W timeout -k 10 200 python -u tools/kbench.py --cases 1024:4 --symbols 1048576 --repeat 3 2>&1 | grep median
  SDR_XCD_WINDOW=$W timeout -k 10 200 python -u tools/kbench.py --cases 2048:6 --symbols 524288 --repeat 3 2>&1 | grep median
  SDR_XCD_WINDOW=$W timeout -k 10 200 python -u tools/kbench.py --cases 4096:8 --symbols 131072 --repeat 3 2>&1 | grep median
  SDR_XCD_WINDOW=$W timeout -k 10 200 python -u tools/kbench.py --cases 1024:4 2048:6 1024:6 --symbols 32768 --repeat 3 2>&1 | grep median
done
