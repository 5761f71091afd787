#!/bin/bash
# A/B of two library builds on the same box, interleaved: tools/ab.sh <libA> <libB> <cases...>
set -e
A=$1; B=$2; shift 2
for round in 1 2; do
  for L in "$A" "$B"; do
    echo "== $L"
    SDR_AMD_LIB=$PWD/$L timeout -k 10 200 python -u tools/kbench.py --cases "$@" --steps 20 --warmup 8 --repeat 3 2>&1 | grep median
  done
done
