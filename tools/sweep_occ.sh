#!/bin/bash
# Occupancy sweep of the receive kernels: SDR_LDS_PAD (unused dynamic LDS per workgroup) throttles the
# resident workgroups per CU; SDR_DEBUG prints what each instantiation got.
#   tools/sweep_occ.sh <lib dir under sdr_amd/> "<pads>" <kbench cases...>
L=$1; PADS=$2; shift 2
for P in $PADS; do
  echo "== $L PAD=$P"
  SDR_DEBUG=1 SDR_LDS_PAD=$P SDR_AMD_LIB=$PWD/sdr_amd/$L/libsdr_amd.so timeout -k 10 300 python -u tools/kbench.py --cases "$@" --repeat 3 2>&1 | grep -E "median|WG/CU" | sort -u
done
