#!/bin/bash
# Builds a VARIANT of libsdr_amd.so for same-buffer A/Bs (tools/abbench.py): the fused receive (or link) instantiation
# units of the named sizes are recompiled with extra flags and linked with the shipped objects of sdr_amd/lib.
#   tools/variant.sh <name> "<extra flags>" [rx:<log2n> | link:<log2n> ...]      -> sdr_amd/lib_<name>/libsdr_amd.so
set -e
cd "$(dirname "$0")/../sdr_amd/csrc"
NAME=$1; EXTRA=$2; shift 2
OUT=../lib_$NAME
mkdir -p $OUT
FLAGS="-O3 -std=c++20 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -I../../include -I. -Wall -Wno-unused-function"
declare -A REPLACED
pids=()
for u in "$@"; do
    kind=${u%%:*}; sz=${u##*:}
    src=rx_fused_inst.hip; obj=rx_fused_inst_$sz.o
    if [ "$kind" = link ]; then src=link_fused_inst.hip; obj=link_fused_inst_$sz.o; fi
    /opt/rocm/bin/hipcc $FLAGS $EXTRA -DSDR_LOG2N=$sz -c $src -o $OUT/$obj &
    pids+=($!)
    REPLACED[$obj]=1
done
OBJS=""
for o in ../lib/*.o; do
    b=$(basename $o)
    if [ -n "${REPLACED[$b]}" ]; then OBJS="$OBJS $OUT/$b"; else OBJS="$OBJS $o"; fi
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc -shared --offload-arch=gfx950 -o $OUT/libsdr_amd.so $OBJS
ls -la $OUT/libsdr_amd.so
