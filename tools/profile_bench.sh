#!/bin/bash
# rocprofv3 passes over bench.py whose summaries go into profiles/ (run on the GPU box via gpurun):
#   1. kernel trace + stats of the default bench run (per-kernel average duration)
#   2./3. PMC passes FETCH_SIZE / WRITE_SIZE (separate runs, short bench)
set -e
cd "${GRAFT_REPO_ROOT:-$PWD}"
export TMPDIR=/tmp
OUT=gpurun_out/prof
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 bench.py --no-cpu > $OUT/bench_trace.json 2> $OUT/trace.err
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu > $OUT/bench_fetch.json 2> $OUT/fetch.err
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o write -- python3 bench.py --steps 2 --warmup 1 --no-cpu > $OUT/bench_write.json 2> $OUT/write.err
find $OUT -name "*.csv" | head -20
