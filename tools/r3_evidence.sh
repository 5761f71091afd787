#!/bin/bash
# Round 3 evidence on the GPU box (gpurun): rocprofv3 kernel stats + PMC traffic of the default bench, an unprofiled bench
# line, the Eb/N0 sweep (with its fixed overhead), the kernel table and the rate matrices.  Results land in
# gpurun_out/evidence; tools/r3_evidence_collect.sh copies the summaries into profiles/.
cd "${GRAFT_REPO_ROOT:-$PWD}"
export TMPDIR=/tmp
OUT=gpurun_out/evidence
rm -rf $OUT && mkdir -p $OUT
bash tools/profile_bench.sh > $OUT/profile_bench.log 2>&1 || { echo "profile_bench failed"; tail -5 $OUT/profile_bench.log; exit 1; }
cp -r gpurun_out/prof $OUT/prof
echo "profile passes done"
timeout -k 10 300 python bench.py > $OUT/bench_line.json 2> $OUT/bench_line.err || { echo "bench failed"; exit 1; }
echo "bench done"
# the north_star shape (and its plain counterpart) as the headline workload of five fresh processes each: the spread between processes
for i in 1 2 3 4 5; do
  timeout -k 10 120 python bench.py --workload fft1024_64qam_eq --no-cpu --no-extra --steps 25 >> $OUT/fresh_process_eq.jsonl 2>> $OUT/fresh_process.err || { echo "fresh-process bench failed"; exit 1; }
done
for i in 1 2 3; do
  timeout -k 10 120 python bench.py --workload fft1024_64qam_1M --no-cpu --no-extra --steps 25 >> $OUT/fresh_process_64qam.jsonl 2>> $OUT/fresh_process.err || { echo "fresh-process bench failed"; exit 1; }
  timeout -k 10 120 python bench.py --workload fft2048_64qam_eq --no-cpu --no-extra --steps 25 >> $OUT/fresh_process_2048_eq.jsonl 2>> $OUT/fresh_process.err || { echo "fresh-process bench failed"; exit 1; }
done
# the same with plain allocations (--placement default): what the allocator's draw gives
for i in 1 2 3; do
  timeout -k 10 120 python bench.py --workload fft1024_64qam_eq --placement default --no-cpu --no-extra --steps 25 >> $OUT/fresh_process_eq_default_placement.jsonl 2>> $OUT/fresh_process.err || { echo "fresh-process bench failed"; exit 1; }
done
timeout -k 10 300 python bench.py --placement default --no-cpu > $OUT/bench_line_default_placement.json 2>> $OUT/fresh_process.err || { echo "default-placement bench failed"; exit 1; }
echo "fresh processes done"
timeout -k 10 200 python ber_sweep.py > $OUT/ber_sweep.json 2> $OUT/ber_sweep.txt || { echo "ber_sweep failed"; exit 1; }
timeout -k 10 200 python ber_sweep.py --hard-only > $OUT/ber_sweep_hard_only.json 2> $OUT/ber_sweep_hard_only.txt || { echo "ber_sweep hard failed"; exit 1; }
echo "sweeps done"
timeout -k 10 900 python -u tools/kbench.py --repeat 3 --cases $(cat tools/r3_kernel_table_cases.txt) > $OUT/kernel_table.txt 2>&1 || { echo "kbench failed"; exit 1; }
echo "kernel table done"
timeout -k 10 900 python -u tools/ratematrix.py > $OUT/rate_matrices.txt 2> $OUT/rate_matrices.err || { echo "ratematrix failed"; exit 1; }
timeout -k 10 900 python -u tools/ratematrix.py --placed > $OUT/rate_matrices_placed.txt 2> $OUT/rate_matrices_placed.err || { echo "ratematrix --placed failed"; exit 1; }
echo "rate matrices done"
