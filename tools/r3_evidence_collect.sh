#!/bin/bash
# copies the summaries of tools/r3_evidence.sh from gpurun_out/evidence into profiles/ (run here, after the gpurun call)
cd "$(dirname "$0")/.."
E=gpurun_out/evidence
cp $E/prof/trace/bench_kernel_stats.csv profiles/r3_bench_kernel_stats.csv
cp $E/prof/bench_trace.json profiles/r3_bench_line_under_rocprof.json
cp $E/prof/fetch/fetch_counter_collection.csv profiles/r3_pmc_fetch_size.csv
cp $E/prof/write/write_counter_collection.csv profiles/r3_pmc_write_size.csv
python3 tools/traffic_json.py $E/prof profiles/r3_traffic.json
cp $E/bench_line.json profiles/r3_bench_line.json
cp $E/bench_line_default_placement.json profiles/r3_bench_line_default_placement.json
python3 - <<'PY'
import json
out = ["# python bench.py --workload <w> --no-cpu --no-extra --steps 25 in fresh processes, one after the other on one box (tools/r3_evidence.sh): roofline.frac of each",
       "# (buffers placed by sdr_placement_search_c32 in a 200 GiB arena, bench.py's default; *_default_placement: --placement default, plain allocations)"]
for name in ("fresh_process_eq", "fresh_process_64qam", "fresh_process_2048_eq", "fresh_process_eq_default_placement"):
    rows = [json.loads(l) for l in open("gpurun_out/evidence/%s.jsonl" % name) if l.startswith("{")]
    out.append("%s (%s): %s" % (name, rows[0]["config"]["workload"], "  ".join("%.4f" % r["roofline"]["frac"] for r in rows)))
open("profiles/r3_fresh_process_runs.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
cp $E/ber_sweep.json profiles/r3_ber_sweep.json
grep -v amdgpu.ids $E/ber_sweep.txt > profiles/r3_ber_sweep_table.txt
cp $E/ber_sweep_hard_only.json profiles/r3_ber_sweep_hard_only.json
(echo "# tools/kbench.py --repeat 3 (8 untimed + 20 timed launches per repeat, fresh buffers, bins of unit average energy), kernels at the end of round 3; case = n_fft:bps[:eq][:hard][:cnt][:nollr][:gen][:mc], bps 0 = bins out, fft = in-place fft2; gen: bytes = samples written; mc: LLR-equivalent bytes (nothing touches HBM)"; grep "median" $E/kernel_table.txt) > profiles/r3_kernel_table.txt
grep -v amdgpu.ids $E/rate_matrices.txt > profiles/r3_rate_matrices.txt
grep -v amdgpu.ids $E/rate_matrices_placed.txt > profiles/r3_rate_matrices_placed.txt
ls -la profiles/r3_*
