#!/bin/bash
# PMC instruction counts of the Monte-Carlo bodies at n_fft = 1024 (all six modulations, full and hard-only): the inputs of
# ber_sweep.py's issue-cost tables.  gpurun; summaries in gpurun_out/mc_pmc/*.txt
cd "${GRAFT_REPO_ROOT:-$PWD}"
mkdir -p gpurun_out/mc_pmc
bash tools/pmc_kernels.sh mc_full 65536 1024:1:mc 1024:2:mc 1024:4:mc 1024:6:mc 1024:8:mc 1024:10:mc > gpurun_out/mc_pmc/full.log 2>&1 || { echo "pmc full failed"; tail -5 gpurun_out/mc_pmc/full.log; exit 1; }
python3 tools/pmc_summary.py gpurun_out/pmc_mc_full > gpurun_out/mc_pmc/full.txt
for b in 1 2 4 6 8 10; do
  bash tools/pmc_kernels.sh mc_hard_$b 65536 1024:$b:mch > gpurun_out/mc_pmc/hard_$b.log 2>&1 || { echo "pmc hard $b failed"; exit 1; }
  (echo "######## 1024:$b:mch (sweep_fused_kernel<10>, every point SDR_POINT_HARD_ONLY)"; python3 tools/pmc_summary.py gpurun_out/pmc_mc_hard_$b) >> gpurun_out/mc_pmc/hard.txt
done
echo done
