#!/bin/bash
# tests/test_gpu_parity.py::test_hard_decisions_on_boundaries_through_the_fused_kernels (19 shapes) from many seeds
cd "${GRAFT_REPO_ROOT:-$PWD}"
for s in $(seq ${1:-1} ${2:-40}); do
  SDR_SOAK_SEED=$s timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -k test_hard_decisions_on_boundaries -x -q --timeout 120 --timeout-method thread > gpurun_out/soak_boundaries.log 2>&1 || { echo "seed $s failed"; tail -20 gpurun_out/soak_boundaries.log; exit 1; }
  echo "seed $s: $(tail -1 gpurun_out/soak_boundaries.log)"
done
