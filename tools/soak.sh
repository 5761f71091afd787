#!/bin/bash
# Longer seeded sweeps of the parity checks than the test suite runs by default (gpurun): the receive path over 1500 random
# configurations from another seed, the chunked link over 150 random shapes in both forms.
cd "${GRAFT_REPO_ROOT:-$PWD}"
SDR_SOAK_COUNT=${2:-1500} SDR_SOAK_SEED=${1:-7} timeout -k 10 900 python -u -m pytest tests/test_gpu_parity.py -k test_demod_random_configurations -x -q --timeout 120 --timeout-method thread > gpurun_out/soak_demod.log 2>&1 || { echo "demod soak failed"; tail -20 gpurun_out/soak_demod.log; exit 1; }
tail -1 gpurun_out/soak_demod.log
timeout -k 10 900 python -u tools/soak_chain.py ${3:-150} ${1:-7} > gpurun_out/soak_chain.log 2>&1 || { echo "chain soak failed"; tail -20 gpurun_out/soak_chain.log; exit 1; }
tail -1 gpurun_out/soak_chain.log
