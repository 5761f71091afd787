cd "${GRAFT_REPO_ROOT:-$PWD}"
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread -k "chain or generate or link or channel" > gpurun_out/t_r3f.log 2>&1 || { tail -30 gpurun_out/t_r3f.log; exit 1; }
tail -2 gpurun_out/t_r3f.log
echo "== chain"
timeout -k 10 200 python bench.py --chain fft2048_64qam_eq_4M_chain | cut -c1-520
bash tools/r3_chain_prof.sh
