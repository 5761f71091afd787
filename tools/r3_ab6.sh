cd "${GRAFT_REPO_ROOT:-$PWD}"
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/t_r3c.log 2>&1 || { tail -30 gpurun_out/t_r3c.log; exit 1; }
tail -2 gpurun_out/t_r3c.log
echo "== generator, Monte-Carlo link and counting demodulator (kbench, fresh buffers)"
timeout -k 10 300 python -u tools/kbench.py --cases 1024:6:gen 2048:6:gen 1024:6:mc 2048:6:mc 1024:2:mc 2048:6:eq:hard:cnt 2048:6:eq:cnt --repeat 3 2>&1 | grep -i median
echo "== chain"
timeout -k 10 200 python bench.py --chain fft2048_64qam_eq_4M_chain
echo "== sweep"
timeout -k 10 200 python bench.py --sweep
