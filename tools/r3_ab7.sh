cd "${GRAFT_REPO_ROOT:-$PWD}"
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/t_r3d.log 2>&1 || { tail -30 gpurun_out/t_r3d.log; exit 1; }
tail -2 gpurun_out/t_r3d.log
echo "== A = LLR magnitude statistic as packed 16-bit conversion + dot product, B = trunc(min(|L| * 1024, 2^30)) (previous build)"
timeout -k 10 300 python -u tools/abbench.py sdr_amd/lib/libsdr_amd.so sdr_amd/lib_base/libsdr_amd.so --cases 1024:4:hard:cnt 1024:6:hard:cnt 1024:8:hard:cnt 2048:6:hard:cnt 2048:6:eq:hard:cnt 2048:6:eq:cnt 512:4:hard:cnt 4096:8:hard:cnt 64:2:hard:cnt
echo "== generator, Monte-Carlo link (kbench, fresh buffers)"
timeout -k 10 300 python -u tools/kbench.py --cases 1024:6:gen 2048:6:gen 1024:6:mc 2048:6:mc 1024:2:mc --repeat 3 2>&1 | grep -i median
echo "== chain"
timeout -k 10 200 python bench.py --chain fft2048_64qam_eq_4M_chain | cut -c1-600
echo "== sweep"
timeout -k 10 200 python bench.py --sweep | cut -c1-500
