cd "${GRAFT_REPO_ROOT:-$PWD}"
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/t_r3g.log 2>&1 || { tail -30 gpurun_out/t_r3g.log; exit 1; }
tail -2 gpurun_out/t_r3g.log
echo "== A = LLRs + counters as an instantiation of its own (outputs fixed at compile time), B = every-output kernel with run-time tests"
timeout -k 10 600 python -u tools/abbench.py sdr_amd/lib/libsdr_amd.so sdr_amd/lib_base/libsdr_amd.so --rounds 2 --cases 64:2:cnt 64:6:cnt 256:6:cnt 512:4:cnt 1024:2:cnt 1024:4:cnt 1024:6:cnt 1024:8:cnt 1024:10:cnt 2048:6:cnt 2048:6:eq:cnt 1024:6:eq:cnt 4096:8:cnt 4096:8:eq:cnt 8192:6:cnt 8192:6:eq:cnt || exit 1
echo "== chain"
timeout -k 10 200 python bench.py --chain fft2048_64qam_eq_4M_chain | cut -c1-520
