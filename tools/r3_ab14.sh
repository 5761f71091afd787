cd "${GRAFT_REPO_ROOT:-$PWD}"
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/t_r3h.log 2>&1 || { tail -30 gpurun_out/t_r3h.log; exit 1; }
tail -2 gpurun_out/t_r3h.log
timeout -k 10 300 python -u tools/abbench.py sdr_amd/lib/libsdr_amd.so sdr_amd/lib_base/libsdr_amd.so --rounds 3 --cases 64:2 128:2 64:4 128:4 256:2 || exit 1
