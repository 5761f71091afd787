cd "${GRAFT_REPO_ROOT:-$PWD}"
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/t_r3i.log 2>&1 || { tail -30 gpurun_out/t_r3i.log; exit 1; }
tail -2 gpurun_out/t_r3i.log
echo "== element-wise kernels, this build"
timeout -k 10 300 python -u tools/elembench.py 2>&1 | grep "from_constl\|nearest"
echo "== previous build"
SDR_AMD_LIB=sdr_amd/lib_base/libsdr_amd.so timeout -k 10 300 python -u tools/elembench.py 2>&1 | grep "from_constl\|nearest"
