cd "${GRAFT_REPO_ROOT:-$PWD}"
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/t_r3b.log 2>&1 || { tail -30 gpurun_out/t_r3b.log; exit 1; }
tail -2 gpurun_out/t_r3b.log
H="1024:4:hard:nollr 1024:6:hard:nollr 1024:8:hard:nollr 2048:6:hard:nollr 4096:8:hard:nollr 512:4:hard:nollr 256:6:hard:nollr 64:6:hard:nollr 1024:6:hard 2048:6:hard 1024:4:hard:cnt 2048:6:eq:hard:cnt 1024:4 1024:6:eq"
echo "== A = new slicer (reach test, half coordinates, hard-only block with deferred fix-up, unit twiddles), B = base"
timeout -k 10 300 python -u tools/abbench.py sdr_amd/lib/libsdr_amd.so sdr_amd/lib_base/libsdr_amd.so --cases $H
