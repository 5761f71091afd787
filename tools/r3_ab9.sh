cd "${GRAFT_REPO_ROOT:-$PWD}"
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread -k "taps or weights or equaliser or random_configurations or dispatch or no_alloc" > gpurun_out/t_r3e.log 2>&1 || { tail -30 gpurun_out/t_r3e.log; exit 1; }
tail -2 gpurun_out/t_r3e.log
echo "== A = LLR-only tap-reading kernels without run-time tests of their arrays, B = previous build"
C=""
for n in 64 256 512 1024 2048 4096; do for b in 2 4 6 8; do C="$C $n:$b:eq"; done; done
timeout -k 10 600 python -u tools/abbench.py sdr_amd/lib/libsdr_amd.so sdr_amd/lib_base/libsdr_amd.so --rounds 2 --cases $C 1024:6:taps 2048:6:taps 1024:6:wft 8192:6:eq 1024:10:eq || exit 1
