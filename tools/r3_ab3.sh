set -e
cd "${GRAFT_REPO_ROOT:-$PWD}"
H="1024:4:hard:nollr 1024:6:hard:nollr 2048:6:hard:nollr 4096:8:hard:nollr 512:4:hard:nollr 256:6:hard:nollr"
echo "== A = shipped, B = s_setprio 3 behind the sample request (hard-only)"
timeout -k 10 300 python -u tools/abbench.py sdr_amd/lib/libsdr_amd.so sdr_amd/lib_prio/libsdr_amd.so --cases $H
echo "== hard-only in natural workgroup order (SDR_XCD_WINDOW=0) against the XCD-aware order: kbench, fresh buffers"
SDR_XCD_WINDOW=0 timeout -k 10 200 python -u tools/kbench.py --cases $H --repeat 3 2>&1 | grep -i median
timeout -k 10 200 python -u tools/kbench.py --cases $H --repeat 3 2>&1 | grep -i median
echo "== generator and chain pieces"
timeout -k 10 300 python -u tools/kbench.py --cases 1024:6:gen 2048:6:gen 1024:6:mc 2048:6:mc 1024:2:mc 2048:6:eq:hard:cnt 2048:6:eq:cnt --repeat 3 2>&1 | grep -i median
