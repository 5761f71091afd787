cd "${GRAFT_REPO_ROOT:-$PWD}"
for v in lib lib_nu2 lib_nu4; do
echo "== $v: generator, Monte-Carlo link (kbench, fresh buffers)"
SDR_AMD_LIB=sdr_amd/$v/libsdr_amd.so timeout -k 10 300 python -u tools/kbench.py --cases 1024:6:gen 2048:6:gen 1024:6:mc 2048:6:mc 1024:2:mc 1024:8:mc --repeat 3 2>&1 | grep -i median
done
