set -e
cd "${GRAFT_REPO_ROOT:-$PWD}"
H="1024:4:hard:nollr 1024:6:hard:nollr 1024:8:hard:nollr 2048:6:hard:nollr 4096:8:hard:nollr 512:4:hard:nollr 256:6:hard:nollr 64:2:hard:nollr 8192:6:hard:nollr"
echo "== A = baseline (table parked ahead of the samples), B = hard-only parks it behind the sample request"
timeout -k 10 300 python -u tools/abbench.py sdr_amd/lib_a/libsdr_amd.so sdr_amd/lib/libsdr_amd.so --cases $H
echo "== A = late table, one symbol per workgroup, B = runs of 2 with next-symbol prefetch"
timeout -k 10 300 python -u tools/abbench.py sdr_amd/lib/libsdr_amd.so sdr_amd/lib_p2/libsdr_amd.so --cases $H
echo "== A = late table, one symbol per workgroup, B = runs of 4 with next-symbol prefetch"
timeout -k 10 300 python -u tools/abbench.py sdr_amd/lib/libsdr_amd.so sdr_amd/lib_p4/libsdr_amd.so --cases $H
echo "== A = baseline, B = every kernel parks the table behind the sample request"
timeout -k 10 400 python -u tools/abbench.py sdr_amd/lib_a/libsdr_amd.so sdr_amd/lib_twall/libsdr_amd.so --cases 1024:4:hard:cnt 1024:8:hard:cnt 2048:6:hard:cnt 512:4:hard:cnt 1024:6:eq:hard:cnt 2048:6:eq:hard:cnt 1024:4:hard 2048:6:hard 1024:6 1024:6:eq 2048:6:eq 1024:10 256:6 64:6 4096:8
