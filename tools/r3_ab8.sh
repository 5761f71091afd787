cd "${GRAFT_REPO_ROOT:-$PWD}"
C="1024:6:eq 2048:6:eq 1024:4:eq 1024:8:eq"
for v in pf8 pf32 pf128; do
echo "== A = shipped, B = $v (leader workgroup prefetches the tap lines of the set N ahead into L2)"
timeout -k 10 200 python -u tools/abbench.py sdr_amd/lib/libsdr_amd.so sdr_amd/lib_$v/libsdr_amd.so --cases $C || exit 1
done
