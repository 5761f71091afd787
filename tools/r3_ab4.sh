cd "${GRAFT_REPO_ROOT:-$PWD}"
C="1024:6:eq 2048:6:eq 1024:4:eq 1024:6:taps"
for v in t16 t1 t17 s16 t16s16; do
echo "== A = shipped, B = $v"
timeout -k 10 200 python -u tools/abbench.py sdr_amd/lib/libsdr_amd.so sdr_amd/lib_$v/libsdr_amd.so --cases $C || exit 1
done
