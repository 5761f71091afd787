cd "${GRAFT_REPO_ROOT:-$PWD}"
export SDR_LC_EQ=1
for v in lcw3 lcw0; do
echo "== A = $v: LLRs + counters WITH taps as a compile-time instantiation, B = the every-output kernel (run-time tests)"
timeout -k 10 300 python -u tools/abbench.py sdr_amd/lib_$v/libsdr_amd.so sdr_amd/lib/libsdr_amd.so --rounds 2 --cases 2048:6:eq:cnt 1024:6:eq:cnt 1024:4:eq:cnt 2048:4:eq:cnt 512:6:eq:cnt 4096:6:eq:cnt || exit 1
done
