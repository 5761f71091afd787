cd "${GRAFT_REPO_ROOT:-$PWD}"
for v in sr2 sr3 sr5; do
echo "== A = shipped (one symbol group per workgroup), B = $v (runs of symbol groups per workgroup at N <= 256, BPSK / QPSK, LLR-only)"
timeout -k 10 300 python -u tools/abbench.py sdr_amd/lib/libsdr_amd.so sdr_amd/lib_$v/libsdr_amd.so --rounds 3 --cases 64:2 64:1 128:2 128:1 256:2 256:1 || exit 1
done
