cd "${GRAFT_REPO_ROOT:-$PWD}"
echo "== A = round 3 (this tree), B = end of round 2 (commit 21fec3e): equalised variants"
C=""
for n in 64 256 512 1024 2048 4096; do for b in 2 4 6 8; do C="$C $n:$b:eq"; done; done
timeout -k 10 600 python -u tools/abbench.py sdr_amd/lib/libsdr_amd.so sdr_amd/lib_r2/libsdr_amd.so --b-round2 --rounds 2 --cases $C 1024:6:eq:hard 2048:6:eq:hard 1024:6:taps 1024:6:wft || exit 1
