cd "${GRAFT_REPO_ROOT:-$PWD}"
echo "== A = round 3 (this tree), B = end of round 2 (commit 21fec3e), same buffers (tools/abbench.py), % of 8 TB/s"
C=""
for n in 64 128 256 512 1024 2048 4096; do for b in 2 4 6 8; do C="$C $n:$b"; done; done
timeout -k 10 600 python -u tools/abbench.py sdr_amd/lib/libsdr_amd.so sdr_amd/lib_r2/libsdr_amd.so --b-round2 --rounds 2 --cases $C || exit 1
C=""
for n in 64 256 1024 2048 4096; do for b in 4 6 8; do C="$C $n:$b:eq $n:$b:hard $n:$b:hard:nollr $n:$b:hard:cnt"; done; done
timeout -k 10 900 python -u tools/abbench.py sdr_amd/lib/libsdr_amd.so sdr_amd/lib_r2/libsdr_amd.so --b-round2 --rounds 2 --cases $C 1024:10:hard:nollr 1024:2:hard:nollr 1024:bins 64:bins 2048:bins || exit 1
