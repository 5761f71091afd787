set -e
cd "${GRAFT_REPO_ROOT:-$PWD}"
H="1024:4:hard:nollr 1024:6:hard:nollr 1024:8:hard:nollr 2048:6:hard:nollr 4096:8:hard:nollr 512:4:hard:nollr 256:6:hard:nollr 64:2:hard:nollr 8192:6:hard:nollr"
echo "== A = one symbol per workgroup, B = runs of 2, next symbol requested ahead of the transform, 3 waves per SIMD"
timeout -k 10 300 python -u tools/abbench.py sdr_amd/lib/libsdr_amd.so sdr_amd/lib_p2/libsdr_amd.so --cases $H
echo "== A = one symbol per workgroup, B = runs of 4, next symbol requested ahead of the transform, 3 waves per SIMD"
timeout -k 10 300 python -u tools/abbench.py sdr_amd/lib/libsdr_amd.so sdr_amd/lib_p4/libsdr_amd.so --cases $H
echo "== one library: taps + weight array / taps alone (scalar weight, no weight loads) / weights derived from the taps / no equaliser"
timeout -k 10 400 python -u tools/abbench.py sdr_amd/lib/libsdr_amd.so sdr_amd/lib_p2/libsdr_amd.so --cases 1024:6 1024:6:eq 1024:6:taps 1024:6:wft 2048:6 2048:6:eq 2048:6:taps 2048:6:wft 1024:4:eq 1024:4:wft 4096:8:eq 4096:8:wft 256:6:eq 256:6:wft 64:2:eq 64:2:wft 8192:6:eq 8192:6:wft 512:4:eq 512:4:wft
