cd "${GRAFT_REPO_ROOT:-$PWD}"
for v in lib lib_mcu2 lib_mcu4; do
echo "== $v"
SDR_AMD_LIB=sdr_amd/$v/libsdr_amd.so timeout -k 10 300 python -u tools/kbench.py --cases 1024:2:mc 1024:4:mc 1024:6:mc 1024:8:mc --repeat 3 2>&1 | grep -i median
SDR_AMD_LIB=sdr_amd/$v/libsdr_amd.so timeout -k 10 100 python bench.py --sweep | cut -c1-260
done
