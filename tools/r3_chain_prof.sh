cd "${GRAFT_REPO_ROOT:-$PWD}"
export TMPDIR=/tmp
rm -rf gpurun_out/chainprof && mkdir -p gpurun_out/chainprof
SDR_CHAIN_SERIAL=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/chainprof -o chain -- python3 bench.py --chain fft2048_64qam_eq_4M_chain > gpurun_out/chainprof/line.json 2> gpurun_out/chainprof/err.log
echo rc=$?
