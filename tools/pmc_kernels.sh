#!/bin/bash
# PMC characterisation of kernels through tools/kbench.py (run on the GPU box via gpurun):
#   tools/pmc_kernels.sh <tag> <symbols> <case> [<case> ...]
# One kernel-trace pass (durations, VGPR/SGPR/LDS per dispatch) and six counter passes, each a run of
# its own with the program directly behind `--` and at most 8 SQ / 4 TCC / 4 TCP / 2 GRBM counters.
# Output: gpurun_out/pmc_<tag>/{trace,p1..p4}/*.csv ; fold with tools/pmc_summary.py.
set -e
cd "${GRAFT_REPO_ROOT:-$PWD}"
export TMPDIR=/tmp
TAG=$1; SYMS=$2; shift 2
OUT=gpurun_out/pmc_$TAG
rm -rf $OUT && mkdir -p $OUT
RUN="python3 tools/kbench.py --steps 3 --warmup 2 --symbols $SYMS --cases $*"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $RUN > $OUT/trace.out 2> $OUT/trace.err
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/p1 -o c -- $RUN > $OUT/p1.out 2> $OUT/p1.err
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/p2 -o c -- $RUN > $OUT/p2.out 2> $OUT/p2.err
timeout -s KILL 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $OUT/p3 -o c -- $RUN > $OUT/p3.out 2> $OUT/p3.err
timeout -s KILL 200 rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_MISC --output-format csv -d $OUT/p4 -o c -- $RUN > $OUT/p4.out 2> $OUT/p4.err
# instruction classes (issue cost differs: tools/valubench) -- two more passes
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FLOPS_FP32 SQ_INSTS_VALU_IOPS --output-format csv -d $OUT/p5 -o c -- $RUN > $OUT/p5.out 2> $OUT/p5.err
timeout -s KILL 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU2 SQ_BUSY_CU_CYCLES SQ_INST_CYCLES_SALU --output-format csv -d $OUT/p6 -o c -- $RUN > $OUT/p6.out 2> $OUT/p6.err
find $OUT -name "*.csv" | wc -l
