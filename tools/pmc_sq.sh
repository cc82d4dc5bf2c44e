#!/bin/bash
# Shader-side PMC passes (one counter group per run, kernel-trace only) of the default bench command:
# what the waves of each kernel spend their cycles on.  -> gpurun_out/pmc_sq_<tag>/{A,B,C}.csv
set -u
tag=${1:-latest}
extra=${2:-}          # extra bench.py arguments, e.g. "--workload cfg4b"
# PMC_SQ_PROG: another program (and its arguments) instead of bench.py, e.g. "tools/matrix_bench.py image 16384 12"
root=$(pwd)
out=$root/gpurun_out/pmc_sq_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run() {  # name, counters...
    name=$1; shift
    rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/run_$name -- python3 $root/${PMC_SQ_PROG:-bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-configs $extra} > $out/$name.log 2>&1
    cp $(ls $out/run_$name/*/*counter_collection.csv | head -1) $out/$name.csv && rm -rf $out/run_$name
}
run A SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU
run B SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY
run C SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL
run D SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY
ls -la $out
