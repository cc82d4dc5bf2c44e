#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate PMC passes, kernel-trace only) of config 5 at 1024^3 per kernel: the one-read
# pass 1 (default) and the two first passes (RF_WALK=0 in the A/B build).  -> gpurun_out/pmc_cfg5/{walk,staged}.json
set -u
root=$(pwd)
out=$root/gpurun_out/pmc_cfg5
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export RECFILTER_AMD_LIB=$root/recfilter_amd/librecfilter_amd_ab.so
for mode in walk staged; do
  if [ $mode = staged ]; then export RF_WALK=0; else unset RF_WALK; fi
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/${mode}_$c -- python3 $root/bench.py --workload cfg5 --size 1024 --steps 3 --warmup 1 --no-cpu-baseline --no-extra-configs > /dev/null 2>&1
  done
  python3 $root/tools/pmc_summary.py $(ls $out/${mode}_FETCH_SIZE/*/*counter_collection.csv | head -1) $(ls $out/${mode}_WRITE_SIZE/*/*counter_collection.csv | head -1) $out/$mode.json > /dev/null
  rm -rf $out/${mode}_FETCH_SIZE $out/${mode}_WRITE_SIZE
done
ls -la $out
