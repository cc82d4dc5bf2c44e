#!/bin/bash
# HBM traffic of the matrix path's kernels (FETCH_SIZE / WRITE_SIZE, separate PMC passes, kernel-trace only): 16384^2, order 12,
# x +- y +-, clamped.   -> gpurun_out/pmc_matrix/matrix_image_16384_order12.pmc.json
set -u
root=$(pwd)
out=$root/gpurun_out/pmc_matrix
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/$c -- python3 $root/tools/matrix_bench.py image 16384 12 > /dev/null 2>&1
done
cd $root
python3 tools/pmc_summary.py $(ls $out/FETCH_SIZE/*/*counter_collection.csv | head -1) $(ls $out/WRITE_SIZE/*/*counter_collection.csv | head -1) $out/matrix_image_16384_order12.pmc.json
rm -rf $out/FETCH_SIZE $out/WRITE_SIZE
