#!/bin/bash
# A/B of the matrix path's pair stages on one box: the planner's choice against one stage per scan (RF_MX_NO_PAIR, A/B build)
run() { RECFILTER_AMD_LIB=recfilter_amd/librecfilter_amd_ab.so "$@"; }
for n in 2048 4096 8192 16384; do
  for o in 5 8 12 16; do
    echo -n "image $n order $o   pairs:   "; run python tools/matrix_bench.py image $n $o 2>&1 | grep -o "clamped: path.* ms" | head -1
    echo -n "image $n order $o   singles: "; RF_MX_NO_PAIR=1 run python tools/matrix_bench.py image $n $o 2>&1 | grep -o "clamped: path.* ms" | head -1
  done
done
for s in 1000000 10000000 100000000; do
  for o in 8 12; do
    echo -n "pairs:   "; run python tools/matrix_bench.py zerophase $s $o 2>&1 | grep -o "^[0-9]* samples.* ms" | head -1
    echo -n "singles: "; RF_MX_NO_PAIR=1 run python tools/matrix_bench.py zerophase $s $o 2>&1 | grep -o "^[0-9]* samples.* ms" | head -1
  done
done
