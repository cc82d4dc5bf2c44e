#!/bin/bash
# A/B of the matrix path's tile width on one box: the planner's choice (wide tiles, y tiles of 128 under the x -> y hand-over)
# against tiles of 128 everywhere (RF_MX_NB=4, the choice until round 5)
run() { RECFILTER_AMD_LIB=recfilter_amd/librecfilter_amd_ab.so "$@"; }
for n in 1024 2048 4096 8192 16384; do
  for o in 12 32; do
    echo -n "image $n order $o   planner: "; run python tools/matrix_bench.py image $n $o 2>&1 | grep -o "clamped: path.* ms" | head -1
    echo -n "image $n order $o   T = 128: "; RF_MX_NB=4 run python tools/matrix_bench.py image $n $o 2>&1 | grep -o "clamped: path.* ms" | head -1
  done
done
for s in 1000000 10000000 100000000; do
  echo "audio $s planner:"; run python tools/matrix_bench.py audio $s 2>&1 | grep -E "^(5|13|29)\s"
  echo "audio $s T = 128:"; RF_MX_NB=4 run python tools/matrix_bench.py audio $s 2>&1 | grep -E "^(5|13|29)\s"
done
