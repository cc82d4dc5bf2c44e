#!/bin/bash
# A/B of the matrix path's carry chain on one box: chunks of 16 + propagation (two passes over the tails) against ONE sequential
# chain over the line (RF_MX_TOP = the longest line, in tiles, that is chained in one go; A/B build)
run() { RECFILTER_AMD_LIB=recfilter_amd/librecfilter_amd_ab.so "$@"; }
for n in 2048 4096 8192 16384; do
  for o in 12 32; do
    for top in 24 512; do
      echo -n "image $n order $o RF_MX_TOP=$top: "; RF_MX_TOP=$top run python tools/matrix_bench.py kernels image $n $o 2>&1 | grep -A1 "clamped" | tr '\n' ' ' | sed 's/path tiled_matrix//; s/Msamples.*rel err nan//' | cut -c1-420; echo
    done
  done
done
RF_MX_TOP=24 run python tools/matrix_bench.py audio 2>&1 | grep -E "^(13|29)\s"
