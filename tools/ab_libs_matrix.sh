#!/bin/bash
# same-box A/B of libraries on the matrix path: tools/ab_libs_matrix.sh lib1.so lib2.so ...  (16384^2 and 16380^2, order 12)
for round in 1 2; do
for lib in "$@"; do
  for n in 16384 16380; do
    echo -n "$lib $n: "; RECFILTER_AMD_LIB=$lib python tools/matrix_bench.py image $n 12 2>&1 | grep -o "clamped: path.* ms" | head -1
  done
done
done
