#!/bin/bash
root=$(pwd)
for rep in 1 2 3; do for v in prev amd; do
  export RECFILTER_AMD_LIB=$root/recfilter_amd/librecfilter_$v.so
  echo "$v $(python3 tools/probes/shard_probe.py 8 2>/dev/null | tail -1)"
done; done
