#!/bin/bash
# whole-step times (HIP events around 20 executes, tools/probes/p1_probe.py) of several builds of the library, alternating:
#   bash tools/ab_step.sh "prev amd" cfg3_gaussian2_xy [more configs]      REPS=5 by default
root=$(pwd)
libs=$1; shift
REPS=${REPS:-5}
for c in "$@"; do for rep in $(seq $REPS); do for v in $libs; do
  export RECFILTER_AMD_LIB=$root/recfilter_amd/librecfilter_$v.so
  python3 $root/tools/probes/p1_probe.py $c 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v $c step_ms', d['step_ms'], ' '.join(f'{k}={v}' for k,v in d['kernels'].items()))"
done; done; done
