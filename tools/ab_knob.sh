#!/bin/bash
# rocprofv3 kernel averages of the A/B build under several knob settings on the same box, alternating:
#   bash tools/ab_knob.sh "RF_TAILS_YMFMA=0 RF_TAILS_YMFMA=1" cfg4b_gaussian3_rgb [more configs]
# (a setting is NAME=VALUE, or "-" for no knob; needs recfilter_amd/librecfilter_amd_ab.so: make -C recfilter_amd/csrc ab)
root=$(pwd)
settings=$1; shift
REPS=${REPS:-3}
export RECFILTER_AMD_LIB=$root/recfilter_amd/librecfilter_amd_ab.so
cd /tmp && export TMPDIR=/tmp
for c in "$@"; do for rep in $(seq $REPS); do for v in $settings; do
  d=/tmp/rp_ab_$rep; rm -rf $d
  if [ "$v" != "-" ]; then export "$v"; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $root/tools/probes/p1_probe.py $c > /dev/null 2>&1
  if [ "$v" != "-" ]; then unset "${v%%=*}"; fi
  python3 - $d "$v" $c <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/*/*kernel_stats.csv')[0]
out=[]; tot=0.0
for r in csv.DictReader(open(f)):
    n=r['Name']
    for k in ('fused_tails_kernel','fused_pass2','carry_pair','carry_block','xscan_rows','stream_tails','strided_pass','mfma_tails'):
        if k in n:
            out.append(f"{k}={float(r['AverageNs'])/1e3:.1f}")
print(sys.argv[2], sys.argv[3], ' '.join(out))
PY
done; done; done
