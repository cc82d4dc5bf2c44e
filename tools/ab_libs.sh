#!/bin/bash
# rocprofv3 kernel averages of several builds of the library on the same box, alternating:
#   bash tools/ab_libs.sh "amd t2" cfg3_gaussian2_xy [more configs]      (recfilter_amd/librecfilter_<name>.so)
root=$(pwd)
libs=$1; shift
cd /tmp && export TMPDIR=/tmp
for c in "$@"; do for rep in 1 2 3; do for v in $libs; do
  d=/tmp/rp_${v}_${c}_$rep; rm -rf $d
  export RECFILTER_AMD_LIB=$root/recfilter_amd/librecfilter_$v.so
  rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $root/tools/probes/p1_probe.py $c > /dev/null 2>&1
  python3 - $d $v $c <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/*/*kernel_stats.csv')[0]
out=[]
for r in csv.DictReader(open(f)):
    n=r['Name']
    for k in ('fused_tails_kernel','fused_pass2','carry_pair','carry_block','xscan_rows','stream_tails','strided_pass','mfma_tails'):
        if k in n: out.append(f"{k}={float(r['AverageNs'])/1e3:.1f}")
print(sys.argv[2], sys.argv[3], ' '.join(out))
PY
done; done; done
