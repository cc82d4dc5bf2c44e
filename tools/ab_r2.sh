#!/bin/bash
# rocprofv3 kernel averages of the current library against the round-2 library kept under old_r2/ (same box, alternating)
# usage: bash tools/ab_r2.sh cfg3_gaussian2_xy [cfg4b_gaussian3_rgb ...]
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
for c in "$@"; do for rep in 1 2; do for v in new old; do
  d=/tmp/rp_${v}_${c}_$rep; rm -rf $d
  if [ $v = old ]; then cd $root/old_r2; else cd $root; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 tools/p1_probe.py $c > /dev/null 2>&1
  python3 - $d $v $c <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/*/*kernel_stats.csv')[0]
out=[]
for r in csv.DictReader(open(f)):
    n=r['Name']
    for k in ('fused_tails_kernel','fused_pass2','carry_pair','carry_block','xscan_rows','stream_tails'):
        if k in n: out.append(f"{k}={float(r['AverageNs'])/1e3:.1f}")
print(sys.argv[2], sys.argv[3], ' '.join(out))
PY
done; done; done
