#!/bin/bash
# rocprofv3 kernel averages of the current library against the libraries of earlier rounds kept under old_r2/ and old_r3/
# (git archive <round's last commit> recfilter_amd tools/p1_probe.py tests/ref_cases.py include | tar -x -C old_rN; make),
# each with its own Python layer, same box, alternating.
#   bash tools/ab_rounds.sh cfg3_gaussian2_xy [cfg4b_gaussian3_rgb ...]      REPS=4 by default
root=$(pwd)
REPS=${REPS:-4}
cd /tmp && export TMPDIR=/tmp
for c in "$@"; do for rep in $(seq $REPS); do for v in r2 r3 new; do
  d=/tmp/rp_${v}_$rep; rm -rf $d
  if [ $v = new ]; then cd $root; probe=tools/probes/p1_probe.py; else cd $root/old_$v; probe=tools/p1_probe.py; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $probe $c > /tmp/p1_$v.json 2>/dev/null
  python3 - $d $v $c /tmp/p1_$v.json <<'PY'
import csv,glob,json,sys
f=glob.glob(sys.argv[1]+'/*/*kernel_stats.csv')[0]
out=[]
for r in csv.DictReader(open(f)):
    n=r['Name']
    for k in ('fused_tails_kernel','mfma_tails','stream_tails','fused_pass2','carry_pair','carry_block','xscan_rows'):
        if k in n: out.append(f"{k}={float(r['AverageNs'])/1e3:.1f}")
try: step=json.loads(open(sys.argv[4]).read().strip().splitlines()[-1])['step_ms']
except Exception: step=None
print(sys.argv[2], sys.argv[3], 'step_ms', step, ' '.join(out), flush=True)
PY
done; done; done
