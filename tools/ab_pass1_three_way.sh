#!/bin/bash
# three-way A/B of pass 1 on 64-row tiles: stream / fused / mfma
root=$(pwd)
export RECFILTER_AMD_LIB=$root/recfilter_amd/librecfilter_amd_ab.so
export RF_FUSED_TY=64
cd /tmp && export TMPDIR=/tmp
for c in cfg3_gaussian2_xy cfg2_summed_table; do for rep in 1 2 3; do for v in stream fused mfma; do
  d=/tmp/rp_ab3; rm -rf $d
  unset RF_NO_STREAM_TAILS RF_TAILS_XMFMA
  if [ $v = fused ]; then export RF_NO_STREAM_TAILS=1 RF_TAILS_XMFMA=0; fi
  if [ $v = mfma ]; then export RF_NO_STREAM_TAILS=1 RF_TAILS_XMFMA=1; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $root/tools/probes/p1_probe.py $c > /dev/null 2>&1
  python3 - $d $v $c <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/*/*kernel_stats.csv')[0]
out=[]
for r in csv.DictReader(open(f)):
    n=r['Name']
    for k in ('fused_tails_kernel','fused_pass2','stream_tails','mfma_tails','xscan_rows','carry_pair'):
        if k in n: out.append(f"{k}={float(r['AverageNs'])/1e3:.1f}")
print(sys.argv[2], sys.argv[3], ' '.join(out))
PY
done; done; done
