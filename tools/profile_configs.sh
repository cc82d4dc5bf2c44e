#!/bin/bash
# Bench line + rocprofv3 kernel-trace stats of every BASELINE config that fits one GPU (cfg5 at 1024^3 and 2048^3).
#   bash tools/profile_configs.sh <tag>   -> gpurun_out/configs_<tag>/<cfg>.{json,csv}
set -u
tag=${1:-latest}
root=$(pwd)
out=$root/gpurun_out/configs_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for spec in "cfg2 0" "cfg3 0" "cfg4a 0" "cfg4b 0" "cfg5 1024" "cfg5 2048"; do
    set -- $spec
    name=$1; [ "$2" != "0" ] && name=$1_$2
    extra=""; [ "$2" != "0" ] && extra="--size $2"
    python3 $root/bench.py --workload $1 $extra --steps 50 --warmup 10 --no-cpu-baseline > $out/$name.json 2> $out/$name.err
    rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_$name -- python3 $root/bench.py --workload $1 $extra --steps 10 --warmup 3 --no-cpu-baseline > $out/$name.trace.log 2>&1
    cp $(ls $out/trace_$name/*/*kernel_stats.csv | head -1) $out/$name.kernel_stats.csv
    rm -rf $out/trace_$name $out/$name.trace.log
    echo "$name: $(cut -c1-60 $out/$name.json | head -1)"
done
