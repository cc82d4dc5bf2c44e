#!/bin/bash
# Collect the round's evidence on a GPU box: bench line, rocprofv3 kernel-trace stats and the two PMC passes
# (FETCH_SIZE, WRITE_SIZE -- separate runs, never combined with a sys trace) of the same command.
#   bash tools/refresh_profiles.sh <round, e.g. r3>   -> gpurun_out/prof_<round>/{bench.json,kernel_stats.csv,pmc_traffic.json}
set -u
tag=${1:-latest}
root=$(pwd)
out=$root/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $root/bench.py --no-cpu-baseline --no-extra-configs > $out/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-configs > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-configs > /dev/null 2>&1
cd $root
cp $(ls $out/trace/*/*kernel_stats.csv | head -1) $out/kernel_stats.csv
python3 tools/pmc_summary.py $(ls $out/pmc_fetch/*/*counter_collection.csv | head -1) $(ls $out/pmc_write/*/*counter_collection.csv | head -1) $out/pmc_traffic.json
rm -rf $out/trace $out/pmc_fetch $out/pmc_write
# config 5 at 2048^3 (the `configs[0]` line of a default run): its own two PMC passes -> pmc_traffic_cfg5_2048.json
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc5_$c -- python3 $root/bench.py --workload cfg5 --steps 3 --warmup 1 --no-cpu-baseline --no-extra-configs > /dev/null 2>&1)
done
python3 tools/pmc_summary.py $(ls $out/pmc5_FETCH_SIZE/*/*counter_collection.csv | head -1) $(ls $out/pmc5_WRITE_SIZE/*/*counter_collection.csv | head -1) $out/pmc_traffic_cfg5_2048.json
rm -rf $out/pmc5_FETCH_SIZE $out/pmc5_WRITE_SIZE
cp $out/pmc_traffic_cfg5_2048.json $root/profiles/$tag/pmc_traffic_cfg5_2048.json
# the bench line LAST, with the fresh PMC summary in place: bench.py reports `traffic` only from a summary collected on the
# kernel sources that are running (git_head is stamped afterwards, where .git exists: tools/stamp_head.py)
mkdir -p $root/profiles/$tag && cp $out/pmc_traffic.json $root/profiles/$tag/pmc_traffic.json
python3 $root/bench.py > $out/bench.json 2> $out/bench.err
cut -c1-1500 $out/bench.json; head -8 $out/kernel_stats.csv; cat $out/pmc_traffic.json | head -30
