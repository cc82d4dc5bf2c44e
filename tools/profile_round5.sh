#!/bin/bash
# Round-5 evidence that is not a BASELINE config: the matrix path (audio sweep, 2-D order 12 / 32) with rocprofv3 kernel stats,
# the cascaded apps with their stages merged into one plan against chained stages.   -> gpurun_out/r5x/
set -u
root=$(pwd)
out=$root/gpurun_out/r5x
mkdir -p $out
python3 tools/matrix_bench.py kernels audio > $out/matrix_audio_sweep.txt 2>&1
python3 tools/matrix_bench.py kernels image 16384 12 > $out/matrix_image_16384.txt 2>&1
python3 tools/matrix_bench.py kernels image 16384 32 >> $out/matrix_image_16384.txt 2>&1
python3 tools/matrix_bench.py kernels image 4096 12 >> $out/matrix_image_16384.txt 2>&1
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_mx -- python3 $root/tools/matrix_bench.py image 16384 12 > /dev/null 2>&1)
cp $(ls $out/trace_mx/*/*kernel_stats.csv | head -1) $out/matrix_image_16384_order12.kernel_stats.csv
rm -rf $out/trace_mx
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_au -- python3 $root/tools/matrix_bench.py audio > /dev/null 2>&1)
cp $(ls $out/trace_au/*/*kernel_stats.csv | head -1) $out/matrix_audio_sweep.kernel_stats.csv
rm -rf $out/trace_au
{
  echo "# tools/profile_app.py <app> -w 16384 -iter 10: ms per realize of the LAST stage, merged (one plan for the whole cascade, round 5) | chained (one plan per stage)"
  for a in gaussian_1xy_2xy gaussian_3x_3y gaussian_1xy_1xy_1xy gaussian_1xy_2x_2y biquintic_cascaded gaussian_3xy; do
    m=$(python3 tools/profile_app.py $a -w 16384 -iter 10 2>/dev/null | tail -1 | cut -f2)
    c=$(PROFILE_APP_CHAINED_CASCADES=1 python3 tools/profile_app.py $a -w 16384 -iter 10 2>/dev/null | tail -1 | cut -f2)
    echo "$a merged $m ms   chained $c ms"
  done
} > $out/cascade_merge_ab.txt 2>&1
python3 tools/profile_app.py audio_high_order -w 10000000 -t 32 -iter 20 > $out/audio_high_order_10M.txt 2>&1
python3 tools/profile_app.py audio_biquads -w 10000000 -t 32 -iter 20 > $out/audio_biquads_10M.txt 2>&1
cat $out/cascade_merge_ab.txt; tail -16 $out/audio_high_order_10M.txt
