#!/bin/bash
# Regenerates profiles/rN/apps: every app of the reference at 4096^2 and 16384^2, the audio sweeps at 10 Mi samples and the
# 64..4096 width sweep (run on a GPU box; results land in gpurun_out/apps).
mkdir -p gpurun_out/apps
for w in 4096 16384; do
  for app in summed_table gaussian_3xy gaussian_1xy_2xy gaussian_1xy_1xy_1xy gaussian_1xy_2x_2y gaussian_3x_3y bicubic biquintic_overlapped biquintic_cascaded usm_naive usm_optimized box_filter_1 box_filter_3 box_filter_6 diff_gauss; do
    echo -n "$app " ; python tools/profile_app.py $app -w $w -iter 50 --outdir gpurun_out/apps 2>/dev/null
  done
done > gpurun_out/apps/apps_w4096_w16384.txt 2>&1
python tools/profile_app.py audio_biquads -w 10000000 -t 1000 -iter 50 --outdir gpurun_out/apps > gpurun_out/apps/audio_biquads.txt 2>/dev/null      # the app's own length and tile
python tools/profile_app.py audio_high_order -w 10000000 -t 1000 -iter 20 --outdir gpurun_out/apps > gpurun_out/apps/audio_high_order.txt 2>/dev/null
python tools/profile_app.py gaussian_3xy -w 0 -iter 20 --outdir gpurun_out/apps > /dev/null 2>&1
cat gpurun_out/apps/apps_w4096_w16384.txt gpurun_out/apps/audio_biquads.txt gpurun_out/apps/audio_high_order.txt
