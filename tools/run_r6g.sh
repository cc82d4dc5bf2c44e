mkdir -p gpurun_out/r6g
python -m pytest tests -q -m gpu > gpurun_out/r6g/suite_run3.log 2>&1; echo suite rc=$?; tail -3 gpurun_out/r6g/suite_run3.log
python tools/walk_sizes.py --int32 1024 > gpurun_out/r6g/int32_sat_1024.txt 2>&1; cat gpurun_out/r6g/int32_sat_1024.txt | tail -1
