# What the driver does at round end, on one GPU box: the GPU suite with -x, smoke(), the default bench line with 5 + 20 steps.
#   gpurun -- bash tools/driver_rehearsal.sh   -> gpurun_out/r6i/
mkdir -p gpurun_out/r6i
python -m pytest tests -x -q -m gpu > gpurun_out/r6i/suite_run_x.log 2>&1; echo "suite (-x, as the driver runs it) rc=$?"; tail -2 gpurun_out/r6i/suite_run_x.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6i/smoke.log 2>&1; echo smoke rc=$?; tail -1 gpurun_out/r6i/smoke.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r6i/bench_driver_protocol.json 2> gpurun_out/r6i/bench_driver_protocol.err; echo bench rc=$?
python -c "
import json
d=json.load(open('gpurun_out/r6i/bench_driver_protocol.json'))
print(d['ms_per_step'], d['ms_per_step_steady'], d['roofline']['frac'], d['roofline']['traffic'], d['configs'][0]['ms_per_step'], d['configs'][0]['roofline']['traffic'])
"
