mkdir -p gpurun_out/r6e
python -m pytest tests -q -m gpu > gpurun_out/r6e/suite_run3.log 2>&1; echo suite rc=$?; tail -3 gpurun_out/r6e/suite_run3.log
python bench.py > gpurun_out/r6e/bench_default_n1.json 2> gpurun_out/r6e/bench_default_n1.err; echo bench rc=$?
python tools/rehearse_n8.py > gpurun_out/r6e/rehearse_n8_one_gpu.txt 2>&1; echo rehearse rc=$?; cat gpurun_out/r6e/rehearse_n8_one_gpu.txt
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --force-stepping --workload cfg3 --rows 2048 --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/r6e/bench_cfg3_one_rank_of_8_strong.json 2> gpurun_out/r6e/bench_cfg3_one_rank_of_8_strong.err; echo slab rc=$?
