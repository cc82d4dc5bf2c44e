# (developer switches exist only in the A/B build of the library: make -C recfilter_amd/csrc ab)
make -s -C recfilter_amd/csrc ab -j8 && export RECFILTER_AMD_LIB=$PWD/recfilter_amd/librecfilter_amd_ab.so || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests_ytm.log 2>&1 || { tail -30 gpurun_out/gpu_tests_ytm.log; exit 1; }
tail -3 gpurun_out/gpu_tests_ytm.log
for rep in 1 2 3; do
for w in cfg3 cfg4b cfg4a cfg2; do
for v in tile row; do
if [ $v = row ]; then export RF_YT_ROW_MAJOR=1; else unset RF_YT_ROW_MAJOR; fi
python bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w $v', d['ms_per_step'])"
done; done; done | tee gpurun_out/ab_ytm.txt
