# merged x carry scan (xscan_rows completes the x tails) against the separate launch, mid-size images
# (developer switches exist only in the A/B build of the library: make -C recfilter_amd/csrc ab)
make -s -C recfilter_amd/csrc ab -j8 && export RECFILTER_AMD_LIB=$PWD/recfilter_amd/librecfilter_amd_ab.so || exit 1
export RF_MERGED_CARRY_X_ALL=1
for s in ${SIZES:-1280 2048 3072 4096}; do for w in ${WORKLOADS:-cfg3 cfg4b cfg2 cfg4a}; do for v in merged separate; do
if [ $v = separate ]; then export RF_NO_MERGED_CARRY_X=1; else unset RF_NO_MERGED_CARRY_X; fi
python bench.py --workload $w --size $s --steps 300 --warmup 30 --no-cpu-baseline | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w $s $v', d['ms_per_step'], d['config']['tiles'], {k:round(v,4) for k,v in d['kernels_ms'].items()})"
done; done; done
