for rep in 1 2; do
for v in ${VARIANTS:-amd}; do
RECFILTER_AMD_LIB=recfilter_amd/librecfilter_$v.so python bench.py --steps 50 --warmup 5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['roofline']['achieved'], d['kernels_ms'] if 'kernels_ms' in d else '')"
done; done
