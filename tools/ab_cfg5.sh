# same-box A/B of two builds of the library on config 5 (2048^3): tools/ab_cfg5.sh [reps]
mkdir -p gpurun_out/r6c
for rep in $(seq ${1:-3}); do for v in prev amd; do
  export RECFILTER_AMD_LIB=$PWD/recfilter_amd/librecfilter_$v.so
  python3 tools/probes/p1_probe.py cfg5_generic_xyz 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v step_ms', d['step_ms'], ' '.join(f'{k}={v}' for k,v in d['kernels'].items()))"
done; done
