mkdir -p gpurun_out/r6c
export RECFILTER_AMD_LIB=$PWD/recfilter_amd/librecfilter_amd_ab.so
for rep in 1 2; do for v in 0 1 2 3; do
  RF_Z_AB=$v python3 tools/probes/p1_probe.py cfg5_generic_xyz 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ab=$v step_ms', d['step_ms'], ' '.join(f'{k}={v}' for k,v in d['kernels'].items()))"
done; done
