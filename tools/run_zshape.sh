mkdir -p gpurun_out/r6c
./tools/microbench/zpass_shape 2048 512 > gpurun_out/r6c/zpass_shape.txt 2>&1
export RECFILTER_AMD_LIB=$PWD/recfilter_amd/librecfilter_amd_ab.so
for w in 3 2 1 4 3 2; do RF_STRIDED_WGS=$w python3 tools/probes/p1_probe.py cfg5_generic_xyz 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wgs=$w step_ms', d['step_ms'], ' '.join(f'{k}={v}' for k,v in d['kernels'].items()))"; done > gpurun_out/r6c/strided_wgs.txt 2>&1
cat gpurun_out/r6c/zpass_shape.txt gpurun_out/r6c/strided_wgs.txt
