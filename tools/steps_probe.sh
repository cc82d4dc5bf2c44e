# how the driver's --steps/--warmup choice moves ms_per_step on one box
for rep in 1 2; do
for sw in "20 5" "50 10" "200 20" "20 5"; do
set -- $sw
python bench.py --steps $1 --warmup $2 --no-cpu-baseline | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('steps $1 warmup $2', d['ms_per_step'], sum(d['kernels_ms'].values()))"
done; done
