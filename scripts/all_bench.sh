export CUDAMAT_BENCH_OTHER_CONFIGS=off   # the headline alone: no side sections (bench.py other_configs) under a profiler / in an A/B
for a in "--workload poisson5" "--precond ilu0 --steps 10" "--precond bjilu0 --steps 10" "--workload mat10000 --steps 500 --warmup 50" "--loop pipelined --workload poisson5" "--workload mat10000 --loop pipelined --steps 500 --warmup 50" "--rows 2000000 --precond ilu0 --steps 10"; do
  python bench.py $a --cpu-baseline off 2>/tmp/err.txt | python -c "
import json,sys
a=sys.argv[1]
try:
    d=json.loads(sys.stdin.read()); print(a, '->', round(d['value'],1), 'it/s', d['spmv_form'], d['roofline']['avg_launch_ms'], d.get('trsv_ms_per_apply'), list(d.get('trsv_traffic',{}).keys())[:2])
except Exception as e:
    print(a, 'FAILED', e); print(open('/tmp/err.txt').read()[-1500:])
" "$a"
done
