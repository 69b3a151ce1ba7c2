# run on the GPU box: single-launch loop vs three launches per iteration over system sizes (5-point Poisson grids)
R=${GRAFT_REPO_ROOT:-/root/repo}
for nx in 50 100 141 180 200 230 255; do
  n=$((nx*nx))
  for r in 0 1; do
    CUDAMAT_RESIDENT=$r timeout -k 10 120 python3 $R/bench.py --workload poisson5 --rows $n --nx $nx --steps 3000 --warmup 200 --cpu-baseline off 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('n', sys.argv[2], 'tiles', (int(sys.argv[2])+255)//256, 'resident', sys.argv[1], round(d['value']), 'it/s', round(d['ms_per_step']*1000,2), 'us/iter')" $r $n
  done
done
