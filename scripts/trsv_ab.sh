#!/bin/bash
export CUDAMAT_BENCH_OTHER_CONFIGS=off   # the headline alone: no side sections (bench.py other_configs) under a profiler / in an A/B
# GPU box: C5 per-application time of L^-1 U^-1, alternating env configurations R times
# usage: scripts/trsv_ab.sh R "ENV=.. ENV=.." "ENV=.." ...
R=$1; shift
for r in $(seq $R); do for cfg in "$@"; do
  out=$(env $cfg timeout -k 10 300 python bench.py --precond ilu0 --steps 10 --warmup 2 --cpu-baseline off --drop-in off 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms per apply  %.2f it/s  SpMV in the loop %.4f ms' % (d['trsv_ms_per_apply'], d['value'], d['roofline']['avg_launch_ms']))")
  echo "run $r [$cfg] $out"
done; done
