#!/bin/bash
export CUDAMAT_BENCH_OTHER_CONFIGS=off   # the headline alone: no side sections (bench.py other_configs) under a profiler / in an A/B
# GPU box: C5 (ILU(0), 1e7 x 50) per-application time of L^-1 U^-1 over the knobs of the hybrid solve
# usage: scripts/trsv_sweep.sh "K1 K2 ..." "LANES1 ..." [extra env ...]
Ks=${1:-"5 8 12 16"}; Ls=${2:-"4"}; shift 2
for kv in "$@"; do export "$kv"; done
for K in $Ks; do for L in $Ls; do
  out=$(CUDAMAT_TRSV_GROUPS=$K CUDAMAT_TRSV_LANES=$L timeout -k 10 300 python bench.py --precond ilu0 --steps 10 --warmup 2 --cpu-baseline off 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms per apply  %.2f it/s  spmv %.3f ms' % (d['trsv_ms_per_apply'], d['value'], d['roofline']['avg_launch_ms']))")
  echo "groups=$K lanes=$L  $out"
done; done
