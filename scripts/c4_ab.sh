#!/bin/bash
export CUDAMAT_BENCH_OTHER_CONFIGS=off   # the headline alone: no side sections (bench.py other_configs) under a profiler / in an A/B
# GPU box: C4 (1e7 x 50, fp64 values) SpMV pair time and it/s, alternating env configurations R times
# usage: scripts/c4_ab.sh R "ENV=.. ENV=.." "ENV=.." ...
R=$1; shift
for r in $(seq $R); do for cfg in "$@"; do
  out=$(env $cfg timeout -k 10 300 python bench.py --steps 10 --warmup 2 --cpu-baseline off --drop-in off --other-configs off 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); w=d.get('with_value_dictionary') or {}
print('%.4f ms per SpMV  %.1f it/s  frac %.4f | dict %.4f ms %.1f it/s' % (d['roofline']['avg_launch_ms'], d['value'], d['roofline']['frac'], w.get('avg_launch_ms',0), w.get('value',0)))")
  echo "run $r [$cfg] $out"
done; done
