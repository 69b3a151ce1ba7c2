#!/bin/bash
# GPU box: C3 (Poisson 4000 x 2500, fp64 values) SpMV time and it/s, alternating env configurations R times
# usage: scripts/c3_ab.sh R "ENV=.. ENV=.." "ENV=.." ...
R=$1; shift
for r in $(seq $R); do for cfg in "$@"; do
  out=$(env $cfg timeout -k 10 300 python bench.py --workload poisson5 --steps 200 --warmup 20 --cpu-baseline off --drop-in off --other-configs off 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f ms per SpMV  %.1f it/s  kernel %s' % (d['roofline']['avg_launch_ms'], d['value'], d['roofline']['kernel']))")
  echo "run $r [$cfg] $out"
done; done
