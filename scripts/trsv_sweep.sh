#!/bin/bash
# A/B sweep of the triangular-solve knobs (run on the GPU box)
# usage: scripts/trsv_sweep.sh "<bench args>" "<env assignments>" ...
#   e.g. scripts/trsv_sweep.sh "--precond ilu0" "CUDAMAT_TRSV_GROUPS=4" "CUDAMAT_TRSV_GROUPS=8 CUDAMAT_TRSV_LANES=2"
cd /root/repo
args=$1; shift
for cfg in "$@"; do
  out=$(env $cfg timeout -k 10 200 python bench.py $args --steps 5 --warmup 1 --cpu-baseline off 2>/dev/null) || { echo "$cfg FAILED"; exit 1; }
  echo "$out" | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$cfg', 'it/s %.2f' % d['value'], 'trsv ms %.3f' % d['trsv_ms_per_apply'], d['levels'])"
done
