#!/bin/bash
export CUDAMAT_BENCH_OTHER_CONFIGS=off   # the headline alone: no side sections (bench.py other_configs) under a profiler / in an A/B
# GPU box: alternate configurations (env assignments, one quoted string each) R times, so that clock / thermal drift
# during the sequence hits all of them alike.  usage: scripts/ab_env.sh R "<command>" "ENV_A=.. ENV_B=.." "ENV_A=.." ...
R=$1; CMD=$2; shift 2
for r in $(seq $R); do
  for cfg in "$@"; do
    out=$(env $cfg bash -c "$CMD" 2>/dev/null | tail -1)
    echo "run $r [$cfg] $out"
  done
done
