#!/bin/bash
export CUDAMAT_BENCH_OTHER_CONFIGS=off   # the headline alone: no side sections (bench.py other_configs) under a profiler / in an A/B
# GPU box: phase 2 of the blocked SpMV on a G = 8 shard and on the full matrix -- resident waves vs loads in flight
# (CUDAMAT_PB_MIN_WAVES x CUDAMAT_PB_DEPTH), scripts/rank_probe.py's standard loop line.  usage: scripts/p2_sweep.sh
mkdir -p gpurun_out/p2sweep
for G in 8 1; do
  for W in 4096 2048 1536 1024; do
    for D in 4 8 16; do
      echo "== G=$G min_waves=$W depth=$D"
      CUDAMAT_VALUE_DICT=0 CUDAMAT_PB_MIN_WAVES=$W CUDAMAT_PB_DEPTH=$D timeout -k 10 300 python scripts/rank_probe.py $G 2>&1 | grep "standard  pieces\|Error\|error" || echo "(no line)"
    done
  done
done
