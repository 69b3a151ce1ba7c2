#!/bin/bash
export CUDAMAT_BENCH_OTHER_CONFIGS=off
# GPU box: C5 with the far parts' blocked plans at fewer, longer-segment waves (PB_MIN_WAVES; the loop's own copy has 12 288
# natural waves and is not affected), alternating with the default on one box
cd /root/repo
for rep in 1 2; do
for cfg in "X=0" "CUDAMAT_PB_MIN_WAVES=1024" "CUDAMAT_PB_MIN_WAVES=1536" "CUDAMAT_PB_MIN_WAVES=3072"; do
  out=$(env $cfg timeout -k 10 300 python bench.py --precond ilu0 --steps 10 --warmup 2 --cpu-baseline off --drop-in off 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value'],2),'it/s  trsv',round(d['trsv_ms_per_apply'],3),'ms  spmv',round(d['roofline']['avg_launch_ms'],3),'ms')")
  echo "$cfg: $out"
done
done
