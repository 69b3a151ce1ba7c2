#!/bin/bash
export CUDAMAT_BENCH_OTHER_CONFIGS=off   # the headline alone: no side sections (bench.py other_configs) under a profiler / in an A/B
# GPU box: mid-size Poisson systems, five launches per iteration with the tuner's SpMV (default) against three launches
# (vector updates folded into the stream SpMV: CUDAMAT_SPMV_MODE=csr CUDAMAT_FUSED=<rows>).  it/s and us per iteration.
R=/root/repo; O=$R/gpurun_out
for rows in ${1:-400000 1000000 3000000}; do
  for form in default fused; do
    if [ $form = fused ]; then export CUDAMAT_SPMV_MODE=csr CUDAMAT_FUSED=100000000; else unset CUDAMAT_SPMV_MODE CUDAMAT_FUSED; fi
    timeout -k 10 200 python3 $R/bench.py --workload poisson5 --rows $rows --nx 1000 --steps 400 --warmup 50 --cpu-baseline off --drop-in off --other-configs off > $O/fm_${rows}_$form.json 2>$O/fm_${rows}_$form.err
    python3 - $O/fm_${rows}_$form.json $rows $form <<'PY'
import json, sys
b = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("rows %s %-8s %8.0f it/s  %7.1f us/iter  loop_form %s  %s" % (sys.argv[2], sys.argv[3], b["value"], 1e6 / b["value"], b.get("loop_form"), b["roofline"]["kernel"][:60]), flush=True)
PY
  done
done
