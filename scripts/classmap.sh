#!/bin/bash
# The judged loop unplaced and placed, with the memory classes of the blocked copy's arrays (VERBOSE=2) beside its speed.
#   gpurun -- bash scripts/classmap.sh
export CUDAMAT_BENCH_OTHER_CONFIGS=off
for pl in 0 1; do
  CUDAMAT_PB_PLACE=$pl CUDAMAT_VERBOSE=2 python bench.py --steps 100 --warmup 5 --cpu-baseline off --drop-in off 2> gpurun_out/classmap.err > gpurun_out/classmap.json || exit 1
  grep "pb classes" gpurun_out/classmap.err | head -4 | cut -c11-120
  python - "$pl" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/classmap.json").read().strip().splitlines()[-1])
print("PB_PLACE=%s  %.1f it/s  spmv %.3f ms" % (sys.argv[1], d["value"], d["roofline"]["avg_launch_ms"]), flush=True)
PY
done
