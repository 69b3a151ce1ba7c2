#!/bin/bash
# Does the per-launch timing (4 HIP event records per iteration) change the iteration it measures?  Alternates the judged
# loop with and without FLAG_PROFILE in separate processes on one box.   gpurun -- bash scripts/events_ab.sh [pairs]
export CUDAMAT_BENCH_OTHER_CONFIGS=off
for i in $(seq 1 "${1:-3}"); do
  for ev in 0 1; do
    CUDAMAT_BENCH_NO_EVENTS=$ev python bench.py --steps 100 --warmup 5 --cpu-baseline off --drop-in off > gpurun_out/events_ab.json 2> gpurun_out/events_ab.err || exit 1
    python - "$i" "$ev" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/events_ab.json").read().strip().splitlines()[-1])
print("pair %s no_events=%s: %.1f it/s  %.3f ms/step  spmv(events) %s" % (sys.argv[1], sys.argv[2], d["value"], d["ms_per_step"], d["roofline"].get("avg_launch_ms")), flush=True)
PY
  done
done
