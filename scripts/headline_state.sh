#!/bin/bash
# The judged line alone (no side sections, no CPU baseline sweep beyond the default) with the device's state beside it:
# one line per run, for comparing boxes / processes (HISTORY round 5, "what still moves the headline").
#   gpurun -- bash scripts/headline_state.sh [runs]
runs=${1:-2}
mkdir -p gpurun_out
for i in $(seq 1 "$runs"); do
  CUDAMAT_BENCH_OTHER_CONFIGS=off python bench.py --steps 100 --warmup 5 > gpurun_out/headline_state_$i.json 2> gpurun_out/headline_state_$i.err || exit 1
  python - "$i" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/headline_state_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
g, r = d.get("gpu_state") or {}, d["roofline"]
c = r["measured_stream_ceiling"]
print("run %s: %.1f it/s  spmv %.3f ms  triad %.0f read %.0f GB/s  sclk %s  power %s  junction %s  hbm %s  %s" % (
    sys.argv[1], d["value"], r["avg_launch_ms"], c["gbs"], c["read_gbs"], g.get("sclk_mhz"), g.get("power_w"), g.get("junction_c"), g.get("hbm_c"),
    (d.get("host_placement") or {}).get("gpu_pci")), flush=True)
PY
done
