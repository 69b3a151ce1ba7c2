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
m = d.get("memory_placement") or {}
print("run %s: %.1f it/s  spmv %.3f ms  triad %.0f read %.0f GB/s  sclk %s  power %s  hbm %s  %s  placed %s in %.3f s (%s)" % (
    sys.argv[1], d["value"], r["avg_launch_ms"], c["gbs"], c["read_gbs"], (g.get("sclk_mhz") or {}).get("median"), (g.get("power_w") or {}).get("median"),
    (g.get("hbm_c") or {}).get("median"), (d.get("host_placement") or {}).get("gpu_pci"), m.get("placed"), m.get("seconds", 0.0), m.get("blocks_by_class")), flush=True)
PY
done
