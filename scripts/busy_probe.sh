#!/bin/bash
export CUDAMAT_BENCH_OTHER_CONFIGS=off   # the headline alone: no side sections (bench.py other_configs) under a profiler / in an A/B
# GPU box: is the five-launch loop host-bound at mid sizes?  Kernel-trace busy fraction of the timed region for Poisson systems
# of a few sizes: sum of kernel durations / (last end - first start) over the last 60 % of the trace's loop kernels.
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out
for rows in ${1:-400000 1000000 3000000}; do
  rm -rf $O/busy_$rows
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/busy_$rows -- python3 $R/bench.py --workload poisson5 --rows $rows --nx 1000 --steps 400 --warmup 50 --cpu-baseline off --drop-in off --other-configs off > $O/busy_$rows.json 2>/dev/null
  python3 - $O/busy_$rows $rows $O/busy_$rows.json <<'PY'
import csv, glob, json, sys
d, rows, js = sys.argv[1:4]
f = sorted(glob.glob(d + "/*/*_kernel_trace.csv"))[-1]
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]) for r in csv.DictReader(open(f))), key=lambda t: t[0])
loop = [k for k in ks if any(s in k[2] for s in ("k_update_p", "k_half", "k_full", "k_spmv", "k_fspmv"))]
loop = loop[int(len(loop) * 0.4):]
busy = sum(e - s for s, e, _ in loop)
span = loop[-1][1] - loop[0][0]
b = json.loads(open(js).read().strip().splitlines()[-1])
print("rows %s: %d loop kernels, busy %.1f %% of the span, avg kernel %.1f us, avg gap %.1f us; bench %.0f it/s (%s)" % (
    rows, len(loop), 100.0 * busy / span, busy / len(loop) / 1e3, (span - busy) / len(loop) / 1e3, b["value"], b["roofline"]["kernel"][:40]), flush=True)
PY
done
