#!/bin/bash
# GPU box: HIP API time of the preconditioned drop-in calls (scripts/setup_probe.py --precond): which runtime calls the set-up spends its host time in
cd /tmp && export TMPDIR=/tmp
O=/root/repo/gpurun_out
rm -rf $O/hiptrace
CUDAMAT_VERBOSE=0 timeout -k 10 400 rocprofv3 --hip-trace --stats --output-format csv -d $O/hiptrace -- python3 /root/repo/scripts/setup_probe.py --precond > $O/hiptrace.log 2>&1
grep "^call" $O/hiptrace.log
python3 - <<'PY'
import csv, glob
f = glob.glob("/root/repo/gpurun_out/hiptrace/*/*hip_api_stats.csv")
if f:
    rows = list(csv.DictReader(open(f[0])))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:14]:
        print("%-34s calls %7s total %9.1f ms avg %9.1f us max %9.1f us" % (r["Name"][:34], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
