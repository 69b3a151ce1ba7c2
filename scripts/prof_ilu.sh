#!/bin/bash
export CUDAMAT_BENCH_OTHER_CONFIGS=off   # the headline alone: no side sections (bench.py other_configs) under a profiler / in an A/B
# kernel-trace of the C5 bench (run on the GPU box); prints the top kernels
cd /tmp && export TMPDIR=/tmp
O=/root/repo/gpurun_out/prof_ilu_$1
shift
rm -rf $O
env "$@" true
for kv in "$@"; do export "$kv"; done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 /root/repo/bench.py --precond ilu0 --steps 4 --warmup 1 --cpu-baseline off > $O.json 2> $O.err || exit 1
f=$(find $O -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print(r['Name'][:60].ljust(60), r['Calls'].rjust(6), '%10.3f ms total' % (int(r['TotalDurationNs'])/1e6), '%10.1f us avg' % (float(r['AverageNs'])/1e3))
PY
