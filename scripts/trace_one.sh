#!/bin/bash
export CUDAMAT_BENCH_OTHER_CONFIGS=off   # the headline alone: no side sections (bench.py other_configs) under a profiler / in an A/B
# GPU box: ONE rocprofv3 kernel-trace pass of bench.py (no counters) -> gpurun_out/trace_<name>/ and a per-kernel summary on stdout
# usage: trace_one.sh <name> <bench args...>
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out
name=$1; shift
rm -rf $O/trace_$name
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$name -- python3 $R/bench.py "$@" --cpu-baseline off --drop-in off --other-configs off > $O/trace_$name.json 2> $O/trace_$name.err
python3 - "$O/trace_$name" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    acc[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
rows = sorted(acc.items(), key=lambda kv: -sum(kv[1]))
print("%-50s %6s %12s %12s" % ("kernel", "n", "total ms", "avg us (>20us)"))
for k, v in rows[:45]:
    big = [x for x in v if x > 20]
    print("%-50s %6d %12.3f %12.1f" % (k[:50], len(v), sum(v) / 1e3, sum(big) / len(big) if big else 0))
PY
