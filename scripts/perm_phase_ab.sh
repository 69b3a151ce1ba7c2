#!/bin/bash
export CUDAMAT_BENCH_OTHER_CONFIGS=off   # the headline alone: no side sections (bench.py other_configs) under a profiler / in an A/B
# GPU box: per-phase durations of the SpMV inside the preconditioned C5 loop with the loop in level-major spaces
# (CUDAMAT_TRSV_PERM=1: second blocked copy, rows in L's order, columns in U's positions) and without (=0: the original copy),
# alternating on one box, from rocprofv3 kernel traces.   usage: scripts/perm_phase_ab.sh [rounds]
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out
for r in $(seq ${1:-2}); do for perm in 1 0; do
  rm -rf $O/permab_$perm
  CUDAMAT_TRSV_PERM=$perm timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/permab_$perm -- python3 $R/bench.py --precond ilu0 --steps 6 --warmup 2 --cpu-baseline off --drop-in off --other-configs off > /dev/null 2>&1
  python3 - $O/permab_$perm $perm $r <<'PY'
import csv, glob, sys, collections
d, perm, r = sys.argv[1:4]
f = sorted(glob.glob(d + "/*/*_kernel_trace.csv"))[-1]
dur = collections.defaultdict(list)
for row in csv.DictReader(open(f)):
    k = row["Kernel_Name"].split("(")[0].replace("void ", "")
    t = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
    if k.startswith("cm::k_pb_phase") and t > 600000:          # the loop's own SpMV (the far parts of the triangular solves are shorter)
        dur[k].append(t)
print("round %s TRSV_PERM=%s  " % (r, perm) + "  ".join("%s n=%d avg %.1f us" % (k, len(v), sum(v) / len(v) / 1e3) for k, v in sorted(dur.items())), flush=True)
PY
done; done
