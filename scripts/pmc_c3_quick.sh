#!/bin/bash
# GPU box: FETCH_SIZE / L2 hit-miss / TCP->TCC requests of the C3 SpMV kernel under the given env assignments
cd /tmp && export TMPDIR=/tmp
O=/root/repo/gpurun_out
for kv in "$@"; do export "$kv"; done
for c in FETCH_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum" "TA_BUSY_avr"; do
  tag=$(echo $c | tr ' ' '_')
  rm -rf $O/pmc3q_$tag
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc3q_$tag -- python3 /root/repo/bench.py --workload poisson5 --steps 10 --warmup 2 --cpu-baseline off --drop-in off --other-configs off > /dev/null 2>&1 || echo "pass $c failed"
done
python3 - <<'PY'
import csv,glob,collections
for d in sorted(glob.glob("/root/repo/gpurun_out/pmc3q_*")):
    fs=glob.glob(d+"/*/*counter_collection.csv")
    if not fs: continue
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        k=r["Kernel_Name"].split("(")[0].replace("void ","")
        if "spmv_stream" in k: acc[(k,r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k,c),v in sorted(acc.items()):
        print("%-28s %-28s n=%4d avg %.4g" % (k[:28], c, len(v), sum(v)/len(v)))
PY
