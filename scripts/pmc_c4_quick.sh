#!/bin/bash
# GPU box: L1->L2 requests / L2 hit-miss / fabric read requests of the C4 blocked SpMV's two kernels
cd /tmp && export TMPDIR=/tmp
O=/root/repo/gpurun_out
for kv in "$@"; do export "$kv"; done
for c in "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum" "TCC_EA0_RDREQ_sum TCC_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum" "TCP_TCC_WRITE_REQ_sum" "TCP_TOTAL_ACCESSES_sum"; do
  tag=$(echo $c | tr ' ' '_')
  rm -rf $O/pmc4q_$tag
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc4q_$tag -- python3 /root/repo/bench.py --steps 3 --warmup 1 --cpu-baseline off --drop-in off --other-configs off > /dev/null 2>&1 || echo "pass $c failed"
done
python3 - <<'PY'
import csv,glob,collections
for d in sorted(glob.glob("/root/repo/gpurun_out/pmc4q_*")):
    fs=glob.glob(d+"/*/*counter_collection.csv")
    if not fs: continue
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        k=r["Kernel_Name"].split("(")[0].replace("void ","")
        if "pb_phase" in k: acc[(k,r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k,c),v in sorted(acc.items()):
        print("%-28s %-28s n=%4d avg %.4g" % (k[:28], c, len(v), sum(v)/len(v)))
PY
