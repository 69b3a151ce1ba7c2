#!/bin/bash
# GPU box: FETCH_SIZE / TCC hit-miss / LDS / wait counters of one rank's share at G = 8 (scripts/rank_probe.py 8)
cd /tmp && export TMPDIR=/tmp
O=/root/repo/gpurun_out
for c in FETCH_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "TCC_EA0_RDREQ_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum" "TCC_EA0_RDREQ_32B_sum"; do
  tag=$(echo $c | tr ' ' '_')
  rm -rf $O/pmcr_$tag
  for kv in "$@"; do export "$kv"; done
  CUDAMAT_OVERLAP=0 timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmcr_$tag -- python3 /root/repo/scripts/rank_probe.py 8 > /dev/null 2>&1 || echo "pass $c failed"
done
python3 - <<'PY'
import csv,glob,collections
for d in sorted(glob.glob("/root/repo/gpurun_out/pmcr_*")):
    fs=glob.glob(d+"/*/*counter_collection.csv")
    if not fs: continue
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        k=r["Kernel_Name"].split("(")[0].replace("void ","")
        if "pb_phase" in k: acc[(k,r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k,c),v in sorted(acc.items()):
        print("%-28s %-22s n=%4d avg %.4g" % (k[:28], c, len(v), sum(v)/len(v)))
PY
