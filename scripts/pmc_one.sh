#!/bin/bash
export CUDAMAT_BENCH_OTHER_CONFIGS=off   # the headline alone: no side sections (bench.py other_configs) under a profiler / in an A/B
# usage: scripts/pmc_one.sh <name> <kernel substring> <bench args...>   (GPU box) -- FETCH_SIZE / WRITE_SIZE of one kernel
cd /tmp && export TMPDIR=/tmp
name=$1; kern=$2; shift 2
O=/root/repo/gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc1_${name}_$c
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc1_${name}_$c -- python3 /root/repo/bench.py "$@" --steps 2 --warmup 1 --cpu-baseline off > /dev/null 2>&1 || exit 1
done
python3 - "$O" "$name" "$kern" <<'PY'
import csv,glob,sys
O,name,kern=sys.argv[1:4]
tot={}
for c in ("FETCH_SIZE","WRITE_SIZE"):
    f=glob.glob("%s/pmc1_%s_%s/*/*counter_collection.csv"%(O,name,c))[0]
    v=[float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if kern in r["Kernel_Name"] and int(r["End_Timestamp"])-int(r["Start_Timestamp"])>20000]
    tot[c]=sum(v)/len(v)
    print(c, "launches", len(v), "avg KB", tot[c])
print("hbm bytes per launch (2*FETCH+WRITE): %.3f GB" % ((2*tot["FETCH_SIZE"]+tot["WRITE_SIZE"])*1024/1e9))
PY
