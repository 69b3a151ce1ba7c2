#!/bin/bash
# GPU box: timing sweep of scripts/probe_xcd + FETCH_SIZE/WRITE_SIZE of two settings
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out; mkdir -p $O
P=$R/scripts/probe_xcd
: > $O/probe_xcd.log
for cfg in "16 2 3 0" "16 2 3 1" "16 3 3 0" "8 2 6 0" "8 4 6 0" "32 2 2 0" "16 2 0 0"; do
  timeout -k 10 60 $P $cfg 3 >> $O/probe_xcd.log 2>&1 || { echo "FAILED $cfg" >> $O/probe_xcd.log; exit 1; }
done
for cfg in "16 2 3 0" "16 2 3 1"; do
  tag=$(echo $cfg | tr ' ' '_')
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/pxcd_${tag}_$c
    timeout -k 10 120 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pxcd_${tag}_$c -- $P $cfg 2 > /dev/null 2>&1 || exit 1
    f=$(ls $O/pxcd_${tag}_$c/*/*counter_collection.csv | head -1)
    echo "== $cfg $c" >> $O/probe_xcd.log
    python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'k_xcd' in r['Kernel_Name']: print(r['Counter_Name'], r['Counter_Value'], int(r['End_Timestamp'])-int(r['Start_Timestamp']))
" >> $O/probe_xcd.log
  done
done
echo done >> $O/probe_xcd.log
