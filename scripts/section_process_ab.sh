#!/bin/bash
# GPU box: is the preconditioned set-up slow only late in a long-lived process?  full bench (C5 as a side section) against the C5 run alone, alternating
cd /root/repo
for i in 1 2; do
  timeout -k 10 300 python bench.py --cpu-baseline off 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c5=d['other_configs']['rand50_ilu0']; f=c5['drop_in']['first_call']
print('full bench : C5 resident factor %.3f  drop-in first %.3f (upload %.3f factor %.3f)  C4 drop-in %.3f' % (c5['setup_s']['factor'], f['end_to_end_s'], f['upload_s'], f['factor_s'], d['drop_in']['first_call']['end_to_end_s']))"
  timeout -k 10 300 python bench.py --precond ilu0 --steps 50 --warmup 5 --cpu-baseline off 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); f=d['drop_in']['first_call']
print('C5 alone   : C5 resident factor %.3f  drop-in first %.3f (upload %.3f factor %.3f)' % (d['setup_s']['factor'], f['end_to_end_s'], f['upload_s'], f['factor_s']))"
done
