cd /tmp && export TMPDIR=/tmp && rm -rf /root/repo/gpurun_out/prof_rank8
CUDAMAT_OVERLAP=0 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /root/repo/gpurun_out/prof_rank8 -- python3 /root/repo/scripts/rank_probe.py 8 > /dev/null 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob("/root/repo/gpurun_out/prof_rank8/*/*_kernel_trace.csv")[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"].split("(")[0].replace("void ","")+" grid="+r["Grid_Size"]+" wg="+r["Workgroup_Size"]
    d[k].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
for k,v in sorted(d.items(), key=lambda kv:-sum(kv[1]))[:12]:
    print("%-70s n=%5d avg %8.1f us" % (k[:70], len(v), sum(v)/len(v)/1e3))
PY
