import csv,glob,sys
f=sorted(glob.glob(sys.argv[1]+"/*/*kernel_trace.csv"), key=lambda p: __import__("os").path.getmtime(p))[-1]
rows=[r for r in csv.DictReader(open(f))]
for name in ("k_pb_group","k_pb_scatter","k_pb_count"):
    v=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in rows if name in r["Kernel_Name"]]
    print(name,"total %.1f ms"%(sum(v)/1e3),[round(x) for x in v])
