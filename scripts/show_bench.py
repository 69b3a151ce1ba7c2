"""One bench line (a JSON file as bench.py prints it) in four lines: headline, memory placement, device state, side sections."""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("%.1f %s  %.3f ms/step  spmv %.3f ms  frac %.3f  %s" % (d["value"], d["unit"], d["ms_per_step"], r["avg_launch_ms"], r["frac"], (d.get("host_placement") or {}).get("gpu_pci")))
print("placement:", d.get("memory_placement"))
g = d.get("gpu_state") or {}
print("gpu_state: sclk %s power %s hbm %s" % ((g.get("sclk_mhz") or {}).get("median"), (g.get("power_w") or {}).get("median"), (g.get("hbm_c") or {}).get("median")))
di = d.get("drop_in") or {}
print("drop_in: first %.3f s, again %.3f s (reused %s)" % (di.get("first_call", {}).get("end_to_end_s", 0), di.get("second_call_same_matrix", {}).get("end_to_end_s", 0), di.get("second_call_same_matrix", {}).get("plan_reused")))
for k, v in (d.get("other_configs") or {}).items():
    dd = (v.get("drop_in") or {}).get("first_call", {})
    print("  %-12s %10.1f it/s  spmv %s ms  setup %s  drop-in first %s  placement %s" % (k, v.get("value", 0), round((v.get("roofline") or {}).get("avg_launch_ms", 0), 4), v.get("setup_s"), dd.get("end_to_end_s"), v.get("memory_placement")))
