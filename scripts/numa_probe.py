import os, glob, torch
p = torch.cuda.get_device_properties(0)
print("props:", [a for a in dir(p) if 'pci' in a.lower()])
try:
    bdf = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
except Exception as e:
    bdf = None; print("no pci ids", e)
print("bdf", bdf)
for f in glob.glob("/sys/bus/pci/devices/*/numa_node"):
    d = os.path.dirname(f)
    try:
        cls = open(d + "/class").read().strip(); ven = open(d + "/vendor").read().strip()
    except OSError: continue
    if ven == "0x1002" and cls.startswith("0x03") or (bdf and bdf in d):
        print(d, "numa_node", open(f).read().strip(), "local_cpulist", open(d + "/local_cpulist").read().strip()[:80])
print("nodes:", sorted(glob.glob("/sys/devices/system/node/node*")))
for n in sorted(glob.glob("/sys/devices/system/node/node*/cpulist")):
    print(n, open(n).read().strip()[:100])
print("affinity", len(os.sched_getaffinity(0)), sorted(os.sched_getaffinity(0))[:8], "...")
print("cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else None)
import subprocess
print(subprocess.run("rocm-smi --showtoponuma 2>&1 | head -20", shell=True, capture_output=True, text=True).stdout)
print(subprocess.run("cat /proc/self/status | grep -i -E 'Cpus_allowed_list|Mems_allowed_list'", shell=True, capture_output=True, text=True).stdout)
