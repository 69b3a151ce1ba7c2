#!/bin/bash
# One box: the judged line (plain), then one kernel-trace pass of the same loop, per-kernel averages of the SpMV pair and the
# vector kernels, and the device's state -- to tell WHICH launch differs between a 175 it/s box and a 190 it/s box.
#   gpurun -- bash scripts/box_phase_probe.sh
R=$(pwd)
bash scripts/headline_state.sh 1 || exit 1
python - <<'PY'
import torch
p = torch.cuda.get_device_properties(0)
print("device: %s  CUs %d  clock %s  L2 %s  mem %d GB" % (p.name, p.multi_processor_count, getattr(p, "clock_rate", None), getattr(p, "L2_cache_size", None), p.total_memory >> 30))
PY
python - <<'PY'
import glob, json
d = json.loads(open("gpurun_out/headline_state_1.json").read().strip().splitlines()[-1])
bdf = (d.get("host_placement") or {}).get("gpu_pci")
for f in ("pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk", "pp_dpm_socclk", "current_compute_partition", "current_memory_partition", "vbios_version", "mem_info_vram_total"):
    try:
        print(f, "->", open("/sys/bus/pci/devices/%s/%s" % (bdf, f)).read().strip().replace("\n", " | "))
    except OSError as e:
        print(f, "->", type(e).__name__)
PY
python - <<'PY'
import glob, json, os
d = json.loads(open("gpurun_out/headline_state_1.json").read().strip().splitlines()[-1])
bdf = (d.get("host_placement") or {}).get("gpu_pci")
fw = []
for f in sorted(glob.glob("/sys/bus/pci/devices/%s/fw_version/*" % bdf)):
    try:
        fw.append("%s=%s" % (os.path.basename(f).replace("_fw_version", ""), open(f).read().strip()))
    except OSError:
        pass
print("fw:", " ".join(fw))
par = []
for f in ("sched_policy", "mes", "mes_kiq", "num_kcq", "cwsr_enable", "noretry", "hws_max_conc_proc", "queue_preemption_timeout_ms", "sdma_phase_quantum", "ppfeaturemask", "gpu_recovery"):
    try:
        par.append("%s=%s" % (f, open("/sys/module/amdgpu/parameters/" + f).read().strip()))
    except OSError:
        pass
print("amdgpu:", " ".join(par))
try:
    print("amdgpu version:", open("/sys/module/amdgpu/version").read().strip(), " kernel:", os.uname().release)
except OSError:
    print("kernel:", os.uname().release)
print("env:", {k: v for k, v in os.environ.items() if k.startswith(("HSA_", "ROCR_", "ROCP", "HIP_", "AMD_", "LD_PRELOAD", "GPU_", "ROC_"))})
PY
rocm-smi --showmemorypartition --showcomputepartition 2>/dev/null | grep -i "partition" | head -4
export CUDAMAT_BENCH_OTHER_CONFIGS=off
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out
rm -rf $O/trace_boxphase
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_boxphase -- python3 $R/bench.py --steps 100 --warmup 5 --cpu-baseline off --drop-in off --other-configs off > $O/trace_boxphase.json 2> $O/trace_boxphase.err || exit 1
python3 - "$O/trace_boxphase" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    acc[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")) for r in csv.DictReader(open(f[0]))))
gaps = collections.defaultdict(list)
for (s0, e0, k0), (s1, e1, k1) in zip(rows, rows[1:]):
    if (s1 - e0) < 200000:          # (inside the loop: not across host synchronisations)
        gaps[(k0.replace("cm::", "")[:14], k1.replace("cm::", "")[:14])].append((s1 - e0) / 1e3)
for kk in sorted(gaps, key=lambda kk: -len(gaps[kk]))[:7]:
    v = sorted(gaps[kk])
    print("gap %-14s -> %-14s n %5d  median %7.2f us  p90 %7.2f" % (kk[0], kk[1], len(v), v[len(v) // 2], v[len(v) * 9 // 10]))
for k in sorted(acc, key=lambda k: -sum(acc[k]))[:8]:
    v = sorted(acc[k])
    print("%-40s n %5d  median %9.1f us  p10 %9.1f  p90 %9.1f" % (k[:40], len(v), v[len(v) // 2], v[len(v) // 10], v[len(v) * 9 // 10]))
PY
rm -rf $O/trace_boxphase
