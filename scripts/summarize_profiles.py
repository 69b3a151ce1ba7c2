"""turn gpurun_out/prof_<name> + pmcf_/pmcw_<name> (scripts/collect_profiles.sh) into profiles/<dir>/"""
import collections, csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def newest(pattern):
    """gpurun merges every call's files into gpurun_out/: take the most recent match"""
    fs = sorted(glob.glob(pattern), key=os.path.getmtime)
    return fs[-1:] if fs else []


def main(tag):
    for name in ("rand50", "poisson5", "ilu0", "mat10000"):
        src = newest(os.path.join(ROOT, "gpurun_out", "prof_%s" % name, "*", "*_kernel_stats.csv"))
        if not src:
            continue
        dst = os.path.join(ROOT, "profiles", "%s_%s" % (tag, name))
        os.makedirs(dst, exist_ok=True)
        shutil.copy(src[0], os.path.join(dst, "kernel_stats.csv"))
        shutil.copy(os.path.join(ROOT, "gpurun_out", "prof_%s.json" % name), os.path.join(dst, "bench_line.json"))
        # real (non-frozen) launch durations from the kernel trace
        tr = newest(os.path.join(ROOT, "gpurun_out", "prof_%s" % name, "*", "*_kernel_trace.csv"))[0]
        dur = collections.defaultdict(list)

        def klass(k, d):
            """C5: the blocked kernels serve the loop's own SpMV (>= 0.6 ms per launch) AND the far parts of the triangular
            solves (shorter launches): keep them apart"""
            if name == "ilu0" and k.startswith("cm::k_pb_phase"):
                return k + ("[far part of a triangular solve]" if d < 600000 else "[SpMV of the loop]")
            return k

        for r in csv.DictReader(open(tr)):
            d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            k = klass(r["Kernel_Name"].split("(")[0].replace("void ", ""), d)
            if (d > (2000 if name == "mat10000" else 20000) or "trsv" in k) and not k.startswith("at::") and "rocclr" not in k:
                dur[k].append(d)
        out = {"_note": "rocprofv3 --kernel-trace (durations of launches > 20 us: the rest are frozen no-ops past the stopping "
                        "point) and separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes; counters in KB per launch; on gfx950 "
                        "FETCH_SIZE tallies wide coalesced reads at half their bytes (calibration: k_half reads 4 vectors, the "
                        "counter shows 2) => hbm_bytes_per_launch_corrected = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024"}
        for k, v in dur.items():
            out[k] = {"launches": len(v), "avg_us": sum(v) / len(v) / 1e3, "min_us": min(v) / 1e3, "max_us": max(v) / 1e3}
        for pre, cn in (("pmcf", "FETCH_SIZE"), ("pmcw", "WRITE_SIZE")):
            fs = newest(os.path.join(ROOT, "gpurun_out", "%s_%s" % (pre, name), "*", "*_counter_collection.csv"))
            if not fs:
                continue
            acc = collections.defaultdict(list)
            for r in csv.DictReader(open(fs[0])):
                d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                k = klass(r["Kernel_Name"].split("(")[0].replace("void ", ""), d)
                if (d > (2000 if name == "mat10000" else 20000) or "trsv" in k) and k in out:
                    acc[k].append(float(r["Counter_Value"]))
            for k, v in acc.items():
                out[k][cn + "_KB_avg"] = sum(v) / len(v)
        for k, v in out.items():
            if isinstance(v, dict) and "FETCH_SIZE_KB_avg" in v:
                v["hbm_bytes_per_launch_corrected"] = 2 * v["FETCH_SIZE_KB_avg"] * 1024 + v.get("WRITE_SIZE_KB_avg", 0) * 1024
        json.dump(out, open(os.path.join(dst, "pmc_fetch_write.json"), "w"), indent=1)
        print(dst)
        for k, v in sorted(out.items()):
            if isinstance(v, dict):
                print("  %-28s n=%3d avg %9.1f us  hbm %s" % (k[:28], v["launches"], v["avg_us"],
                      "%.3f GB" % (v["hbm_bytes_per_launch_corrected"] / 1e9) if "hbm_bytes_per_launch_corrected" in v else "-"))
if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r03")
