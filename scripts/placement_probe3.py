"""Which launch differs between a fast and a slow instance of the solver (placement_probe.py)?  Six instances in one process,
each timed on the loop; run under `rocprofv3 --kernel-trace`, the trace is cut at the marker launches (k_gen_xstar) between
instances and the per-kernel medians are printed per instance.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_place -- python3 scripts/placement_probe3.py
    python3 scripts/placement_probe3.py --summarize gpurun_out/trace_place
"""
import collections
import csv
import glob
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("CUDAMAT_VALUE_DICT", "0")


def summarize(d):
    f = glob.glob(d + "/*/*kernel_trace.csv")
    rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("cm::", "")) for r in csv.DictReader(open(f[0])))
    inst, acc = -1, collections.defaultdict(lambda: collections.defaultdict(list))
    for s, e, k in rows:
        if k.startswith("k_gen_xstar"):
            inst += 1
            continue
        if inst >= 0:
            acc[inst][k[:28]].append((e - s) / 1e3)
    names = ["k_pb_phase1", "k_pb_phase2<16, 64, 4>", "k_full<1>", "k_update_p<1>", "k_half<1>"]
    for i in sorted(acc):
        parts = []
        for k in names:
            v = sorted(x for x in acc[i].get(k[:28], []) if x > 20)
            if v:
                parts.append("%s %.1f" % (k.split("<")[0], v[len(v) // 2]))
        print("instance %d (trace): %s us" % (i, "  ".join(parts)))


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--summarize":
        return summarize(sys.argv[2])
    import cuda_mat_amd as cm
    from placement_probe import timed
    ctx = cm.Context(0)
    n, per = 10_000_000, 50
    rp, ci, va = ctx.empty(n + 1, np.int32), ctx.empty(n * per, np.int32), ctx.empty(n * per)
    ctx.gen_rand_rows(n, per, 7, 0, n, 0, rp, ci, va)
    xs = ctx.empty(n)
    for i in range(6):
        ctx.gen_xstar(0, n, 8, xs)          # the marker
        s = cm.Solver(ctx, n, n, n * per, rp, ci, va, 0)
        b, x = ctx.empty(n), ctx.empty(n)
        s.spmv(xs, b)
        step_ms, spmv_ms = timed(ctx, s, b, x)
        print("instance %d (events): %.3f ms/step (%.1f it/s)  spmv %.3f ms" % (i, step_ms, 1e3 / step_ms, spmv_ms), flush=True)
        b.free()
        x.free()
        s.close()
        cm.lib().cudamat_pool_trim()


if __name__ == "__main__":
    main()
