"""GPU box: what does hipMalloc / hipFree of multi-GB buffers cost?  (setup steps of the preconditioned path allocate and
free such buffers many times)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cuda_mat_amd as cm
ctx = cm.Context(0)
for gb in (0.5, 2.0, 4.0, 8.0):
    n = int(gb * 1e9 / 8)
    ts = []
    for rep in range(4):
        t0 = time.perf_counter(); a = ctx.empty(n); ctx.sync(); t1 = time.perf_counter()
        a.zero(); ctx.sync(); t2 = time.perf_counter()
        a.zero(); ctx.sync(); t3 = time.perf_counter()
        a.free(); ctx.sync(); t4 = time.perf_counter()
        ts.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3))
    print("%.1f GB: " % gb + "  ".join("malloc %.1f ms, first touch %.1f, second %.1f, free %.1f" % tuple(1e3 * v for v in t) for t in ts[1:2]),
          "| all mallocs:", ["%.1f" % (1e3 * t[0]) for t in ts], "frees:", ["%.1f" % (1e3 * t[3]) for t in ts])
