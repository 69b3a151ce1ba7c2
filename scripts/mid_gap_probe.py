"""GPU box: mid-size Poisson systems, five launches per iteration -- wall time per iteration with and without the per-SpMV
event records of CUDAMAT_FLAG_PROFILE (what bench.py times with), against the kernels' own time: is the loop bound by the
host's enqueue rate or by the GPU?   usage: python scripts/mid_gap_probe.py [rows ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import cuda_mat_amd as cm

sizes = [int(a) for a in sys.argv[1:]] or [100_000, 400_000, 1_000_000, 3_000_000]
ctx = cm.Context(0)
ctx.set_option("VALUE_DICT", "0")
for n in sizes:
    nx = 1000
    ny = n // nx
    n = nx * ny
    rp = ctx.empty(n + 1, np.int32); ci = ctx.empty(5 * n, np.int32); va = ctx.empty(5 * n, np.float64)
    ctx.gen_poisson5(nx, ny, 0, n, 0, rp, ci, va)
    ctx.sync()
    nnz = int(rp.download()[-1])
    s = cm.Solver(ctx, n, n, nnz, rp, ci, va, 0)
    xs = ctx.empty(n); ctx.gen_xstar(0, n, 0x5EEE, xs)
    b = ctx.empty(n); s.spmv(xs, b)
    x = ctx.empty(n)
    out = []
    for fl in (0, cm.FLAG_PROFILE):
        best = 1e9
        for rep in range(4):
            ctx.sync()
            t0 = time.perf_counter()
            st = s.solve(b, x, maxit=300, tol=1e-30, flags=cm.FLAG_NO_EXIT | cm.FLAG_X0_ONES | fl)
            ctx.sync()
            dt = time.perf_counter() - t0
            best = min(best, dt / max(st.iters, 1))
        out.append(best * 1e6)
    print("rows %8d  %-16s  %7.1f us/iteration plain   %7.1f with event records   (loop form %d)" % (n, s.spmv_kernel()[:16], out[0], out[1], st.loop_form), flush=True)
    s.close()
    for a in (rp, ci, va, xs, b, x):
        a.free()
