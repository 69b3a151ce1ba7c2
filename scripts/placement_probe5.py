"""EXPERIMENT: the product stream P allocated behind a ballast of G GB (CUDAMAT_DEBUG_P_BALLAST_GB), i.e. elsewhere in device
memory than the arrays phase 1 reads.  Fresh process per G (the driver's state is the same each time)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    import numpy as np
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ.setdefault("CUDAMAT_VALUE_DICT", "0")
    import cuda_mat_amd as cm
    from placement_probe import timed
    ctx = cm.Context(0)
    n, per = 10_000_000, 50
    rp, ci, va = ctx.empty(n + 1, np.int32), ctx.empty(n * per, np.int32), ctx.empty(n * per)
    ctx.gen_rand_rows(n, per, 7, 0, n, 0, rp, ci, va)
    xs = ctx.empty(n)
    ctx.gen_xstar(0, n, 8, xs)
    out = []
    for rep in range(3):
        s = cm.Solver(ctx, n, n, n * per, rp, ci, va, 0)
        b, x = ctx.empty(n), ctx.empty(n)
        s.spmv(xs, b)
        step_ms, spmv_ms = timed(ctx, s, b, x)
        out.append("%.1f it/s (spmv %.3f)" % (1e3 / step_ms, spmv_ms))
        b.free()
        x.free()
        s.close()
        cm.lib().cudamat_pool_trim()
    print("P ballast %3s GB: %s" % (os.environ.get("CUDAMAT_DEBUG_P_BALLAST_GB", "0"), " | ".join(out)), flush=True)
else:
    for g in (0, 30, 50, 70, 90, 120, 0, 70):
        env = dict(os.environ, CUDAMAT_DEBUG_P_BALLAST_GB=str(g))
        subprocess.run([sys.executable, os.path.abspath(__file__), "--one"], env=env, check=False)
