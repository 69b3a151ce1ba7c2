"""Does the ABSOLUTE position of the solver's arrays in device memory matter?  A ballast of G GB is allocated first (and kept),
then the solver is created twice (the alternation of placement_probe.py) and timed.

    python scripts/placement_probe4.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("CUDAMAT_VALUE_DICT", "0")
import cuda_mat_amd as cm  # noqa: E402
from placement_probe import timed  # noqa: E402


def main():
    ctx = cm.Context(0)
    n, per = 10_000_000, 50
    rp, ci, va = ctx.empty(n + 1, np.int32), ctx.empty(n * per, np.int32), ctx.empty(n * per)
    ctx.gen_rand_rows(n, per, 7, 0, n, 0, rp, ci, va)
    xs = ctx.empty(n)
    ctx.gen_xstar(0, n, 8, xs)
    for g in (0, 1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128, 0):
        ballast = [ctx.empty((1 << 30) // 8) for _ in range(g)]          # g blocks of 1 GB, kept while the instances live
        out = []
        for rep in range(2):
            s = cm.Solver(ctx, n, n, n * per, rp, ci, va, 0)
            b, x = ctx.empty(n), ctx.empty(n)
            s.spmv(xs, b)
            step_ms, spmv_ms = timed(ctx, s, b, x)
            out.append("%.3f ms/step (%.1f it/s, spmv %.3f)" % (step_ms, 1e3 / step_ms, spmv_ms))
            b.free()
            x.free()
            s.close()
            for a in ballast:
                a.free()
            cm.lib().cudamat_pool_trim()
            ballast = [ctx.empty((1 << 30) // 8) for _ in range(g)]
        for a in ballast:
            a.free()
        cm.lib().cudamat_pool_trim()
        print("ballast %3d GB: %s | %s" % (g, out[0], out[1]), flush=True)


if __name__ == "__main__":
    main()
