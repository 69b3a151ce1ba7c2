"""Does the speed of the judged loop depend on WHERE the solver's arrays were allocated?  One process, one matrix; the solver
(blocked copy, product stream, work vectors) is created several times on fresh device memory (pool trimmed in between) and
the same 100 iterations are timed on each instance; then the last instance is timed three more times without reallocating.

    python scripts/placement_probe.py [instances]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("CUDAMAT_VALUE_DICT", "0")
import cuda_mat_amd as cm  # noqa: E402


def timed(ctx, s, b, x, steps=100):
    fl = cm.FLAG_NO_EXIT | cm.FLAG_X0_ONES
    s.solve(b, x, maxit=10, tol=1e-8, flags=fl)
    ctx.sync()
    t0 = time.perf_counter()
    ms, nl = 0.0, 0
    for _ in range(steps // 50):
        st = s.solve(b, x, maxit=50, tol=1e-8, flags=fl | cm.FLAG_PROFILE)
        ms += st.ms_spmv
        nl += st.n_spmv
    ctx.sync()
    dt = time.perf_counter() - t0
    return dt / steps * 1e3, ms / max(nl, 1)


PADS_MB = [0, 2, 4, 6, 8, 10, 16, 32, 64, 128, 256, 512, 1024, 0, 22, 0]


def main():
    k = int(sys.argv[1]) if len(sys.argv) > 1 else len(PADS_MB)
    ctx = cm.Context(0)
    n, per = 10_000_000, 50
    rp, ci, va = ctx.empty(n + 1, np.int32), ctx.empty(n * per, np.int32), ctx.empty(n * per)
    ctx.gen_rand_rows(n, per, 7, 0, n, 0, rp, ci, va)
    xs = ctx.empty(n)
    ctx.gen_xstar(0, n, 8, xs)
    for i in range(k):
        pad = None
        mb = PADS_MB[i % len(PADS_MB)]
        if mb:
            pad = ctx.empty((mb << 20) // 8)      # shifts what follows in the driver's address space
        s = cm.Solver(ctx, n, n, n * per, rp, ci, va, 0)
        b, x = ctx.empty(n), ctx.empty(n)
        s.spmv(xs, b)
        step_ms, spmv_ms = timed(ctx, s, b, x)
        print("instance %d (pad %d MB): %.3f ms/step (%.1f it/s)  spmv %.3f ms   b at %#x" % (i, mb, step_ms, 1e3 / step_ms, spmv_ms, b.ptr), flush=True)
        if i == k - 1:
            for r in range(3):
                step_ms, spmv_ms = timed(ctx, s, b, x)
                print("   same instance again: %.3f ms/step (%.1f it/s)  spmv %.3f ms" % (step_ms, 1e3 / step_ms, spmv_ms), flush=True)
        for a in (b, x):
            a.free()
        s.close()
        if pad is not None:
            pad.free()
        cm.lib().cudamat_pool_trim()


if __name__ == "__main__":
    main()
