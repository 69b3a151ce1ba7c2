"""ad-hoc: lanes-per-row CSR vs blocked SpMV across matrix sizes (square, random columns)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cuda_mat_amd as cm
ctx = cm.Context(0)
for n, per in [(100_000, 50), (250_000, 50), (500_000, 50), (1_000_000, 50), (2_000_000, 50), (1_000_000, 16), (4_000_000, 16), (4_000_000, 8)]:
    nnz = n * per
    rp, ci, va = ctx.empty(n + 1, np.int32), ctx.empty(nnz, np.int32), ctx.empty(nnz)
    ctx.gen_rand_rows(n, per, 0x5EED, 0, n, 0, rp, ci, va)
    x, y = ctx.empty(n), ctx.empty(n)
    ctx.gen_xstar(0, n, 3, x)
    res = {}
    for mode in ("csr", "pb", "auto"):
        if mode == "auto":
            os.environ.pop("CUDAMAT_SPMV_MODE", None)
        else:
            os.environ["CUDAMAT_SPMV_MODE"] = mode
        s = cm.Solver(ctx, n, n, nnz, rp, ci, va, 0)
        s.spmv(x, y)
        t = ctx.timer(); ctx.sync(); t.start()
        for _ in range(10):
            s.spmv(x, y)
        t.stop()
        res[mode] = (t.elapsed_ms() / 10, s.spmv_mode())
        s.close()
    print("n=%8d per=%2d  csr %7.3f ms  pb %7.3f ms  auto -> %s %7.3f ms" % (n, per, res["csr"][0], res["pb"][0], "pb" if res["auto"][1] else "csr", res["auto"][0]), flush=True)
    for a in (rp, ci, va, x, y):
        a.free()
