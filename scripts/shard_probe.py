"""ad-hoc: time the SpMV kernels on shard-shaped blocks (n/G rows x n columns) of the 1e7 x 50 matrix on ONE GPU"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cuda_mat_amd as cm
N = 10_000_000
ctx = cm.Context(0)
for G in (1, 2, 4, 8):
    nloc = N // G
    rn = 50
    nnz = nloc * rn
    rp, ci, va = ctx.empty(nloc + 1, np.int32), ctx.empty(nnz, np.int32), ctx.empty(nnz)
    ctx.gen_rand_rows(N, 50, 0x5EED, 0, nloc, 0, rp, ci, va)
    x, y = ctx.empty(N), ctx.empty(nloc)
    ctx.gen_xstar(0, N, 3, x)
    for mode in ("csr", "pb"):
        os.environ["CUDAMAT_SPMV_MODE"] = mode
        s = cm.Solver(ctx, nloc, N, nnz, rp, ci, va, 0)
        s.spmv(x, y)
        t = ctx.timer()
        ctx.sync()
        t.start()
        for _ in range(10):
            s.spmv(x, y)
        t.stop()
        ms = t.elapsed_ms() / 10
        b = 12.0 * nnz + 4 * (nloc + 1) + 8.0 * N + 8.0 * nloc
        print("G=%d rows=%8d mode=%-3s  %7.3f ms   %7.1f GB/s algorithmic" % (G, nloc, mode, ms, b / ms * 1e-6), flush=True)
        s.close()
    for a in (rp, ci, va, x, y):
        a.free()
