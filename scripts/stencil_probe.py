"""ad-hoc: banded matrices with long rows (27-point stencil, 125-point stencil): CSR kernels vs blocked"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
import cuda_mat_amd as cm
ctx = cm.Context(0)
def stencil(m, w):
    e = np.ones(m); d = sp.diags([e] * (2 * w + 1), list(range(-w, w + 1)), shape=(m, m))
    A = sp.kron(sp.kron(d, d), d).tocsr(); A.sort_indices(); return A
for m, w in [(100, 1), (160, 1), (64, 2)]:
    A = stencil(m, w); n = A.shape[0]; nnz = A.nnz
    rp, ci, va = ctx.array(A.indptr.astype(np.int32)), ctx.array(A.indices.astype(np.int32)), ctx.array(A.data)
    x, y = ctx.array(np.random.default_rng(0).integers(-4, 5, n).astype(np.float64)), ctx.empty(n)
    ref = None
    for mode in ("csr", "pb", "auto"):
        if mode == "auto": os.environ.pop("CUDAMAT_SPMV_MODE", None)
        else: os.environ["CUDAMAT_SPMV_MODE"] = mode
        s = cm.Solver(ctx, n, n, nnz, rp, ci, va, 0)
        s.spmv(x, y)
        got = y.download()
        if ref is None: ref = A @ x.download()
        assert np.array_equal(got, ref), mode
        t = ctx.timer(); ctx.sync(); t.start()
        for _ in range(10): s.spmv(x, y)
        t.stop(); ms = t.elapsed_ms() / 10
        print("stencil m=%d w=%d n=%d nnz/row=%.1f mode=%-4s -> %s %7.3f ms  %6.0f GB/s alg" % (m, w, n, nnz / n, mode, "pb" if s.spmv_mode() else "csr", ms, (12.0 * nnz + 20.0 * n) / ms * 1e-6), flush=True)
        s.close()
