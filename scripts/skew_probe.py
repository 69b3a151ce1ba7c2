"""ad-hoc: SpMV on skewed row-length distributions: power-law rows, a few huge rows, banded + hubs"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cuda_mat_amd as cm

ctx = cm.Context(0)
rng = np.random.default_rng(5)


def build(n, lens, local=False):
    lens = np.minimum(lens.astype(np.int64), n // 2)
    rp = np.zeros(n + 1, np.int64)
    np.cumsum(lens, out=rp[1:])
    nnz = int(rp[-1])
    if local:      # columns near the diagonal (banded-ish) so x gathers hit cache: isolates the balance problem
        row = np.repeat(np.arange(n), lens)
        off = np.arange(nnz) - np.repeat(rp[:-1], lens)
        ci = (row + off - np.repeat(lens // 2, lens)) % n
    else:
        ci = rng.integers(0, n, nnz)
    return rp.astype(np.int32), ci.astype(np.int32), np.ones(nnz), nnz


def run(name, n, lens, local):
    rp, ci, va, nnz = build(n, lens, local)
    # columns need not be sorted/distinct for timing the CSR kernel; the blocked form wants sorted rows -> csr only
    os.environ["CUDAMAT_SPMV_MODE"] = "csr"
    d_rp, d_ci, d_va = ctx.array(rp), ctx.array(ci), ctx.array(va)
    x, y = ctx.empty(n), ctx.empty(n)
    ctx.gen_xstar(0, n, 3, x)
    gb = (12.0 * nnz + 20.0 * n) / 1e9
    res = []
    for form in ("lanes", "tiles", None):
        if form:
            os.environ["CUDAMAT_SPMV_FORM"] = form
        else:
            os.environ.pop("CUDAMAT_SPMV_FORM", None)
        s = cm.Solver(ctx, n, n, nnz, d_rp, d_ci, d_va, 0)
        s.spmv(x, y)
        t = ctx.timer(); ctx.sync(); t.start()
        for _ in range(10):
            s.spmv(x, y)
        t.stop()
        res.append(t.elapsed_ms() / 10)
        s.close()
    print("%-36s nnz=%10d mean=%6.1f max=%7d | lanes %7.3f ms %6.0f GB/s | tiles %7.3f ms %6.0f GB/s | auto %7.3f ms" %
          (name, nnz, nnz / n, lens.max(), res[0], gb / res[0] * 1e3, res[1], gb / res[1] * 1e3, res[2]), flush=True)
    for a in (d_rp, d_ci, d_va, x, y):
        a.free()


n = 2_000_000
for local in (True, False):
    tag = "local" if local else "random"
    run("uniform 16 (%s)" % tag, n, np.full(n, 16), local)
    run("pareto a=1.5 mean~16 (%s)" % tag, n, np.minimum(1 + (rng.pareto(1.5, n) * 5.5), 200000), local)
    z = np.full(n, 8); z[rng.integers(0, n, 2000)] = 8000
    run("8 + 2000 rows of 8000 (%s)" % tag, n, z, local)
    z = np.full(n, 8); z[rng.integers(0, n, 20)] = 400000
    run("8 + 20 rows of 400000 (%s)" % tag, n, z, local)
    z = np.where(np.arange(n) % 2 == 0, 2, 62)
    run("alternating 2 / 62 (%s)" % tag, n, z, local)
    z = np.where(np.arange(n) < n // 2, 2, 62)
    run("first half 2, second half 62 (%s)" % tag, n, z, local)
