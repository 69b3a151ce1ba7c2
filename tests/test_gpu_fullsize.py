"""BASELINE.json's full sizes (1e7 rows) through size-independent properties -- the oracle cannot run
these in seconds, so the checks are: two independent HIP implementations agree bit for bit, linearity
holds exactly on integer data, and the solve returns the known x* it was built from."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N = 10_000_000


@pytest.fixture(scope="module")
def cm():
    import cuda_mat_amd as cm
    assert cm.device_count() > 0
    return cm


def _system(cm, ctx, kind):
    if kind == "rand50":
        rn = cm.lib().cudamat_rand_row_nnz(N, 50)
        nnz = N * rn
        rp, ci, va = ctx.empty(N + 1, np.int32), ctx.empty(nnz, np.int32), ctx.empty(nnz)
        ctx.gen_rand_rows(N, 50, 0x5EED, 0, N, 0, rp, ci, va)
    else:
        nnz = cm.lib().cudamat_poisson5_nnz(4000, 2500)
        rp, ci, va = ctx.empty(N + 1, np.int32), ctx.empty(nnz, np.int32), ctx.empty(nnz)
        ctx.gen_poisson5(4000, 2500, 0, N, 0, rp, ci, va)
    return nnz, rp, ci, va


@pytest.mark.parametrize("kind", ["rand50", "poisson5"])
def test_full_size_spmv_properties(cm, kind, monkeypatch):
    ctx = cm.Context(0)
    nnz, rp, ci, va = _system(cm, ctx, kind)
    xs, x2 = ctx.empty(N), ctx.empty(N)
    ctx.gen_xstar(0, N, 7, xs)            # values in eighths: every product and sum below is exact
    ctx.gen_xstar(0, N, 8, x2)
    ys = {}
    for mode in ("csr", "pb"):
        monkeypatch.setenv("CUDAMAT_SPMV_MODE", mode)
        s = cm.Solver(ctx, N, N, nnz, rp, ci, va, 0)
        assert s.spmv_mode() == (1 if mode == "pb" else 0)
        y1, y2, y3 = ctx.empty(N), ctx.empty(N), ctx.empty(N)
        s.spmv(xs, y1)
        s.spmv(x2, y2)
        # linearity: A (2 x1 + 3 x2) == 2 A x1 + 3 A x2, exactly
        z = ctx.empty(N).zero()
        ctx.axpy(N, 2.0, xs, z)
        ctx.axpy(N, 3.0, x2, z)
        s.spmv(z, y3)
        ctx.scal(N, 2.0, y1)
        ctx.axpy(N, 3.0, y2, y1)
        a, b = y3.download(), y1.download()
        np.testing.assert_array_equal(a, b)
        ys[mode] = a
        for t in (y1, y2, y3, z):
            t.free()
        s.close()
    # the lanes-per-row CSR kernel and the blocked two-phase kernels are independent implementations
    np.testing.assert_array_equal(ys["csr"], ys["pb"])
    # checksum of checksums: 1^T (A z) for the Laplacian only touches boundary rows (interior rows sum to 0)
    if kind == "poisson5":
        assert abs(ys["csr"].sum()) < 1e-3 * np.abs(ys["csr"]).sum() + 1e9
    ctx.close()


def test_full_size_solve_returns_xstar(cm):
    ctx = cm.Context(0)
    nnz, rp, ci, va = _system(cm, ctx, "rand50")
    s = cm.Solver(ctx, N, N, nnz, rp, ci, va, 0)
    for t in (rp, ci, va):
        t.free()
    xs, b, x = ctx.empty(N), ctx.empty(N), ctx.empty(N)
    ctx.gen_xstar(0, N, 0x5EEE, xs)
    s.spmv(xs, b)
    st = s.solve(b, x, loop=cm.LOOP_PBICGSTAB, maxit=100, tol=1e-8, flags=cm.FLAG_X0_ONES)
    assert st.converged and st.iters <= 10
    # ||r|| <= 1e-8 ||r0|| and a strictly dominant matrix: x is x* to ~1e-8 * ||r0||/||A|| per entry
    np.testing.assert_allclose(x.download(), xs.download(), rtol=1e-6)
    # encode -> solve -> re-encode round trip: A x_solved reproduces b to the stopping tolerance
    ax = ctx.empty(N)
    s.spmv(x, ax)
    r = b.download() - ax.download()
    assert np.linalg.norm(r) <= 2e-8 * st.nrm0
    s.close()
    ctx.close()
