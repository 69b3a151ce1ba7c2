"""BASELINE.json's full sizes (1e7 rows).

SpMV (the kernel the bench times) is compared with the oracle's MatrixVectorMult restatement
(bicstab_omp/bicstab.cpp:69-80) ON THE FULL MATRIX: the oracle builds the 6 GB system in 2-3 s and multiplies it
in ~0.2 s with all host threads, so the comparison is affordable -- bit for bit, on both value forms the library
has (fp64 values = what bench.py times, CUDAMAT_VALUE_DICT=0; 8-bit value indices = the library's default for
these generators).  What the oracle cannot do in seconds at this size (ILU(0), whole solves) is checked through
size-independent properties: two independent HIP implementations agree bit for bit, linearity holds exactly on
integer data, sampled rows re-eliminated on the host, and the solve returns the x* it was built from."""

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


def _host_free_gb():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                return int(line.split()[1]) / 1e6
    except OSError:
        pass
    return 0.0


def _bits(a):
    return np.ascontiguousarray(a).view(np.int64)


# which kernel(s) a trace of one SpMV shows, per workload and value form (what bench.py's roofline.kernel carries).  For the
# Poisson matrix the tuner TIMES the compressed stream kernel against the row-pattern form and keeps the faster: on fp64
# values that is the row-pattern form (0.113 against 0.149 ms), with the dictionary the stream kernel (0.095 against 0.109).
_KERNELS = {("rand50", "fp64"): ("k_pb_phase1 + k_pb_phase2",), ("rand50", "dict"): ("k_pb_phase1_dict + k_pb_phase2",),
            ("poisson5", "fp64"): ("k_spmv_pat<", "k_spmv_stream_c<"), ("poisson5", "dict"): ("k_spmv_stream_d<", "k_spmv_pat_d<")}


def _values_env(monkeypatch, values):
    if values == "fp64":
        monkeypatch.setenv("CUDAMAT_VALUE_DICT", "0")          # the kernels bench.py times
    else:
        monkeypatch.delenv("CUDAMAT_VALUE_DICT", raising=False)


@pytest.mark.parametrize("values", ["fp64", "dict"])
@pytest.mark.parametrize("kind", ["rand50", "poisson5"])
def test_full_size_spmv_vs_oracle(cm, kind, values, monkeypatch):
    """BASELINE configs[2] / configs[3] at 1e7 rows: the SpMV form the library selects by itself (rand50: blocked
    two-phase; poisson5: compressed stream tiles) against the oracle's MatrixVectorMult (bicstab.cpp:69-80) on the
    host-built full matrix.  values = "fp64" is the form bench.py TIMES (it sets CUDAMAT_VALUE_DICT=0: k_pb_phase1 /
    k_spmv_stream_c), "dict" the library's default for these generators (39 / 2 distinct values: k_pb_phase1_dict /
    k_spmv_stream_d).  Bit for bit on integer-valued x (any summation order is exact) AND on real-valued x: both forms
    add a row's products in column order with one rounding per product and per addition, the rounding sequence of the
    reference loop `b[i] += A.Value[j] * x[A.Col[j]]`.  The lanes-per-row kernel (another summation order) is held to
    SURVEY 8c's 4 nnz_row eps sum|a_ij x_j|.  Also pins the device generators to the oracle's at full size."""
    import os
    from oracle import oracle as O
    need = 20.0 if kind == "rand50" else 6.0
    if _host_free_gb() < need:
        pytest.skip("needs %.0f GB of free host memory for the oracle's copy of the full matrix (have %.1f)" % (need, _host_free_gb()))
    _values_env(monkeypatch, values)
    O.set_num_threads(min(32, len(os.sched_getaffinity(0))))
    A = O.rand_rows(N, 50, 0x5EED) if kind == "rand50" else O.poisson5(4000, 2500)
    ctx = cm.Context(0)
    nnz, rp, ci, va = _system(cm, ctx, kind)
    assert nnz == A.nnz
    # the device generator is the oracle's generator, entry for entry, at the full size
    np.testing.assert_array_equal(rp.download(), A.rowptr)
    np.testing.assert_array_equal(ci.download()[:nnz], A.colidx[:nnz])
    assert np.array_equal(_bits(va.download()[:nnz]), _bits(A.val[:nnz]))
    rng = np.random.default_rng(0xC4)
    x_int = O.xstar(N, 7)                                     # eighths in [1, 2): products and sums are exact
    x_real = rng.standard_normal(N) * np.exp(rng.uniform(-3, 3, N))
    want = {id(x_int): O.spmv(A, x_int), id(x_real): O.spmv(A, x_real)}
    d_x, d_y = ctx.empty(N), ctx.empty(N)
    # the form the library selects by itself, then (Poisson) each of the two candidates forced: all bit for bit
    forms = [None] + (["csr", "pat"] if kind == "poisson5" else [])
    seen = []
    for form in forms:
        ctx.reset_options()
        if form:
            ctx.set_option("SPMV_MODE", form)
        s = cm.Solver(ctx, N, N, nnz, rp, ci, va, 0)
        name = s.spmv_kernel()
        seen.append(name)
        assert name.startswith(_KERNELS[(kind, values)]), name
        if form == "csr":
            assert name.startswith("k_spmv_stream_c<" if values == "fp64" else "k_spmv_stream_d<")
        if form == "pat":
            assert name.startswith("k_spmv_pat<" if values == "fp64" else "k_spmv_pat_d<")
        assert (s.value_dict() > 0) == (values == "dict")
        for x in (x_int, x_real):
            d_x.upload(x)
            s.spmv(d_x, d_y)
            got = d_y.download()
            assert np.array_equal(_bits(got), _bits(want[id(x)])), "%s: max |diff| = %g" % (name, np.abs(got - want[id(x)]).max())
        s.close()
    ctx.reset_options()
    print("full-size %s / %s values: selected %s" % (kind, values, seen[0]))
    if kind == "rand50" and values == "fp64":
        # the lanes-per-row CSR kernel on the same matrix -- the plan's own choice for 50 entries per row (32 lanes) and
        # north_star's literal one-wavefront-per-row form (64 lanes): exact on integer data, SURVEY 8c's bound otherwise
        monkeypatch.setenv("CUDAMAT_SPMV_MODE", "csr")          # (read by the contexts created below)
        absA = O.Csr(A.n, A.rowptr, A.colidx, np.abs(A.val), A.m)
        bound = 4 * 50 * np.finfo(float).eps * O.spmv(absA, np.abs(x_real))
        want_int, want_real = O.spmv(A, x_int), O.spmv(A, x_real)
        for lanes in (None, "64"):
            if lanes:
                monkeypatch.setenv("CUDAMAT_SPMV_LANES", lanes)
            c2 = cm.Context(0)
            s = cm.Solver(c2, N, N, nnz, rp, ci, va, 0)
            assert s.spmv_kernel() == ("k_spmv<%s>" % (lanes or "32")), s.spmv_kernel()
            d_x.upload(x_int)
            s.spmv(d_x, d_y)
            c2.sync()                       # (the vectors belong to ctx: its stream knows nothing of c2's)
            assert np.array_equal(_bits(d_y.download()), _bits(want_int))
            d_x.upload(x_real)
            s.spmv(d_x, d_y)
            c2.sync()
            assert np.all(np.abs(d_y.download() - want_real) <= bound)
            s.close()
            c2.close()
    for t in (rp, ci, va, d_x, d_y):
        t.free()
    ctx.close()


@pytest.mark.parametrize("values", ["fp64", "dict"])
@pytest.mark.parametrize("kind", ["rand50", "poisson5"])
def test_full_size_spmv_properties(cm, kind, values, monkeypatch):
    _values_env(monkeypatch, values)
    ctx = cm.Context(0)
    nnz, rp, ci, va = _system(cm, ctx, kind)
    xs, x2 = ctx.empty(N), ctx.empty(N)
    ctx.gen_xstar(0, N, 7, xs)            # values in eighths: every product and sum below is exact
    ctx.gen_xstar(0, N, 8, x2)
    ys = {}
    for mode in ("csr", "pb"):
        ctx.set_option("SPMV_MODE", mode)
        s = cm.Solver(ctx, N, N, nnz, rp, ci, va, 0)
        assert s.spmv_mode() == (1 if mode == "pb" else 0)
        # the name the bench line's roofline.kernel carries = the kernel(s) a trace of this SpMV shows
        suffix = "_dict" if values == "dict" else ""
        assert s.spmv_kernel().startswith("k_pb_phase1%s + " % suffix if mode == "pb" else
                                          ("k_spmv_stream_d<" if values == "dict" else "k_spmv_stream_c<") if kind == "poisson5" else "k_spmv<"), s.spmv_kernel()
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


@pytest.mark.parametrize("values", ["fp64", "dict"])
def test_full_size_solve_returns_xstar(cm, values, monkeypatch):
    _values_env(monkeypatch, values)
    ctx = cm.Context(0)
    nnz, rp, ci, va = _system(cm, ctx, "rand50")
    s = cm.Solver(ctx, N, N, nnz, rp, ci, va, 0)
    assert s.spmv_kernel() == _KERNELS[("rand50", values)][0], s.spmv_kernel()
    assert (s.value_dict() > 0) == (values == "dict")
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


def test_full_size_poisson_solve_200_iterations(cm):
    """BASELINE configs[2] as a SOLVE (4000 x 2500 five-point Laplacian, no preconditioner): a fixed window of 200
    iterations (tol = 0 never fires); the residual the loop carries by recurrence (pbicgstab.cu:139-142) must still
    be the true residual b - A x of the iterate it returns, and the iterate must have made progress."""
    ctx = cm.Context(0)
    nnz, rp, ci, va = _system(cm, ctx, "poisson5")
    s = cm.Solver(ctx, N, N, nnz, rp, ci, va, 0)
    for t in (rp, ci, va):
        t.free()
    xs, b, x, ax = ctx.empty(N), ctx.empty(N), ctx.empty(N), ctx.empty(N)
    ctx.gen_xstar(0, N, 0x5EEE, xs)
    s.spmv(xs, b)
    st = s.solve(b, x, loop=cm.LOOP_PBICGSTAB, maxit=200, tol=0.0, flags=cm.FLAG_X0_ONES)
    assert st.iters == 200 and not st.converged
    hist = s.history()
    assert len(hist) == 400 and np.all(np.isfinite(hist))
    s.spmv(x, ax)
    r_true = np.linalg.norm(b.download() - ax.download())
    # the recurrence drifts from the true residual by rounding only: O(eps * iterations * ||A|| ||x||) ~ 1e-9 ||r0|| here
    assert abs(r_true - st.nrm) <= 1e-7 * st.nrm0, (r_true, st.nrm, st.nrm0)
    assert st.nrm == hist[-1]
    assert r_true < 0.2 * st.nrm0
    s.close()
    ctx.close()


def _host_row(h_rp, h_ci, vals, i):
    a, e = int(h_rp[i]), int(h_rp[i + 1])
    return h_ci[a:e], vals[a:e]


def test_full_size_ilu0_c5(cm):
    """BASELINE configs[4]: the 1e7 x 50 matrix with ILU(0) on one GPU (pbicgstab.cu:336-374).  The oracle cannot
    factor 5e8 entries in seconds, so: (1) sampled rows of the factor are re-eliminated on the host from A's row and
    the GPU's final pivot rows (the IKJ definition of csrilu0, row by row); (2) the dependency-driven and the
    level-by-level triangular solves agree bit for bit; (3) L (U out) reproduces the right-hand side on sampled rows;
    (4) the preconditioned solve returns x*, with the true residual under the stopping tolerance, without any
    fallback."""
    rng = np.random.default_rng(5)
    ctx = cm.Context(0)
    nnz, rp, ci, va = _system(cm, ctx, "rand50")
    h_rp, h_ci, h_va = rp.download().astype(np.int64), ci.download(), va.download()
    s = cm.Solver(ctx, N, N, nnz, rp, ci, va, 0)
    for t in (rp, ci, va):
        t.free()
    s.ilu0()
    assert s.trsv_form() == 1
    lu = s.ilu0_values()
    assert lu.shape == (nnz,)

    def upper_of(k):                       # (columns, values) of row k from its diagonal on
        c, v = _host_row(h_rp, h_ci, lu, k)
        d = int(np.searchsorted(c, k))
        assert c[d] == k
        return c[d:], v[d:]

    # (1) IKJ re-elimination of sampled rows
    rows = np.unique(np.concatenate([rng.integers(0, N, 1500), [0, 1, N - 2, N - 1]]))
    worst = 0.0
    for i in rows:
        c, w = _host_row(h_rp, h_ci, h_va, i)
        w = w.copy()
        for pos in range(int(np.searchsorted(c, i))):          # pivots k < i in increasing order
            k = int(c[pos])
            uc, uv = upper_of(k)
            w[pos] = w[pos] / uv[0]
            hit = np.searchsorted(c, uc[1:])
            ok = (hit < len(c)) & (c[np.minimum(hit, len(c) - 1)] == uc[1:])
            w[hit[ok]] -= w[pos] * uv[1:][ok]
        got = _host_row(h_rp, h_ci, lu, i)[1]
        worst = max(worst, float(np.max(np.abs(got - w) / np.maximum(np.abs(w), 1e-300))))
    assert worst <= 1e-12, worst

    # (2) + (3) triangular solves
    rhs, out1, out0 = ctx.empty(N), ctx.empty(N), ctx.empty(N)
    ctx.gen_xstar(0, N, 11, rhs)
    s.precond_apply(rhs, out1)
    h_out, h_rhs = out1.download(), rhs.download()
    worst = 0.0
    for i in rows[::4]:
        c, v = _host_row(h_rp, h_ci, lu, i)
        d = int(np.searchsorted(c, i))
        acc = 0.0
        for pos in range(d + 1):                                # (L w)_i with w = U out, unit diagonal
            k = int(c[pos])
            uc, uv = upper_of(k)
            wk = float(np.dot(uv, h_out[uc]))
            acc += wk if pos == d else v[pos] * wk
        worst = max(worst, abs(acc - h_rhs[i]) / abs(h_rhs[i]))
    assert worst <= 1e-10, worst
    ctx.set_option("TRSV_SYNCFREE", "0")
    s.close()
    del lu
    # the level-by-level form needs its own plans: a second solver over the same system
    rp, ci, va = ctx.array(h_rp.astype(np.int32)), ctx.array(h_ci), ctx.array(h_va)
    s0 = cm.Solver(ctx, N, N, nnz, rp, ci, va, 0)
    for t in (rp, ci, va):
        t.free()
    s0.ilu0()
    assert s0.trsv_form() == 0
    s0.precond_apply(rhs, out0)
    np.testing.assert_array_equal(out0.download(), h_out)
    s0.close()
    ctx.reset_options()

    # (4) the preconditioned solve (default forms)
    rp, ci, va = ctx.array(h_rp.astype(np.int32)), ctx.array(h_ci), ctx.array(h_va)
    del h_ci, h_va
    s = cm.Solver(ctx, N, N, nnz, rp, ci, va, 0)
    for t in (rp, ci, va):
        t.free()
    xs, b, x, ax = ctx.empty(N), ctx.empty(N), ctx.empty(N), ctx.empty(N)
    ctx.gen_xstar(0, N, 0x5EEE, xs)
    s.spmv(xs, b)
    st = s.solve(b, x, precond=cm.PRECOND_ILU0, loop=cm.LOOP_PBICGSTAB, maxit=100, tol=1e-8, flags=cm.FLAG_X0_ONES)
    assert st.converged and st.iters <= 5
    assert st.trsv_form == 1 and st.trsv_fallbacks == 0 and s.trsv_form() == 1
    np.testing.assert_allclose(x.download(), xs.download(), rtol=1e-6)
    s.spmv(x, ax)
    assert np.linalg.norm(b.download() - ax.download()) <= 1e-7 * st.nrm0
    # the loop in the level-major spaces (default, used above) and the one that permutes around every application of M^-1
    # return the same solution
    x_perm = x.download()
    ctx.set_option("TRSV_PERM", "0")
    st0 = s.solve(b, x, precond=cm.PRECOND_ILU0, loop=cm.LOOP_PBICGSTAB, maxit=100, tol=1e-8, flags=cm.FLAG_X0_ONES)
    ctx.reset_options()
    assert st0.converged and abs(st0.iters - st.iters) <= 1
    np.testing.assert_allclose(x.download(), x_perm, rtol=1e-8)
    s.precond_apply(rhs, out1)
    h_default = out1.download()
    np.testing.assert_array_equal(h_default, h_out)               # level-major storage, same bits as at the top
    s.close()
    ctx.close()
