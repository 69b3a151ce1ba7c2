"""Systems that are NOT diagonally dominant (test infrastructure: shared by tests/test_gpu_nondominant.py, tests/soak.py and
tests/nondominant_probe.py).  Three families:
  example_*   the reference CLI's own random system (example.cpp:274-288,339 through oracle.example_system: libc rand(),
              entries in [1,10] everywhere, diagonal in [1,10]): what `example` solves by default (n = 10000, P(0) = 0.99,
              example.cpp:173-180) and in its usage line `-N40 -R0.5` (:190)
  convdiff_*  5-point convection-diffusion with central differences at cell Peclet numbers above 1: off-diagonals of both
              signs larger than the diagonal's share, strongly non-normal
  weakdiag_*  random sparse rows whose diagonal is theta * (sum of |off-diagonals|), theta <= 1
Each entry: (name, A as oracle.Csr, b)."""
import numpy as np
import scipy.sparse as sp


def _csr(O, S, base):
    S = S.tocsr()
    S.sort_indices()
    return O.Csr(S.shape[0], (S.indptr + base).astype(np.int32), (S.indices + base).astype(np.int32), S.data.astype(np.float64).copy(),
                 S.shape[0])


def convdiff(O, nx, ny, gamma, delta, base=0):
    """-Laplace(u) + c . grad(u), central differences, row-major grid: centre 4, west -1-gamma, east -1+gamma,
    south -1-delta, north -1+delta (gamma, delta = cell Peclet numbers / 2 in x and y)"""
    n = nx * ny
    i = np.arange(n)
    x, y = i % nx, i // nx
    rows, cols, vals = [i], [i], [np.full(n, 4.0)]
    for mask, off, v in ((x > 0, -1, -1.0 - gamma), (x < nx - 1, 1, -1.0 + gamma), (y > 0, -nx, -1.0 - delta), (y < ny - 1, nx, -1.0 + delta)):
        rows.append(i[mask]); cols.append(i[mask] + off); vals.append(np.full(mask.sum(), v))
    S = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n))
    S.eliminate_zeros()
    return _csr(O, S, base)


def weakdiag(O, n, per, theta, seed, base=0, signs=False):
    """`per` off-diagonal entries per row on average, uniform in (-1, 1); diagonal = theta * sum |off-diagonals| + 0.01
    (every sign + unless `signs`: then a random sign per row)"""
    rng = np.random.default_rng(seed)
    nnz = int(n * per)
    S = sp.csr_matrix((rng.uniform(-1, 1, nnz), (rng.integers(0, n, nnz), rng.integers(0, n, nnz))), shape=(n, n))
    S.sum_duplicates(); S.setdiag(0); S.eliminate_zeros()
    d = theta * np.asarray(abs(S).sum(axis=1)).ravel() + 0.01
    if signs:
        d *= rng.choice(np.array([-1.0, 1.0]), n)
    return _csr(O, S + sp.diags(d), base)


def rhs_for(O, A, seed=0):
    """b = A x* with x* in [1, 2): the initial residual with x0 = 1 is then neither 0 nor huge"""
    rng = np.random.default_rng(1000 + seed)
    return O.spmv(A, 1.0 + rng.random(A.n))


FAMILY = {
    # the reference's default workload and its usage-line example
    "example10000": lambda O: O.example_system(10000, 0.99, 0.2, 1),
    "example40": lambda O: O.example_system(40, 0.5, 0.2, 1),
    "example2000": lambda O: O.example_system(2000, 0.99, 0.2, 1),
    "example1000_p90": lambda O: O.example_system(1000, 0.9, 0.2, 1),
    "example3000_seed7": lambda O: O.example_system(3000, 0.995, 0.2, 7),
    "example300_p98_seed3": lambda O: O.example_system(300, 0.98, 0.2, 3),
    # convection-dominated stencils
    "convdiff_g0.5": lambda O: (lambda A: (A, rhs_for(O, A, 1)))(convdiff(O, 120, 100, 0.5, 0.25)),
    "convdiff_g2": lambda O: (lambda A: (A, rhs_for(O, A, 2)))(convdiff(O, 120, 100, 2.0, 1.0, base=1)),
    "convdiff_g8": lambda O: (lambda A: (A, rhs_for(O, A, 3)))(convdiff(O, 150, 90, 8.0, 3.0)),
    "convdiff_g40": lambda O: (lambda A: (A, rhs_for(O, A, 4)))(convdiff(O, 90, 90, 40.0, 0.0)),
    # random rows with a weak diagonal
    "weakdiag_t1.0": lambda O: (lambda A: (A, rhs_for(O, A, 5)))(weakdiag(O, 6000, 6, 1.0, 11)),
    "weakdiag_t0.6": lambda O: (lambda A: (A, rhs_for(O, A, 6)))(weakdiag(O, 6000, 6, 0.6, 12, base=1)),
    "weakdiag_t0.3": lambda O: (lambda A: (A, rhs_for(O, A, 7)))(weakdiag(O, 4000, 4, 0.3, 13)),
    "weakdiag_t0.8_signs": lambda O: (lambda A: (A, rhs_for(O, A, 8)))(weakdiag(O, 5000, 5, 0.8, 14, signs=True)),
    "weakdiag_t0.5_long": lambda O: (lambda A: (A, rhs_for(O, A, 9)))(weakdiag(O, 2500, 40, 0.5, 15)),
}


def first_bad(h):
    """index of the first history entry that is not finite (len(h) if none)"""
    w = np.nonzero(~np.isfinite(h))[0]
    return int(w[0]) if w.size else len(h)


def prefix(hg, ho, rtol):
    """number of leading history entries over which GPU and oracle agree to rtol (both finite)"""
    m = min(len(hg), len(ho))
    if m == 0:
        return 0
    with np.errstate(invalid="ignore", divide="ignore"):
        ok = np.isfinite(hg[:m]) & np.isfinite(ho[:m]) & (np.abs(hg[:m] - ho[:m]) <= rtol * np.abs(ho[:m]))
    w = np.nonzero(~ok)[0]
    return int(w[0]) if w.size else m
