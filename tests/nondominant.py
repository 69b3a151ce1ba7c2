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


# ---------------------------------------------------------------------------------------------------------------
# The comparison rules of tests/test_gpu_nondominant.py (its docstring states them), as a function that RETURNS its
# findings: the pytest cases assert that there are none, tests/soak.py prints them.
EPS = np.finfo(np.float64).eps
SELF_ITERS = 200                 # iterations of the perturbed oracle runs that measure the system's own amplification
# The yardstick for "how far can two correct implementations be apart": the oracle against itself with b changed by 1, 8 and
# 64 ulp either way.  (64 ulp = 1.4e-14 relative: what the order of summation inside rows of up to ~100 entries, fma
# against two roundings and tree against sequential dot products amount to between the HIP path and the oracle.)
PERTURB = (1, -1, 8, -8, 64, -64)


def noise_breakdown(trace):
    """first iteration of an oracle.pbicgstab(want_trace=True) trace whose rho (pbicgstab.cu:81) or rw.v (:106) is below
    4 eps of the sum of magnitudes it was summed from (not one significant bit left), or not finite, or t.t == 0"""
    with np.errstate(invalid="ignore"):
        bad = ((np.abs(trace[:, 0]) <= 4 * EPS * trace[:, 1]) | ~np.isfinite(trace[:, 0])
               | (np.isfinite(trace[:, 3]) & (np.abs(trace[:, 2]) <= 4 * EPS * trace[:, 3])) | (trace[:, 6] == 0.0))
    w = np.nonzero(bad)[0]
    return int(w[0]) if w.size else None


def oracle_run(O, A, b, loop, vm, maxit, tol):
    """(x, stats, history without the unwritten tail) of the reference loop `loop` (0: pbicgstab.cu:45-154, 1: :581-754)"""
    if loop == 0:
        x, st, h = O.pbicgstab(A, b, vm=vm, maxit=maxit, tol=tol, want_hist=True)
        return x, st, h[:2 * st.iters + st.half_exit]
    ok, x, st, h = O.pbicgstab2(A, b, maxit=maxit, tol=tol, want_hist=True)
    return x, st, h[:st.iters]


def self_spread(O, A, b, loop, vm, h0, tol):
    """the oracle against itself with b changed by PERTURB ulp: (entries over which the histories agree to 1e-6: the
    shortest of the six -- the length itself varies with the perturbation, example40 with ILU(0): 10, 4000, 10, 4 --,
    the cap on the entries compared, the earliest rho-noise point of the perturbed runs of loop 0 or None)"""
    cap = (2 if loop == 0 else 1) * SELF_ITERS
    l_self, k_lo = cap, None
    for k in PERTURB:
        bk = b * (1.0 + k * EPS)
        if loop == 0:
            _, st, h, tr = O.pbicgstab(A, bk, vm=vm, maxit=SELF_ITERS, tol=tol, want_trace=True)
            h = h[:2 * st.iters + st.half_exit]
            kn = noise_breakdown(tr)
            if kn is not None:
                k_lo = kn if k_lo is None else min(k_lo, kn)
        else:
            h = oracle_run(O, A, bk, loop, vm, SELF_ITERS, tol)[2]
        l_self = min(l_self, prefix(h[:cap], h0[:cap], 1e-6))
    return l_self, cap, k_lo


def true_res(O, A, b, x):
    with np.errstate(invalid="ignore", over="ignore"):
        return float(np.linalg.norm(b - O.spmv(A, x)))


def iters_inside_oracle_spread(O, A, b, loop, vm, it_gpu, it_orc, maxit, tol):
    """SURVEY 8c's +-10 % (>= +-2) around the oracle's count -- or around the counts the oracle itself produces when b is
    changed by PERTURB ulp (tests/soak.py's rule for systems that amplify rounding)"""
    if abs(it_gpu - it_orc) <= max(2, 0.1 * it_orc):
        return True, (it_orc, it_orc)
    counts = [it_orc]
    for k in PERTURB:
        _, st, _ = oracle_run(O, A, b * (1.0 + k * EPS), loop, vm, maxit, tol)
        if st.converged:
            counts.append(st.iters)
    lo, hi = min(counts), max(counts)
    # (where the oracle's own count moves by more than 10 % under these perturbations -- soak seed 534 case 19: 53..70 -- the
    # band is widened by that spread: the GPU's 45 there is a draw of the same process, not a finding)
    return lo - max(2, 0.1 * lo, hi - lo) <= it_gpu <= hi + max(2, 0.1 * hi, hi - lo), (lo, hi)


def compare_loop(O, A, b, loop, vm, gpu, maxit, tol, k_nb_plain=None):
    """gpu = (x, stats, history) of the HIP path for the same system / loop / preconditioner (vm: the oracle's ILU(0)
    values or None).  Returns (report line, findings, k_nb): findings is a list of strings, empty when every rule holds;
    k_nb = (the oracle's rho-noise point, the earliest one over its perturbed runs)."""
    xg, st, hg = gpu
    bad = []
    if loop == 0:
        xo, so, ho, trace = O.pbicgstab(A, b, vm=vm, maxit=maxit, tol=tol, want_trace=True)
        ho = ho[:2 * so.iters + so.half_exit]
        k_nb = noise_breakdown(trace)
    else:
        # (the (A0 + I d) loop runs the recurrences of the M = I loop in exact arithmetic: its noise points are that loop's)
        xo, so, ho = oracle_run(O, A, b, loop, vm, maxit, tol)
        k_nb = k_nb_plain[0] if k_nb_plain else None
    l_self, cap, k_lo = self_spread(O, A, b, loop, vm, ho, tol)
    if loop == 1:
        k_lo = k_nb_plain[1] if k_nb_plain else None
    if k_nb is not None:
        k_lo = k_nb if k_lo is None else min(k_lo, k_nb)
    noisy = k_nb is not None or k_lo is not None          # the loop runs on rounding noise from some iteration on, in this run or a perturbed one
    line = "loop%d pc%d: oracle it %d conv %d brk %d, GPU it %d conv %d brk %d, oracle's rho loses its last bit at %s (perturbed runs: from %s)" % (
        loop, int(vm is not None), so.iters, so.converged, so.breakdown, st.iters, st.converged, st.breakdown, k_nb, k_lo)
    # 1. the initial residual
    if not abs(st.nrm0 - so.nrm0) <= 1e-12 * so.nrm0:
        bad.append("nrm0 %r vs %r" % (st.nrm0, so.nrm0))
    # 2. the history agrees for as long as the oracle agrees with itself under PERTURB ulp changes of b
    l_gpu = prefix(hg[:cap], ho[:cap], 1e-6)
    if l_gpu < min(l_self, len(ho), len(hg)) - (6 if loop == 0 else 3):      # three iterations (soak seed 532 case 31: 10 against 16 entries)
        bad.append("history leaves the oracle's after %d entries, the oracle's own (b changed by up to 64 ulp) after %d" % (l_gpu, l_self))
    line += "; history equal to 1e-6 over %d entries (oracle vs itself, b changed by up to 64 ulp: %d)" % (l_gpu, l_self)
    # the flags mean what they say
    hist_bad = first_bad(hg) < len(hg) or not np.isfinite(st.nrm)
    if st.breakdown:
        if st.converged: bad.append("breakdown AND converged")
        if not (hist_bad or loop == 1): bad.append("breakdown flag without a non-finite residual")     # loop 1 also breaks on |omega| < 1e-5 (:735)
    elif hist_bad:
        bad.append("a non-finite residual went unreported")
    if st.converged:
        if not (st.nrm < tol * st.nrm0 and st.iters <= maxit): bad.append("converged flag without a residual below the target")
    elif not st.breakdown and st.iters != maxit:
        bad.append("stopped at %d of %d iterations without converged / breakdown" % (st.iters, maxit))
    # 3. the outcome class
    tr_g, tr_o = true_res(O, A, b, xg), true_res(O, A, b, xo)
    if not noisy:
        # no breakdown in the oracle's runs: the classes must be the same, as on well-behaved systems
        if st.breakdown: bad.append("breakdown where the oracle's rho keeps its bits")
        if bool(st.converged) != bool(so.converged): bad.append("converged %d vs %d" % (st.converged, so.converged))
        elif so.converged:
            ok, (lo, hi) = iters_inside_oracle_spread(O, A, b, loop, vm, st.iters, so.iters, maxit, tol)
            if not ok: bad.append("iterations %d vs %d, outside the oracle's own spread %d..%d" % (st.iters, so.iters, lo, hi))
            # (the recursive residual may have left the true one on BOTH sides: convdiff_g8 with ILU(0) "converges" at a
            # true residual of 1e5 in the oracle and on the GPU alike -- compared like for like)
            if not tr_g <= 10.0 * max(tr_o, tol * so.nrm0): bad.append("true residual %g vs the oracle's %g" % (tr_g, tr_o))
        elif not (st.iters == so.iters == maxit):
            bad.append("iterations %d vs %d at maxit %d" % (st.iters, so.iters, maxit))
    else:
        # from the noise point on the loop runs on rounding noise: any class may follow -- but not before the earliest noise
        # point of the oracle's own runs, and not while the two histories still agree
        per = 2 if loop == 0 else 1
        if not st.converged and (hist_bad or not st.breakdown):
            if st.iters < min(k_lo, so.iters) - 2: bad.append("stopped at %d, before the oracle's rho lost its last bit (%d; perturbed runs: from %d)" % (st.iters, k_nb if k_nb is not None else -1, k_lo))
            if st.breakdown and st.iters + 1 < l_gpu // per: bad.append("breakdown at %d while the histories still agree (%d entries)" % (st.iters, l_gpu))
        elif not st.converged:
            # loop 1 stopped by the reference's own |omega| < 1e-5 guard (pbicgstab.cu:735) on finite residuals: an event of
            # the trajectory itself, which may fire anywhere once the two histories have parted (example1000_p90: the
            # histories part after 5 iterations, the guard fires at 34 on the GPU and at 154 in the oracle) -- not before
            if not (loop == 1 and st.iters >= l_gpu): bad.append("|omega| guard at %d while the histories still agree (%d)" % (st.iters, l_gpu))
        elif so.converged:
            if not tr_g <= 10.0 * max(tr_o, tol * so.nrm0): bad.append("true residual %g vs the oracle's %g" % (tr_g, tr_o))
        elif not tr_g <= 100.0 * tol * so.nrm0:       # the GPU's draw converged, the oracle's did not: a solution in its own right?
            bad.append("converged on noise with a true residual of %g" % tr_g)
    return line, bad, (k_nb, k_lo)


def compare_factors(vm, lu, amax):
    """ILU(0) values of the HIP path against the oracle's; amax = max |entry of A|.  Returns (findings, digits_lost).
    The componentwise error of an elimination grows with its growth factor g = max|factor entry| / max|a_ij|: tolerance
    max(1e-12, 1e-13 g) on every entry (beyond an absolute 2e-14 g max|a_ij|) while g < 1e10 (the reference CLI's default workload: g = 3.4e5, measured 2.6e-12;
    soak: 5.7e-12 at g = 130, 9.4e-12 at g = 460 -- entries that are themselves the result of a cancellation);
    beyond that the factors carry no digits (a cancellation at 1e10+ has eaten them; soak: g = 1e92 ... overflow) --
    `digits_lost`: which entries end as inf, NaN, 1e100 or 1e300 is decided by the rounding of the row updates; asserted is
    only that the GPU's factorisation lost its digits as well (growth >= 1e8 or non-finite entries), and the preconditioned
    loop is then not compared beyond its flags."""
    fin = np.isfinite(vm)
    with np.errstate(invalid="ignore", over="ignore", divide="ignore"):
        big = float(np.max(np.abs(vm[fin]))) if fin.any() else np.inf
        g = big / amax if fin.all() else np.inf
        if g >= 1e10:
            fl = np.isfinite(lu)
            blew = (not fl.all()) or (fl.any() and float(np.max(np.abs(lu[fl]))) >= 1e8 * amax)
            return ([] if blew else ["the oracle's factors grow by %.1e, the GPU's do not" % g]), True
        if not np.isfinite(lu).all():
            return ["non-finite factor entries on the GPU only (oracle growth %.1e)" % g], False
        tolf = max(1e-12, 1e-13 * g)
        # an entry that is itself what a cancellation left (|entry| << the terms of its row update) carries the ABSOLUTE error
        # of those terms: 2e-14 of the largest magnitude in play (soak seed 601 case 75: g = 1.2, every entry but a handful
        # to 1e-13, one small entry 1.96e-12 off in relative terms)
        excess = np.abs(lu - vm) - 2e-14 * amax * max(1.0, g)
        rel = np.maximum(excess, 0.0) / np.maximum(np.abs(vm), 1e-300)
    return ([] if rel.max() <= tolf else ["factors differ by %.2e beyond the absolute term (growth %.1e: tolerance %.1e)" % (rel.max(), g, tolf)]), False
