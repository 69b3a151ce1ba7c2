"""GPU parity on systems that are NOT diagonally dominant: the reference CLI's own default workload first
(example.cpp:173-180,274-288,339: n = 10000, P(0) = 0.99, entries in [1,10], through bicgstab_lu_precond, :352), its usage
line's `-N40 -R0.5` (:190), convection-dominated stencils and random rows with a weak diagonal (tests/nondominant.py).

On most of these systems BiCGSTAB stagnates or breaks down, and the reference's loop has no guard (pbicgstab.cu:81,107,137
divide whatever they get).  What CAN agree between two correct implementations there, and what is asserted:

 1. ||r0|| to 1e-12.
 2. the residual history agrees for as long as the oracle agrees with ITSELF when b is changed by 1, 8 or 64 ulp either way
    (the measure of how fast this system amplifies rounding: 1e-14 -> 1e-6 within 2..90 iterations here; 64 ulp is what
    summation order and fma amount to between the two implementations); the GPU's history may leave the oracle's at most
    3 iterations (6 entries) earlier than that.
 3. the OUTCOME CLASS.  The oracle's trace of the loop's scalars tells when rho = rw.r (:81) or rw.v (:106) has not one
    significant bit left (|rho| <= 4 eps sum|rw_j r_j|: a rho-breakdown in exact arithmetic).  Before that point everything is
    compared as on well-behaved systems (same class: converged / maxit, iteration count +-10 %, no breakdown flag).  From
    that point on the reference's loop divides rounding noise by rounding noise and the two runs are two draws of a random
    process: the oracle's sequential dot products give noise that is never exactly 0 and it wanders on (often to maxit or
    until alpha, omega underflow hundreds of iterations later); the GPU's tree-summed dot products give noise quantised to a
    few ulp of the partial sums, which hits EXACTLY 0 within tens of iterations -> 0/0 -> a NaN residual, which the product
    reports as breakdown = 1 instead of spinning to maxit (deliberate difference).  Asserted there: the GPU
    never stops BEFORE the earliest point at which the oracle's rho loses its last bit over its own perturbed runs (that
    point moves by several iterations with an ulp of b), nor while the two histories still agree; `breakdown` <=> a
    non-finite residual (or the |omega| guard of :735); a run that reports convergence is backed by its true residual as
    well as the oracle's converged runs are.
 4. ILU(0) factors with small pivots (down to 2.5e-4 after cancellation among entries of size 1..10): every entry to
    max(1e-12, 1e-13 g) beyond an absolute 2e-14 g max|a_ij| (entries that a cancellation left), g = the growth factor max|factor entry| / max|a_ij| (the default workload: g = 3.4e5, measured
    2.6e-12; the row updates use fma on the GPU and two roundings in the oracle); beyond g = 1e10 the factors carry no
    digits (example1000_p90 overflows to 4e252): asserted is that the GPU's factorisation blows up as well, and the
    preconditioned loop is then held to the meaning of its flags only (tests/nondominant.py compare_factors).
"""
import numpy as np
import pytest

from tests import nondominant as ND

pytestmark = pytest.mark.gpu

MAXIT, TOL = 2000, 1e-6          # the reference CLI's constants (example.cpp:179-180)


@pytest.fixture(scope="module")
def cm():
    import cuda_mat_amd as cm
    assert cm.device_count() > 0, "no HIP device: the product has no CPU fallback"
    return cm


@pytest.fixture(scope="module")
def ctx(cm):
    c = cm.Context(0)
    yield c
    c.close()


@pytest.fixture(autouse=True)
def _serial_oracle(oracle):
    before = oracle.num_threads()
    oracle.set_num_threads(1)
    yield
    oracle.set_num_threads(before)


def _gpu_run(cm, ctx, A, b, loop, precond):
    s = cm.Solver.from_host_csr(ctx, A.rowptr, A.colidx, A.val)
    db, dx = ctx.array(b), ctx.array(np.ones(A.n))
    lu = None
    try:
        if precond:
            s.ilu0()
            lu = s.ilu0_values()
        st = s.solve(db, dx, precond=precond, loop=loop, maxit=MAXIT, tol=TOL)
        return dx.download(), st, s.history(), lu
    finally:
        for a in (db, dx):
            a.free()
        s.close()


@pytest.mark.parametrize("name", list(ND.FAMILY))
def test_reference_loops_on_nondominant_systems(cm, ctx, oracle, name):
    """rules 1-4 of this file's docstring (tests/nondominant.py compare_loop / compare_factors) on one system, through
    gpu_pbicgstab with M = I, the (A0 + I d) loop with d = 0, and gpu_pbicgstab with ILU(0)"""
    O = oracle
    A, b = ND.FAMILY[name](O)
    vm = O.ilu0(A)
    report, findings, k_nb_plain = [], [], None
    amax = float(np.max(np.abs(A.val)))
    for loop, precond in ((0, 0), (1, 0), (0, 1)):
        xg, st, hg, lu = _gpu_run(cm, ctx, A, b, loop, precond)
        digits_lost = False
        if precond:
            fbad, digits_lost = ND.compare_factors(vm, lu, amax)
            findings += ["%s ILU(0): %s" % (name, m) for m in fbad]
        if digits_lost:
            # factors without a digit left (example1000_p90: |u| up to 4e252): M^-1 is noise times 1e250 on both sides; only
            # the flags are held to their meaning
            report.append("%s loop%d pc%d: the factors carry no digits; GPU it %d conv %d brk %d" % (name, loop, precond, st.iters, st.converged, st.breakdown))
            if st.converged and not st.nrm < TOL * st.nrm0:
                findings.append("%s loop%d pc%d: converged flag without a residual below the target" % (name, loop, precond))
            continue
        line, bad, k_nb = ND.compare_loop(O, A, b, loop, vm if precond else None, (xg, st, hg), MAXIT, TOL, k_nb_plain)
        if loop == 0 and not precond:
            k_nb_plain = k_nb
        report.append("%s %s" % (name, line))
        findings += ["%s loop%d pc%d: %s" % (name, loop, precond, m) for m in bad]
    print("\n".join(report))
    assert not findings, "\n".join(findings)


@pytest.mark.parametrize("dim,p_zero", [(10000, 0.99), (40, 0.5)])
def test_reference_cli_workloads_through_the_drop_in_entry_points(cm, oracle, dim, p_zero):
    """`example` with no arguments (n = 10000, P(0) = 0.99) and `example -N40 -R0.5` (example.cpp:173-180,190): the system
    the reference generates from libc rand(), handed to bicgstab_lu_precond (:352) and to bicgstab() as host arrays.
    The default workload's ILU(0) has pivots of 2.5e-4 and entries of 3e6: M^-1 blows the first direction up, the second
    residual is NaN in the oracle's restatement of the reference (which then spins to maxit = 2000 on NaNs and returns
    `true`, pbicgstab.cu:408) and on the GPU (which stops there: breakdown = 1, not converged, ok = True as upstream)."""
    O = oracle
    A, b = O.example_system(dim, p_zero, 0.2, 1)
    assert A.nnz == {10000: 1007629, 40: 841}[dim]          # the reference's own draws (srand(1) stream of glibc)
    vm = O.ilu0(A)
    xo, so, ho, trace = O.pbicgstab(A, b, vm=vm, maxit=MAXIT, tol=TOL, want_trace=True)
    ok, x, dt, st = cm.bicgstab_lu_precond(A.n, A.nnz, A.val, A.rowptr, A.colidx, b, MAXIT, TOL)
    assert ok is True and not st.converged and not so.converged
    assert abs(st.nrm0 - so.nrm0) <= 1e-12 * so.nrm0
    if dim == 10000:
        first_nan = ND.first_bad(ho[:2 * so.iters])
        assert first_nan == 2 and so.iters == MAXIT          # the oracle: NaN from the second iteration's half step on
        assert st.breakdown and st.iters == 1                # the GPU: stops there
        assert np.all(np.isfinite(x))                        # ... and hands back the last finite iterate
    else:
        assert ND.noise_breakdown(trace) is None and ND.first_bad(ho[:2 * so.iters]) == 2 * MAXIT
        assert not st.breakdown and st.iters == MAXIT == so.iters          # both wander to maxit, all residuals finite
    # the un-preconditioned entry point (pbicgstab.h:113; the reference's own is broken): |omega| guard or NaN
    ok2, x2, dt2, st2 = cm.bicgstab(A.n, A.nnz, A.val, A.rowptr, A.colidx, b, MAXIT, TOL)
    oko, xo2, so2 = O.pbicgstab2(A, b, maxit=MAXIT, tol=TOL)
    assert not ok2 and not oko and st2.breakdown and so2.breakdown
