"""GPU box, by hand: every system of tests/nondominant.py through the reference loops on the GPU and through the oracle;
prints where histories part, how each ends, the ILU(0) factors' agreement.   python tests/nondominant_probe.py [names...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cuda_mat_amd as cm
from oracle import oracle as O
from tests import nondominant as ND
O.set_num_threads(1)
ctx = cm.Context(0)
names = sys.argv[1:] or list(ND.FAMILY)
MAXIT, TOL = 2000, 1e-6            # the reference CLI's constants (example.cpp:179-180)
for name in names:
    A, b = ND.FAMILY[name](O)
    print("== %s n=%d nnz=%d base=%d" % (name, A.n, A.nnz, A.base), flush=True)
    try:
        vm = O.ilu0(A)
    except ZeroDivisionError as e:
        vm = None; print("   oracle ilu0:", e)
    for loop, precond in ((0, 0), (1, 0), (0, 1)):
        if precond and vm is None: continue
        t0 = time.time()
        if loop == 0: xo, so, ho = O.pbicgstab(A, b, vm=vm if precond else None, maxit=MAXIT, tol=TOL, want_hist=True)
        else: ok, xo, so, ho = O.pbicgstab2(A, b, maxit=MAXIT, tol=TOL, want_hist=True)
        to = time.time() - t0
        nho = 2 * so.iters + so.half_exit if loop == 0 else so.iters
        ho = ho[:nho]
        s = cm.Solver.from_host_csr(ctx, A.rowptr, A.colidx, A.val)
        db, dx = ctx.array(b), ctx.array(np.ones(A.n))
        t0 = time.time()
        try:
            if precond:
                s.ilu0()
                lu = s.ilu0_values()
                with np.errstate(invalid="ignore", divide="ignore", over="ignore"):
                    fin = np.isfinite(vm) & np.isfinite(lu)
                    rel = np.abs(lu[fin] - vm[fin]) / np.maximum(np.abs(vm[fin]), 1e-300)
                print("   ilu0: finite both %d / %d, same non-finite set %s, max rel diff %.2e, 99.9%% %.2e; max|vm| %.2e" % (
                    fin.sum(), len(vm), np.array_equal(np.isfinite(vm), np.isfinite(lu)), rel.max() if rel.size else 0, np.quantile(rel, 0.999) if rel.size else 0, np.nanmax(np.abs(vm))))
            st = s.solve(db, dx, precond=precond, loop=loop, maxit=MAXIT, tol=TOL)
        except cm.CudamatError as e:
            print("   loop%d pc%d GPU error: %s" % (loop, precond, e)); s.close(); continue
        tg = time.time() - t0
        xg = dx.download(); hg = s.history()
        with np.errstate(invalid="ignore", over="ignore"):
            tr_g = np.linalg.norm(b - O.spmv(A, xg)); tr_o = np.linalg.norm(b - O.spmv(A, xo))
        print("   loop%d pc%d  oracle: it %d half %d conv %d brk %d nrm %.3e (true %.3e) first-nonfinite %d/%d  [%.1fs]" % (
            loop, precond, so.iters, so.half_exit, so.converged, so.breakdown, so.nrm, tr_o, ND.first_bad(ho), len(ho), to))
        print("             GPU   : it %d half %d conv %d brk %d nrm %.3e (true %.3e) first-nonfinite %d/%d  form %d fallbacks %d/%d [%.1fs]" % (
            st.iters, st.half_exit, st.converged, st.breakdown, st.nrm, tr_g, ND.first_bad(hg), len(hg), st.loop_form, st.loop_fallbacks, st.trsv_fallbacks, tg))
        print("             nrm0 rel diff %.1e; history prefix equal to 1e-12: %d, 1e-9: %d, 1e-6: %d, 1e-3: %d, 1e-1: %d; x finite gpu %s oracle %s" % (
            abs(st.nrm0 - so.nrm0) / so.nrm0, ND.prefix(hg, ho, 1e-12), ND.prefix(hg, ho, 1e-9), ND.prefix(hg, ho, 1e-6), ND.prefix(hg, ho, 1e-3), ND.prefix(hg, ho, 1e-1),
            bool(np.all(np.isfinite(xg))), bool(np.all(np.isfinite(xo)))), flush=True)
        for a in (db, dx): a.free()
        s.close()
