"""soak test (run by hand: python tests/soak.py <seed> <cases>; not collected by pytest): many random systems through
every path, against the oracle -- lives under tests/ because only test code may use the oracle"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
import cuda_mat_amd as cm
from oracle import oracle as O
O.set_num_threads(1)
ctx = cm.Context(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad = 0
forms = {}
dicts = 0
stencils = {}
nd_total = 0
nd_classes = {}     # (converged, breakdown flag, the oracle's rho lost its bits) -> solves


def oracle_spread(A, b, loop, precond, first, nperm=8):
    """iteration counts of the oracle's loop on P A P^T, P b for 8 random permutations P: the same system in exact
    arithmetic, other summation orders in the dots and row sums.  (With ILU(0), whose factors depend on the ordering, b
    is scaled by 1 + k ulp instead: the same solution path up to rounding-sized input changes.)"""
    base = int(A.rowptr[0])
    S = sp.csr_matrix((A.val, A.colidx - base, A.rowptr - base), shape=(A.n, A.n))
    prng = np.random.default_rng(12345)
    counts = [first]
    for k in range(nperm):
        if precond:
            bk = b * (1.0 + (k - nperm // 2 + (k >= nperm // 2)) * 1.1e-16)
            if loop == 2: _, sp_ = O.pipelined_bicgstab(A, bk, vm=O.ilu0(A), maxit=500, tol=1e-9)
            else: _, sp_ = O.pbicgstab(A, bk, vm=O.ilu0(A), maxit=500, tol=1e-9)
            if sp_.converged: counts.append(sp_.iters)
            continue
        perm = prng.permutation(A.n)
        P = sp.csr_matrix((np.ones(A.n), (np.arange(A.n), perm)), shape=(A.n, A.n))
        Sp = (P @ S @ P.T).tocsr(); Sp.sort_indices()
        Ap = O.Csr(A.n, (Sp.indptr + base).astype(np.int32), (Sp.indices + base).astype(np.int32), Sp.data.copy(), A.n)
        bp = np.asarray(P @ b).ravel()
        if loop == 0: _, sp_ = O.pbicgstab(Ap, bp, vm=O.ilu0(Ap) if precond else None, maxit=500, tol=1e-9)
        elif loop == 2: _, sp_ = O.pipelined_bicgstab(Ap, bp, maxit=500, tol=1e-9)      # (precond: handled above by scaling b)
        else: _, _, sp_ = O.pbicgstab2(Ap, bp, maxit=500, tol=1e-9)
        if sp_.converged: counts.append(sp_.iters)
    return min(counts), max(counts)


for case in range(ncase):
    n = int(rng.integers(1, 40000))
    per = float(rng.choice([1.5, 4, 9, 30, 80]))
    nnz_t = int(min(n * per, 3e6))
    if case % 3 == 2:      # skewed rows: Pareto-distributed lengths, a few hub rows (exercises the tile SpMV)
        lens = np.minimum(1 + (rng.pareto(1.2, n) * per / 4).astype(np.int64), max(n // 2, 1))
        lens[rng.integers(0, n, 3)] = max(n // 3, 1)
        ri = np.repeat(np.arange(n), lens)[:int(3e6)]
        nnz_t = ri.size
    else:
        ri = rng.integers(0, n, nnz_t)
    cj = rng.integers(0, n, nnz_t)
    vals = rng.uniform(-1, 1, nnz_t)
    if case % 4 == 1:      # a few distinct values (duplicates are summed below, the diagonal is added: still only dozens
        vals = rng.choice(np.array([-1.0, -0.5, 0.25, 2.0]), nnz_t)      # ... of bit patterns): the value-dictionary forms
    S = sp.csr_matrix((vals, (ri, cj)), shape=(n, n)); S.sum_duplicates()
    S.setdiag(0); S.eliminate_zeros()
    rowsum = np.asarray(abs(S).sum(axis=1)).ravel()
    # few-valued cases: a diagonal from a small set as well (multiples of 8 above the row sum), so that the matrix keeps
    # <= 256 distinct values and the value-dictionary kernels run.  (test_pipelined_loop_verifies_its_iterate replays this
    # generator's DRAWS for seed 21 case 41 with its own diagonal rule: keep the order of the rng calls.)
    diag = 8.0 * np.ceil((rowsum + 1.5) / 8.0) if case % 4 == 1 else 1.0 + rng.random(n) + rowsum
    S = (S + sp.diags(diag)).tocsr(); S.sort_indices()
    base = int(rng.integers(0, 2))
    A = O.Csr(n, (S.indptr + base).astype(np.int32), (S.indices + base).astype(np.int32), S.data.copy(), n)
    x = rng.standard_normal(n); want = O.spmv(A, x)
    msgs = []
    for mode in ("csr", "tiles", "pb", "sell", None):
        ctx.reset_options()                    # (a context reads the environment when it is created; switches go through set_option)
        if mode == "tiles": ctx.set_option("SPMV_MODE", "csr"); ctx.set_option("SPMV_FORM", "tiles")
        elif mode: ctx.set_option("SPMV_MODE", mode)
        s = cm.Solver.from_host_csr(ctx, A.rowptr, A.colidx, A.val)
        dx, dy = ctx.array(x), ctx.empty(n)
        s.spmv(dx, dy); y = dy.download()
        dicts += 1 if s.value_dict() > 0 else 0
        if mode in ("pb", "sell"):
            if not np.array_equal(y, want): msgs.append("%s spmv not bit-exact" % mode)
        else:
            absA = O.Csr(n, A.rowptr, A.colidx, np.abs(A.val), n)
            bound = 4 * (np.diff(A.rowptr).max() + 1) * 2.3e-16 * O.spmv(absA, np.abs(x)) + 1e-300
            if not np.all(np.abs(y - want) <= bound): msgs.append("%s spmv out of tolerance" % mode)
        for a in (dx, dy): a.free()
        s.close()
    ctx.reset_options()
    xs = 1.0 + rng.random(n); b = O.spmv(A, xs)
    for loop in (0, 1, 2):                     # 2 = pipelined BiCGStab: checked against its own restatement
        for precond in ((0, 1) if loop in (0, 2) else (0,)):
            s = cm.Solver.from_host_csr(ctx, A.rowptr, A.colidx, A.val)
            db, dxx = ctx.array(b), ctx.array(np.ones(n))
            try:
                st = s.solve(db, dxx, precond=precond, loop=loop, maxit=500, tol=1e-9)
            except cm.CudamatError as e:
                msgs.append("solve error %s" % e); s.close(); continue
            xg = dxx.download()
            forms[st.loop_form] = forms.get(st.loop_form, 0) + 1
            if loop == 0: xo, so = O.pbicgstab(A, b, vm=O.ilu0(A) if precond else None, maxit=500, tol=1e-9)
            elif loop == 2: xo, so = O.pipelined_bicgstab(A, b, vm=O.ilu0(A) if precond else None, maxit=500, tol=1e-9)
            else: ok, xo, so = O.pbicgstab2(A, b, maxit=500, tol=1e-9)
            # hundreds of un-preconditioned iterations on hub matrices are chaotic in the rounding order: there only
            # "the GPU must not do worse than the oracle by more than 2x" is checked
            long_run = so.iters > 100 or st.iters > 100
            if bool(so.converged) and not bool(st.converged) and so.iters < 400: msgs.append("loop%d pc%d GPU did not converge, oracle did in %d" % (loop, precond, so.iters))
            elif bool(st.converged) != bool(so.converged) and not long_run: msgs.append("loop%d pc%d converged %d vs %d" % (loop, precond, st.converged, so.converged))
            if st.converged and so.converged:
                # iteration count within +-10 % (>= +-2) of the CPU restatement.  BiCGSTAB amplifies rounding
                # differences on some systems (seed 202 case 77: histories equal to 1e-11 up to iteration 5, 1e-3 apart at
                # iteration 10, unrelated from 25 on; the oracle's OWN count moves between 31 and 41 over 24 symmetric
                # permutations of the system, 33...39 when b is scaled by one ulp; the GPU needs 33) -- so a count outside
                # the band is a finding only if it is also outside the band around the oracle's own spread over other
                # summation orders (8 symmetric permutations of the same system)
                lim = so.iters if long_run else max(2, 0.1 * so.iters)
                if abs(st.iters - so.iters) > lim:
                    lo, hi = oracle_spread(A, b, loop, precond, so.iters)
                    # The PIPELINED loop (loop 2) is not a reference loop and its contract is one-sided: its "converged" is
                    # verified against the TRUE residual (<= 2 x target, checked below too), so an iterate reached in FEWER
                    # iterations than the oracle needs is never a defect -- where the last drop off a plateau falls is decided
                    # by rounding (seed 51 case 29: 69 on the GPU, 76..91 for the oracle over 48 summation orders, histories
                    # equal to 1e-12 up to iteration 3 and unrelated from 11 on).  A finding is a count ABOVE the oracle's own
                    # spread + 10 %.  The reference loops (0, 1) keep SURVEY 8c's two-sided band.
                    inside = lambda lo, hi: (loop == 2 or lo - max(2, 0.1 * lo) <= st.iters) and st.iters <= hi + max(2, 0.1 * hi)
                    if not inside(lo, hi):
                        # 8 orders are a small sample: look at 24 before calling it a finding.  (Seed 51 case 29, pipelined
                        # loop, stays one: 79..89 over these 24 permutations -- 76..91 over another 24 -- and 69 on the GPU;
                        # histories equal to 1e-12 up to iteration 3, unrelated from 11 on: tests/soak_case.py 51 29 2)
                        lo, hi = oracle_spread(A, b, loop, precond, so.iters, nperm=24)
                    if not inside(lo, hi):
                        msgs.append("loop%d pc%d iters %d vs %d (oracle over other summation orders: %d..%d)" % (loop, precond, st.iters, so.iters, lo, hi))
            if st.converged:
                if so.converged and np.linalg.norm(xg - xo) > 1e-5 * np.linalg.norm(xo): msgs.append("loop%d pc%d x differs" % (loop, precond))
                if np.linalg.norm(b - O.spmv(A, xg)) > 1e-7 * so.nrm0 + 1e-300: msgs.append("loop%d pc%d residual" % (loop, precond))
            for a in (db, dxx): a.free()
            s.close()
    # the hybrid triangular solves in level-major index spaces, forced on this (small) system: the preconditioned loop
    # run in those spaces and the one that permutes around every application must both return the reference loop's solution
    if np.diff(A.rowptr).max() <= 1024 and n >= 64:
        xref = None
        for hyb, perm in (("0", "1"), ("1", "1"), ("1", "0")):
            ctx.set_option("TRSV_HYBRID", hyb); ctx.set_option("TRSV_PERM", perm)
            s = cm.Solver.from_host_csr(ctx, A.rowptr, A.colidx, A.val)
            db, dxx = ctx.array(b), ctx.array(np.ones(n))
            try:
                st = s.solve(db, dxx, precond=1, loop=0, maxit=500, tol=1e-9)
                xg = dxx.download()
                if xref is None: xref, itref = xg, st.iters
                elif st.converged and np.linalg.norm(xg - xref) > 1e-6 * np.linalg.norm(xref): msgs.append("hybrid=%s perm=%s x differs from the plain factors" % (hyb, perm))
                elif abs(st.iters - itref) > max(2, 0.1 * itref) and max(st.iters, itref) <= 100: msgs.append("hybrid=%s perm=%s iters %d vs %d" % (hyb, perm, st.iters, itref))
            except cm.CudamatError as e:
                msgs.append("hybrid=%s perm=%s solve error %s" % (hyb, perm, e))
            for a in (db, dxx): a.free()
            s.close()
        ctx.reset_options()
    outs = []
    rhs = rng.standard_normal(n)
    for form in ("1", "0"):
        ctx.set_option("TRSV_SYNCFREE", form)
        s = cm.Solver.from_host_csr(ctx, A.rowptr, A.colidx, A.val)
        try:
            s.ilu0()
            dr, do = ctx.array(rhs), ctx.empty(n)
            s.precond_apply(dr, do); outs.append(do.download())
            for a in (dr, do): a.free()
        except cm.CudamatError as e:
            msgs.append("ilu/trsv error %s" % e)
        s.close()
    ctx.reset_options()
    if len(outs) == 2:
        if not np.array_equal(outs[0], outs[1]): msgs.append("dependency-driven trsv differs from level trsv")
        lu = O.ilu0(A); ref = O.trsv_upper(A, lu, O.trsv_lower_unit(A, lu, rhs))
        if np.linalg.norm(outs[0] - ref) > 1e-9 * np.linalg.norm(ref) + 1e-300: msgs.append("trsv vs oracle")
    # every fifth case also brings a STENCIL-like system of its own (own generator: the draws above stay what they were): a few
    # diagonals at random offsets, clipped at the boundary, real or few-valued -- the row-pattern dictionary forms
    # (csrc/spmv_pat.hip: k_spmv_pat / k_spmv_pat_d) forced, against the oracle bit for bit, and one solve through them
    if case % 5 == 4:
        r2 = np.random.default_rng(1000 * (int(sys.argv[1]) if len(sys.argv) > 1 else 0) + case)
        n2 = int(r2.integers(70000, 400000))
        offs = np.unique(np.concatenate([[0], r2.integers(-n2 // 3, n2 // 3, int(r2.integers(2, 9)))]))
        few = bool(r2.integers(0, 2))
        diags = []
        for o in offs:
            m = n2 - abs(int(o))
            diags.append(r2.choice(np.array([-1.0, -0.25, 0.5]), m) if few else r2.uniform(-1, 1, m))
        S2 = sp.diags(diags, [int(o) for o in offs], shape=(n2, n2), format="lil")
        for r in r2.integers(0, n2, 4): S2[int(r), :] = 0                  # a few empty rows
        S2 = S2.tocsr(); S2.setdiag(0); S2.eliminate_zeros()
        rs2 = np.asarray(abs(S2).sum(axis=1)).ravel()
        S2 = (S2 + sp.diags(8.0 * np.ceil((rs2 + 1.5) / 8.0) if few else 1.0 + rs2 + r2.random(n2))).tocsr(); S2.sort_indices()
        b2 = int(r2.integers(0, 2))
        A2 = O.Csr(n2, (S2.indptr + b2).astype(np.int32), (S2.indices + b2).astype(np.int32), S2.data.copy(), n2)
        x2 = r2.standard_normal(n2); want2 = O.spmv(A2, x2)
        for mode in ("pat", None):
            ctx.reset_options()
            if mode: ctx.set_option("SPMV_MODE", mode)
            try:
                s2 = cm.Solver.from_host_csr(ctx, A2.rowptr, A2.colidx, A2.val)
                dx, dy = ctx.array(x2), ctx.empty(n2)
                s2.spmv(dx, dy)
                name = s2.spmv_kernel()
                if mode == "pat" and not name.startswith("k_spmv_pat"): msgs.append("stencil: forced pat runs %s" % name)
                if mode == "pat": stencils[name] = stencils.get(name, 0) + 1
                if name.startswith("k_spmv_pat") and not np.array_equal(dy.download(), want2): msgs.append("stencil: %s not bit-exact" % name)
                elif not np.allclose(dy.download(), want2, rtol=1e-12, atol=1e-12): msgs.append("stencil: %s differs" % name)
                if mode == "pat":
                    xs2 = 1.0 + r2.random(n2); bb = O.spmv(A2, xs2)
                    db, dxx = ctx.array(bb), ctx.array(np.ones(n2))
                    st2 = s2.solve(db, dxx, loop=0, maxit=300, tol=1e-9)
                    xo2, so2 = O.pbicgstab(A2, bb, maxit=300, tol=1e-9)
                    if bool(st2.converged) != bool(so2.converged) or (st2.converged and abs(st2.iters - so2.iters) > max(2, 0.1 * so2.iters)):
                        msgs.append("stencil: solve through %s: %d iterations (converged %d) vs %d (%d)" % (name, st2.iters, st2.converged, so2.iters, so2.converged))
                    elif st2.converged and np.linalg.norm(dxx.download() - xo2) > 1e-5 * np.linalg.norm(xo2): msgs.append("stencil: x differs")
                    for a in (db, dxx): a.free()
                for a in (dx, dy): a.free()
                s2.close()
            except cm.CudamatError as e:
                msgs.append("stencil (%s): %s" % (mode, e))
        ctx.reset_options()
    # every fourth case also brings a system that is NOT diagonally dominant (own generator: the draws above stay what they
    # were): the same kind of random rows with diagonal = theta * sum |off-diagonals|, theta in {1, 0.6, 0.3}, a random sign
    # per row in half of them -- or, every eighth case, the reference CLI's own recipe (example.cpp:274-288: entries in
    # [1,10] from libc rand()).  Compared by the rules of tests/test_gpu_nondominant.py (tests/nondominant.py compare_loop):
    # initial residual, history prefix against the oracle's own agreement under 1..3 ulp changes of b, outcome class with the
    # oracle's rho-noise point, ILU(0) factors with small pivots
    if case % 4 == 3:
        from tests import nondominant as ND
        r3 = np.random.default_rng(7000 * (int(sys.argv[1]) if len(sys.argv) > 1 else 0) + case)
        if case % 8 == 7:
            A3, b3 = O.example_system(int(r3.integers(30, 3000)), float(r3.choice([0.5, 0.9, 0.98, 0.995])), 0.2, int(r3.integers(1, 1 << 20)))
        else:
            A3 = ND.weakdiag(O, int(r3.integers(200, 20000)), float(r3.choice([3, 6, 12, 40])), float(r3.choice([1.0, 0.6, 0.3])),
                             int(r3.integers(0, 1 << 30)), base=int(r3.integers(0, 2)), signs=bool(r3.integers(0, 2)))
            b3 = ND.rhs_for(O, A3, int(r3.integers(0, 1000)))
        nd_total += 1
        try:
            vm3 = O.ilu0(A3)
        except ZeroDivisionError:
            vm3 = None
        k_plain = None
        for loop3, pc3 in ((0, 0), (1, 0), (0, 1)):
            if pc3 and vm3 is None: continue
            s3 = cm.Solver.from_host_csr(ctx, A3.rowptr, A3.colidx, A3.val)
            db, dxx = ctx.array(b3), ctx.array(np.ones(A3.n))
            try:
                lu3 = None
                if pc3:
                    s3.ilu0(); lu3 = s3.ilu0_values()
                st3 = s3.solve(db, dxx, precond=pc3, loop=loop3, maxit=600, tol=1e-6)
                lost = False
                if pc3:
                    fbad, lost = ND.compare_factors(vm3, lu3, float(np.max(np.abs(A3.val))))
                    msgs += ["non-dominant n=%d nnz=%d ILU(0): %s" % (A3.n, A3.nnz, m) for m in fbad]
                if lost:               # factors without a digit left: the preconditioned loop is noise on both sides
                    nd_classes[("factors without digits",)] = nd_classes.get(("factors without digits",), 0) + 1
                    for a in (db, dxx): a.free()
                    s3.close()
                    continue
                line, found, k_nb = ND.compare_loop(O, A3, b3, loop3, vm3 if pc3 else None, (dxx.download(), st3, s3.history()), 600, 1e-6, k_plain)
                if loop3 == 0 and not pc3: k_plain = k_nb
                key = (bool(st3.converged), bool(st3.breakdown), k_nb[1] is not None)
                nd_classes[key] = nd_classes.get(key, 0) + 1
                msgs += ["non-dominant n=%d nnz=%d loop%d pc%d: %s" % (A3.n, A3.nnz, loop3, pc3, m) for m in found]
            except cm.CudamatError as e:
                msgs.append("non-dominant loop%d pc%d: %s" % (loop3, pc3, e))
            for a in (db, dxx): a.free()
            s3.close()
    if msgs:
        bad += 1
        print("case %d n=%d per=%g base=%d: %s" % (case, n, per, base, "; ".join(msgs)), flush=True)
print("soak: %d cases, %d with findings; solves by loop form (0 five launches, 1 three, 2 one): %s" % (ncase, bad, sorted(forms.items())) + "; SpMV runs on a value dictionary: %d" % dicts
      + "; stencil systems through the row-pattern forms: %s" % sorted(stencils.items())
      + "; non-dominant systems: %d, their solves by (converged, breakdown, oracle on rounding noise): %s" % (nd_total, sorted(nd_classes.items(), key=str)))
