"""GPU box: replay ONE system of tests/soak.py (same draws) and put the residual histories of the GPU loop and of the
oracle's restatement side by side: where do they part, how do they end?   python tests/soak_case.py SEED CASE LOOP [PRECOND]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
import cuda_mat_amd as cm
from oracle import oracle as O
seed, target, loop = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
precond = int(sys.argv[4]) if len(sys.argv) > 4 else 0
rng = np.random.default_rng(seed)
for case in range(target + 1):
    n = int(rng.integers(1, 40000)); per = float(rng.choice([1.5, 4, 9, 30, 80])); nnz_t = int(min(n * per, 3e6))
    if case % 3 == 2:
        lens = np.minimum(1 + (rng.pareto(1.2, n) * per / 4).astype(np.int64), max(n // 2, 1))
        lens[rng.integers(0, n, 3)] = max(n // 3, 1)
        ri = np.repeat(np.arange(n), lens)[:int(3e6)]; nnz_t = ri.size
    else:
        ri = rng.integers(0, n, nnz_t)
    cj = rng.integers(0, n, nnz_t); vals = rng.uniform(-1, 1, nnz_t)
    if case % 4 == 1: vals = rng.choice(np.array([-1.0, -0.5, 0.25, 2.0]), nnz_t)
    if case == target:
        S = sp.csr_matrix((vals, (ri, cj)), shape=(n, n)); S.sum_duplicates(); S.setdiag(0); S.eliminate_zeros()
        rowsum = np.asarray(abs(S).sum(axis=1)).ravel()
    diag_draw = None if case % 4 == 1 else rng.random(n)
    base = int(rng.integers(0, 2)); x = rng.standard_normal(n); xsr = rng.random(n); rhs = rng.standard_normal(n)
diag = 8.0 * np.ceil((rowsum + 1.5) / 8.0) if target % 4 == 1 else 1.0 + diag_draw + rowsum
S = (S + sp.diags(diag)).tocsr(); S.sort_indices()
A = O.Csr(n, (S.indptr + base).astype(np.int32), (S.indices + base).astype(np.int32), S.data.copy(), n)
b = O.spmv(A, 1.0 + xsr)
vm = O.ilu0(A) if precond else None
if loop == 2: xo, so, ho = O.pipelined_bicgstab(A, b, vm=vm, maxit=500, tol=1e-9, want_hist=True)
else: xo, so, ho = O.pbicgstab(A, b, vm=vm, maxit=500, tol=1e-9, want_hist=True)
ctx = cm.Context(0)
s = cm.Solver.from_host_csr(ctx, A.rowptr, A.colidx, A.val)
db, dx = ctx.array(b), ctx.array(np.ones(n))
st = s.solve(db, dx, precond=precond, loop=loop, maxit=500, tol=1e-9)
hg = s.history()
ho = ho[~np.isnan(ho)]
m = min(len(hg), len(ho))
rel = np.abs(hg[:m] - ho[:m]) / np.maximum(np.abs(ho[:m]), 1e-300)
print("n %d nnz %d  GPU %d iterations (restarts %d), oracle %d; history entries %d / %d" % (n, A.nnz, st.iters, st.restarts, so.iters, len(hg), len(ho)))
for thr in (1e-12, 1e-9, 1e-6, 1e-3, 1e-1):
    w = np.nonzero(rel > thr)[0]
    print("  first history entry differing by more than %g: %s" % (thr, ("#%d (iteration %d)" % (w[0], w[0] // 2)) if w.size else "none"))
print("  target %.3e;  last 12 entries GPU   : %s" % (1e-9 * so.nrm0, " ".join("%.2e" % v for v in hg[-12:])))
print("                      last 12 entries oracle: %s" % " ".join("%.2e" % v for v in ho[-12:]))
k = max(0, len(hg) - 12)
print("  oracle at the GPU's last 12 positions       : %s" % " ".join("%.2e" % v for v in ho[k:k + 12]))
