"""ad-hoc: where does host time go in the small-problem path?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
t0 = time.perf_counter()
import cuda_mat_amd as cm
from oracle import oracle as O
def T(msg, t):
    print("%-40s %.4f s" % (msg, time.perf_counter() - t), flush=True)
    return time.perf_counter()
t = T("import", t0)
ctx = cm.Context(0); t = T("Context()", t)
A = O.mtx_load("tests/golden/mat900.mtx"); t = T("oracle load", t)
a = ctx.array(A.val); t = T("ctx.array first", t)
a2 = ctx.array(A.val); t = T("ctx.array second", t)
a.free(); t = T("free", t)
s = cm.Solver.from_host_csr(ctx, A.rowptr, A.colidx, A.val); t = T("Solver.from_host_csr", t)
b = O.spmv(A, 1 + np.sin(np.arange(A.n)))
db, dx = ctx.array(b), ctx.array(np.ones(A.n)); t = T("arrays", t)
st = s.solve(db, dx, maxit=2000, tol=1e-8); t = T("solve 1 (%d it, t_solve %.4f)" % (st.iters, st.t_solve), t)
dx.upload(np.ones(A.n))
st = s.solve(db, dx, maxit=2000, tol=1e-8); t = T("solve 2 (%d it, t_solve %.4f)" % (st.iters, st.t_solve), t)
h = s.history(); t = T("history", t)
s.close(); t = T("solver close", t)
for i in range(3):
    ok, x, dt, st = cm.bicgstab(A.n, A.nnz, A.val, A.rowptr, A.colidx, b, 2000, 1e-8); t = T("cm.bicgstab host-pointer (dtAlg %.4f)" % dt, t)
v = ctx.dot(A.n, db, db); t = T("dot", t)
ctx.close(); t = T("ctx close", t)
