import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cuda_mat_amd as cm
from oracle import oracle as O
ctx = cm.Context(0)
for name in ("mat10000", "mat900"):
    A = O.mtx_load(os.path.join("tests/golden", name + ".mtx"))
    xs = 1.0 + np.sin(np.arange(A.n)); b = O.spmv(A, xs)
    for loop in (0, 1):
        out = {}
        for res in ("1", "0"):
            os.environ["CUDAMAT_RESIDENT"] = res
            s = cm.Solver.from_host_csr(ctx, A.rowptr, A.colidx, A.val)
            db, dx = ctx.array(b), ctx.empty(A.n)
            st = s.solve(db, dx, loop=loop, maxit=2000, tol=1e-8, flags=cm.FLAG_X0_ONES)
            h = s.history()
            out[res] = (st, h, dx.download())
            print(name, "loop", loop, "resident", res, "form", st.loop_form, "iters", st.iters, "half", st.half_exit, "conv", st.converged,
                  "nrm", st.nrm, "tolabs", 1e-8 * st.nrm0, "hist tail", h[-3:])
            s.close()
        h1, h0 = out["1"][1], out["0"][1]
        k = min(len(h1), len(h0))
        rel = np.abs(h1[:k] - h0[:k]) / np.abs(h0[:k])
        print("   max rel hist diff over first", k, ":", rel.max(), "at", int(rel.argmax()), " first 5:", rel[:5])
