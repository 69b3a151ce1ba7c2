"""GPU box: what a drop-in call (cudamat_solve on host arrays, the C4 matrix) costs, first call and same matrix again, after a
throw-away call (clocks, page tables and the runtime's buffers are cold in a new process); CUDAMAT_VERBOSE=1 prints the stage
stamps of the first calls.   usage: python scripts/upload_probe.py [rows] [rounds] [full|small|none]"""
import os
import sys
import time

os.environ.pop("OMP_PROC_BIND", None)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import cuda_mat_amd as cm
from cuda_mat_amd import api
from oracle import oracle as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
warm = sys.argv[3] if len(sys.argv) > 3 else "full"          # the throw-away call: full (the same arrays) | small (a 4096-row system) | none
O.set_num_threads(min(32, len(os.sched_getaffinity(0))))
A = O.rand_rows(n, 50, 0x5EED)
xs = O.xstar(n, 0x5EEE)
b = O.spmv(A, xs)
print("affinity: %d CPUs" % len(os.sched_getaffinity(0)), flush=True)
os.environ["CUDAMAT_VALUE_DICT"] = "0"
# a throw-away call first: clocks, page tables and the runtime's own buffers are cold in a new process
if warm == "full":
    api._solve(n, A.nnz, A.val, A.rowptr, A.colidx, None, None, b, cm.PRECOND_NONE, cm.LOOP_PBICGSTAB, 200, 1e-8, False)
elif warm == "small":
    As = O.rand_rows(4096, 50, 0x11)
    api._solve(As.n, As.nnz, As.val, As.rowptr, As.colidx, None, None, O.spmv(As, O.xstar(As.n, 3)), cm.PRECOND_NONE, cm.LOOP_PBICGSTAB, 200, 1e-8, False)
print("---- throw-away call: %s" % warm, flush=True)
for r in range(rounds):
    for again in (0, 1):
        if not again:
            cm.lib().cudamat_plan_cache_clear()
        t0 = time.perf_counter()
        x, st = api._solve(n, A.nnz, A.val, A.rowptr, A.colidx, None, None, b, cm.PRECOND_NONE, cm.LOOP_PBICGSTAB, 200, 1e-8, False)
        dt = time.perf_counter() - t0
        print("round %d %s: end to end %.4f s  upload %.4f s (%.1f GB/s)  set-up not hidden %.4f s  loop %.4f s  iters %d  err %.1e"
              % (r, "same matrix again" if again else "first call       ", dt, st.t_upload, 6.12 * n / 1e7 / max(st.t_upload, 1e-9),
                 st.t_setup, st.t_solve, st.iters, float(np.abs(x - xs).max())), flush=True)
cm.lib().cudamat_plan_cache_clear()
