"""GPU box: what a drop-in call (cudamat_solve on host arrays, the C4 matrix) costs with the caller's memory read directly by the
runtime (CUDAMAT_UPLOAD_THREADS=0, the default) and staged through pinned buffers by N host threads -- alternating, in a
process whose threads are not bound to one core (no torch, OMP_PROC_BIND unset).   usage: python scripts/upload_probe.py [rows] [rounds]"""
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
O.set_num_threads(min(32, len(os.sched_getaffinity(0))))
A = O.rand_rows(n, 50, 0x5EED)
xs = O.xstar(n, 0x5EEE)
b = O.spmv(A, xs)
print("affinity: %d CPUs" % len(os.sched_getaffinity(0)), flush=True)
os.environ["CUDAMAT_VALUE_DICT"] = "0"
# a throw-away call first: clocks, page tables and the runtime's own buffers are cold in a new process
api._solve(n, A.nnz, A.val, A.rowptr, A.colidx, None, None, b, cm.PRECOND_NONE, cm.LOOP_PBICGSTAB, 200, 1e-8, False)
print("---- warm", flush=True)
for r in range(rounds):
    for threads in (os.environ.get("PROBE_THREADS") or "0,4,8,12").split(","):
        os.environ["CUDAMAT_UPLOAD_THREADS"] = threads.split(":")[0]
        os.environ.pop("CUDAMAT_UPLOAD_PIECE_MB", None)
        if ":" in threads:
            os.environ["CUDAMAT_UPLOAD_PIECE_MB"] = threads.split(":")[1]
        for again in (0, 1):
            if not again:
                cm.lib().cudamat_plan_cache_clear()
            t0 = time.perf_counter()
            x, st = api._solve(n, A.nnz, A.val, A.rowptr, A.colidx, None, None, b, cm.PRECOND_NONE, cm.LOOP_PBICGSTAB, 200, 1e-8, False)
            dt = time.perf_counter() - t0
            print("round %d threads %5s %s: end to end %.4f s  upload %.4f s (%.1f GB/s)  exposed set-up %.4f s  loop %.4f s  iters %d  err %.1e"
                  % (r, threads, "same matrix again" if again else "first call       ", dt, st.t_upload, 6.12 * n / 1e7 / max(st.t_upload, 1e-9),
                     st.t_setup, st.t_solve, st.iters, float(np.abs(x - xs).max())), flush=True)
cm.lib().cudamat_plan_cache_clear()
