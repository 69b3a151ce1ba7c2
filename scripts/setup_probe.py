"""Where does a drop-in call's time go?  cudamat_solve() on host arrays at C4 (or --rows), three calls, verbose stamps.
   python scripts/setup_probe.py [--rows N] [--per-row K] [--precond]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("CUDAMAT_VERBOSE", "1")
import numpy as np, torch
import cuda_mat_amd as cm
from cuda_mat_amd import api
ap = argparse.ArgumentParser(); ap.add_argument("--rows", type=int, default=10_000_000); ap.add_argument("--per-row", type=int, default=50)
ap.add_argument("--precond", action="store_true"); a = ap.parse_args()
dev = torch.device("cuda", 0); n = a.rows
ctx = cm.Context(0, stream=torch.cuda.current_stream().cuda_stream)
rn = cm.lib().cudamat_rand_row_nnz(n, a.per_row)
rp = torch.empty(n + 1, dtype=torch.int32, device=dev); ci = torch.empty(n * rn, dtype=torch.int32, device=dev); va = torch.empty(n * rn, dtype=torch.float64, device=dev)
ctx.gen_rand_rows(n, a.per_row, 0x5EED, 0, n, 0, rp, ci, va)
xs = torch.empty(n, dtype=torch.float64, device=dev); ctx.gen_xstar(0, n, 0x5EEE, xs)
s = cm.Solver(ctx, n, n, n * rn, rp, ci, va, 0); b = torch.empty(n, dtype=torch.float64, device=dev); s.spmv(xs, b); s.close()
rph, cih, vah, bh = rp.cpu().numpy(), ci.cpu().numpy(), va.cpu().numpy(), b.cpu().numpy()
del rp, ci, va; torch.cuda.empty_cache()
for k in range(3):
    t = time.perf_counter()
    x, st = api._solve(n, n * rn, vah, rph, cih, None, None, bh, cm.PRECOND_ILU0 if a.precond else cm.PRECOND_NONE, cm.LOOP_PBICGSTAB, 200, 1e-8, False)
    print("call %d: wall %.3f s  upload %.3f setup %.3f tune %.3f analysis %.3f factor %.3f loop %.3f total %.3f  reused %d iters %d"
          % (k, time.perf_counter() - t, st.t_upload, st.t_setup, st.t_tune, st.t_analysis, st.t_factor, st.t_solve, st.t_total, st.plan_reused, st.iters), flush=True)
