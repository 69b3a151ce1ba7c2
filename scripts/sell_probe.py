"""GPU box: what the SpMV tuner measures for the CSR forms vs SELL-C-sigma on matrices with varying row lengths
(CUDAMAT_VERBOSE prints the timings to stderr).  python scripts/sell_probe.py"""
import os, sys
import numpy as np
import scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CUDAMAT_VERBOSE"] = "1"
import cuda_mat_amd as cm

ctx = cm.Context(0)
rng = np.random.default_rng(1)


def run(name, n, lens, band):
    rows = np.repeat(np.arange(n), lens)
    cols = (rows + rng.integers(-band, band, rows.size)) % n
    S = sp.csr_matrix((np.ones(rows.size), (rows, cols)), shape=(n, n))
    S.sum_duplicates()
    S.sort_indices()
    print("== %s: n=%d nnz=%d mean %.1f max %d" % (name, n, S.nnz, S.nnz / n, np.diff(S.indptr).max()), file=sys.stderr, flush=True)
    s = cm.Solver.from_host_csr(ctx, S.indptr.astype(np.int32), S.indices.astype(np.int32), S.data)
    s.spmv_mode()
    s.close()


n = 2_000_000
run("uniform 14..70, band 2000", n, rng.integers(14, 70, n), 2000)
run("uniform 14..70, scattered", n, rng.integers(14, 70, n), n // 2)
run("two classes 16 / 64 alternating", n, np.where(np.arange(n) % 2 == 0, 16, 64), 5000)
run("pareto", n, np.minimum(13 + (rng.pareto(1.5, n) * 6).astype(np.int64), 3000), 5000)
run("uniform 13..20 (mild)", n, rng.integers(13, 21, n), 2000)
run("equal rows of 50, band 5000", n, np.full(n, 50), 5000)
run("equal rows of 20, band 2000", n, np.full(n, 20), 2000)
run("equal rows of 9, band 500", n, np.full(n, 9), 500)
