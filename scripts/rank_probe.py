"""GPU box, ONE GPU: what one rank of a G-rank sharded C4 solve computes per iteration, with every exchange skipped
(cudamat_comm_dry_create): shard-shaped blocked SpMV incl. the per-piece phase-1 launches of the overlapped gather,
fused vector kernels on n/G rows, the reduce kernels.  An upper bound for the strong-scaling curve: the exchanges come
on top (or hide behind the SpMV).  python scripts/rank_probe.py [G ...]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cuda_mat_amd as cm
from cuda_mat_amd._lib import Comm, check

N = 10_000_000
ctx = cm.Context(0)
for G in [int(a) for a in sys.argv[1:]] or (1, 2, 4, 8):
    per = (N + G - 1) // G
    nloc = per
    nnz = nloc * 50
    rp, ci, va = ctx.empty(nloc + 1, np.int32), ctx.empty(nnz, np.int32), ctx.empty(nnz)
    ctx.gen_rand_rows(N, 50, 0x5EED, 0, nloc, 0, rp, ci, va)
    for loop, overlap in ((cm.LOOP_PBICGSTAB, "1"), (cm.LOOP_PBICGSTAB, "0"), (cm.LOOP_PIPELINED, "1")):
        if G == 1 and overlap == "0":
            continue
        os.environ["CUDAMAT_OVERLAP"] = overlap
        s = cm.Solver(ctx, nloc, N, nnz, rp, ci, va, 0)
        comm = Comm()
        if G > 1:
            check(cm.lib().cudamat_comm_dry_create(ctx.h, 0, G, C.byref(comm)))
            s.set_comm(comm)
        b, x = ctx.empty(nloc), ctx.empty(nloc)
        ctx.gen_xstar(0, nloc, 3, b)
        st = s.solve(b, x, loop=loop, maxit=10, tol=1e-8, flags=cm.FLAG_NO_EXIT | cm.FLAG_X0_ONES)      # warm-up + tuner
        t = ctx.timer()
        ctx.sync()
        t.start()
        st = s.solve(b, x, loop=loop, maxit=25, tol=1e-8, flags=cm.FLAG_NO_EXIT | cm.FLAG_X0_ONES)
        t.stop()
        ms = t.elapsed_ms() / 25
        stp = s.solve(b, x, loop=loop, maxit=25, tol=1e-8, flags=cm.FLAG_NO_EXIT | cm.FLAG_X0_ONES | cm.FLAG_PROFILE)
        print("G=%d  %-9s %s  %7.3f ms/iteration (%6.1f it/s if the exchanges were free)   SpMV %.3f ms (alone %.3f), mode %d"
              % (G, "pipelined" if loop == cm.LOOP_PIPELINED else "standard", "pieces" if overlap == "1" else "plain ",
                 ms, 1e3 / ms, stp.ms_spmv / max(stp.n_spmv, 1), stp.ms_spmv_alone, s.spmv_mode()), flush=True)
        s.close()
        if G > 1:
            check(cm.lib().cudamat_comm_dry_destroy(C.byref(comm)))
        b.free(); x.free()
    for a in (rp, ci, va):
        a.free()
