"""Test infrastructure for the row-sharded path.

sharded_pbicgstab(): a Python restatement of the SHARDED branch of the C++ loop
(cuda_mat_amd/csrc/solver.hip, cudamat_solver_solve) with the oracle as the local compute
backend.  It issues exactly the collectives the C++ loop issues, in the same order and with the
same payloads, THROUGH THE SAME cudamat_comm callbacks (ctypes function pointers), so a gloo
world exercises shard_rows(), TorchComm's pointer->tensor plumbing and the collective sequence
without a GPU.  The C++ loop itself is exercised on one GPU by tests/test_gpu_dist.py with
in-process emulated ranks (ThreadComm below).
"""
import ctypes as C
import threading

import numpy as np


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def sharded_pbicgstab(O, comm, A_loc, n, per, b_loc, maxit, tol):
    """comm: a cuda_mat_amd._lib.Comm; A_loc: local row block (global column ids, base 0);
    returns (x_loc, iters, half_exit, converged, hist)"""
    nloc = A_loc.n
    world = comm.world

    def gather(v):
        send = np.zeros(per)
        send[:nloc] = v
        recv = np.empty(per * world)
        assert comm.allgather(None, _p(send), _p(recv), per) == 0
        return recv

    def allreduce(vals):
        buf = np.array(vals, dtype=np.float64)
        assert comm.allreduce(None, _p(buf), len(buf)) == 0
        return buf

    def spmv(v_loc):
        return O.spmv(A_loc, gather(v_loc))

    x = np.ones(nloc)
    r = b_loc - spmv(x)                                  # solver.hip: spmv_local + launch_init
    rw, p = r.copy(), r.copy()
    rho_full = allreduce([r @ r, r @ r])                 # red[4..5]
    nrm0 = np.sqrt(rho_full[1])
    tolabs = tol * nrm0
    rho, rhop, alpha, omega = rho_full[0], 1.0, 1.0, 1.0
    v = np.zeros(nloc)
    hist = []
    it, half_exit, converged = 0, 0, 0
    full = rho_full
    for k in range(maxit):
        # launch_update_p: full-step test of the previous iteration, then rho / beta / p
        if it > 0:
            nrm = np.sqrt(full[1])
            hist.append(nrm)
            if nrm < tolabs:
                converged = 1
                break
        rhop, rho = rho, full[0]
        if it > 0:
            beta = (rho / rhop) * (alpha / omega)
            p = r + beta * (p - omega * v)
        v = spmv(p)                                      # allgather(p) + SpMV + (rw.v) partials
        rv = allreduce([rw @ v])[0]                      # red[0]
        alpha = rho / rv
        r = r - alpha * v                                # launch_half
        x = x + alpha * p
        half2 = r @ r
        t = spmv(r)                                      # allgather(r) + SpMV + (t.r, t.t) partials
        red = allreduce([half2, t @ r, t @ t])           # ONE all-reduce: red[1..3]
        nrm = np.sqrt(red[0])                            # launch_check(CHECK_HALF)
        hist.append(nrm)
        if nrm < tolabs:
            half_exit, converged = 1, 1
            break
        omega = red[1] / red[2]                          # launch_full
        x = x + omega * r
        r = r - omega * t
        it += 1
        full = allreduce([rw @ r, r @ r])                # red[4..5]
    else:
        nrm = np.sqrt(full[1])
        if it > 0:
            hist.append(nrm)
            converged = int(nrm < tolabs)
    return x, it, half_exit, converged, np.array(hist)


class ThreadGroup:
    """shared state of W emulated ranks living in one process on one GPU"""

    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.send = [None] * world
        self.vals = [None] * world


class ThreadComm:
    """cudamat_comm whose collectives are host-synchronised copies between the ranks' buffers:
    slow, but it drives the real C++ sharded loop with W > 1 on a single GPU."""

    def __init__(self, cm, group, rank, ctx, pieces=False, side_reduce=False, windows=False):
        """pieces=True also offers gather_part (the overlapped gather) on a stream of its own: a second Context;
        side_reduce=True offers allreduce_side on a third one (the pipelined loop's reductions beside an SpMV)"""
        from cuda_mat_amd._lib import ALLGATHER_FN, ALLREDUCE_FN, GATHER_PART_FN, GATHER_WINDOW_FN, Comm
        self.cm, self.g, self.rank, self.ctx = cm, group, rank, ctx
        self.n_allgather = self.n_allreduce = self.n_parts = self.n_side = self.n_windows = 0
        self.window_doubles = 0
        self._ag = ALLGATHER_FN(self._allgather)
        self._ar = ALLREDUCE_FN(self._allreduce)
        self.cctx = self.rctx = None
        gp, gstream, ars, rstream = GATHER_PART_FN(), None, ALLREDUCE_FN(), None
        if pieces:
            self.cctx = cm.Context(ctx.device)
            gstream = C.c_void_p()
            assert cm.lib().cudamat_ctx_stream(self.cctx.h, C.byref(gstream)) == 0
            self._gp = gp = GATHER_PART_FN(self._gather_part)
        if side_reduce:
            self.rctx = cm.Context(ctx.device)
            rstream = C.c_void_p()
            assert cm.lib().cudamat_ctx_stream(self.rctx.h, C.byref(rstream)) == 0
            self._ars = ars = ALLREDUCE_FN(self._allreduce_side)
        gw = GATHER_WINDOW_FN()
        if windows:
            self._gw = gw = GATHER_WINDOW_FN(self._gather_window)
        self.struct = Comm(rank, group.world, None, self._ag, self._ar, gp, gstream, ars, rstream, gw)

    def _gather_window(self, user, send, recv, stride, send_off, send_cnt, recv_off, recv_cnt):
        """windowed gather (halo): only [recv_off[q], +recv_cnt[q]) of every other slice; checks that what the peers
        say they send is what this rank expects to receive"""
        try:
            L = self.cm.lib()
            W = self.g.world
            self.ctx.sync()
            self.g.send[self.rank] = (send, [(send_off[q], send_cnt[q]) for q in range(W)])
            self.g.barrier.wait()
            for q in range(W):
                if q == self.rank or recv_cnt[q] == 0:
                    continue
                ptr, ranges = self.g.send[q]
                assert ranges[self.rank] == (recv_off[q], recv_cnt[q]), "sender and receiver disagree on a window"
                assert L.cudamat_d2d(self.ctx.h, recv + 8 * (stride * q + recv_off[q]), ptr + 8 * recv_off[q],
                                     8 * recv_cnt[q]) == 0
                self.window_doubles += recv_cnt[q]
            self.ctx.sync()
            self.g.barrier.wait()
            self.n_windows += 1
            return 0
        except Exception:                                 # noqa: BLE001
            self.g.barrier.abort()
            return 1

    def close(self):
        for c in (self.cctx, self.rctx):
            if c is not None:
                c.close()
        self.cctx = self.rctx = None

    def _allreduce_side(self, user, buf, count):
        """the same sum as _allreduce, ordered on the reduce stream (synchronised here: the emulation is host-driven)"""
        try:
            self.rctx.sync()              # the partial sums were reduced into buf on this stream
            rc = self._reduce_with(self.rctx, buf, count)
            self.n_side += 1
            return rc
        except Exception:                                 # noqa: BLE001
            self.g.barrier.abort()
            return 1

    def _gather_part(self, user, send, recv, stride, offset, count):
        try:
            L = self.cm.lib()
            self.cctx.sync()          # the solver made this stream wait for my producer kernels: my piece is final
            self.g.send[self.rank] = send
            self.g.barrier.wait()
            for r in range(self.g.world):
                if r != self.rank:
                    assert L.cudamat_d2d(self.cctx.h, recv + 8 * (stride * r + offset), self.g.send[r] + 8 * offset,
                                         8 * count) == 0
            self.cctx.sync()
            self.g.barrier.wait()                         # nobody overwrites a send buffer early
            self.n_parts += 1
            return 0
        except Exception:                                 # noqa: BLE001
            self.g.barrier.abort()
            return 1

    def _allgather(self, user, send, recv, count):
        try:
            L = self.cm.lib()
            self.ctx.sync()                               # my producer kernels are done
            self.g.send[self.rank] = send
            self.g.barrier.wait()
            for r in range(self.g.world):
                assert L.cudamat_d2d(self.ctx.h, recv + 8 * count * r, self.g.send[r], 8 * count) == 0
            self.ctx.sync()
            self.g.barrier.wait()                         # nobody overwrites a send buffer early
            self.n_allgather += 1
            return 0
        except Exception:                                 # noqa: BLE001
            self.g.barrier.abort()
            return 1

    def _reduce_with(self, ctx, buf, count):
        L = self.cm.lib()
        mine = np.empty(count)
        assert L.cudamat_d2h(ctx.h, _p(mine), buf, 8 * count) == 0
        self.g.vals[self.rank] = mine
        self.g.barrier.wait()
        tot = np.zeros(count)
        for r in range(self.g.world):                     # fixed order: identical on every rank
            tot = tot + self.g.vals[r]
        self.g.barrier.wait()
        assert L.cudamat_h2d(ctx.h, buf, _p(tot), 8 * count) == 0
        return 0

    def _allreduce(self, user, buf, count):
        try:
            rc = self._reduce_with(self.ctx, buf, count)
            self.n_allreduce += 1
            return rc
        except Exception:                                 # noqa: BLE001
            self.g.barrier.abort()
            return 1
