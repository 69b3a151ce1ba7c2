"""Row-block sharding of the BiCGSTAB path across the GPUs of one node.

The reference is single-GPU; this is the multi-GPU design of SURVEY.md section 8e:
rank g owns the contiguous rows [g*per, min(n, (g+1)*per)), per = ceil(n / world), and the
matching slice of every vector.  The only data-path exchanges are
  * an all-gather of the SpMV input vector (2 per iteration), and
  * an all-reduce of 1-3 scalars (3 per iteration),
issued by the C++ loop (csrc/solver.hip) through the two callbacks of `cudamat_comm`.
TorchComm implements them with torch.distributed: backend "nccl" (= RCCL over xGMI) for
device buffers, "gloo" for host buffers (CPU tests of the plumbing).
"""
import ctypes as C

import numpy as np

from ._lib import ALLGATHER_FN, ALLREDUCE_FN, GATHER_PART_FN, GATHER_WINDOW_FN, Comm

RCCL_ID_BYTES = 384     # CUDAMAT_RCCL_ID_BYTES


def shard_rows(n, world, rank):
    """(row0, row1, per): the uniform row blocks cudamat_solver_set_comm requires"""
    per = (n + world - 1) // world
    row0 = min(n, per * rank)
    row1 = min(n, row0 + per)
    return row0, row1, per


class _CudaPtr:
    """zero-copy view of a raw device pointer for torch.as_tensor"""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (ptr, False),
                                         "version": 2, "strides": None}


class TorchComm:
    """cudamat_comm backed by torch.distributed (one process per GPU)"""

    def __init__(self, group=None, device=None, pieces=False):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.device = device      # torch.device("cuda", i) or None for host pointers (gloo)
        self._views = {}
        self.n_allgather = 0
        self.n_allreduce = 0
        self.error = None
        self.timing = False           # bracket every collective with events on the current stream
        self._ev = {"allgather": [], "allreduce": []}
        self._ag = ALLGATHER_FN(self._allgather)
        self._ar = ALLREDUCE_FN(self._allreduce)
        self.n_parts = 0
        self.side = None
        self.n_windows = 0
        gw = GATHER_WINDOW_FN(self._gather_window) if device is not None else GATHER_WINDOW_FN()
        self._gw = gw
        if pieces and device is not None:
            # pieces of the gather on a side stream (cudamat_comm.gather_part / comm_stream)
            self.side = torch.cuda.Stream(device=device)
            self._gp = GATHER_PART_FN(self._gather_part)
            self.struct = Comm(self.rank, self.world, None, self._ag, self._ar, self._gp, self.side.cuda_stream,
                               ALLREDUCE_FN(), None, gw)
        else:
            self.struct = Comm(self.rank, self.world, None, self._ag, self._ar, GATHER_PART_FN(), None, ALLREDUCE_FN(), None, gw)

    def _gather_window(self, user, send, recv, stride, send_off, send_cnt, recv_off, recv_cnt):
        """windowed gather (halo) on the current stream: this rank sends send[send_off[q] .. + send_cnt[q]) to rank q and
        receives rank q's [recv_off[q] .. + recv_cnt[q]) into recv[q*stride + recv_off[q] ..)"""
        try:
            torch, dist = self.torch, self.dist
            nccl = dist.get_backend(self.group) == "nccl"
            ops, staged = [], []
            for d in range(1, self.world):
                to, frm = (self.rank + d) % self.world, (self.rank - d) % self.world
                if send_cnt[to] > 0:
                    t = self._view(send + 8 * send_off[to], send_cnt[to])
                    ops.append(dist.P2POp(dist.isend, t if nccl else t.cpu(), to, self.group))
                if recv_cnt[frm] > 0:
                    dst = self._view(recv + 8 * (stride * frm + recv_off[frm]), recv_cnt[frm])
                    if nccl:
                        ops.append(dist.P2POp(dist.irecv, dst, frm, self.group))
                    else:
                        buf = torch.empty(recv_cnt[frm], dtype=torch.float64)
                        staged.append((dst, buf))
                        ops.append(dist.P2POp(dist.irecv, buf, frm, self.group))
            if ops:
                reqs = dist.batch_isend_irecv(ops)
                if not nccl:
                    for r in reqs:
                        r.wait()
                    for dst, buf in staged:
                        dst.copy_(buf)
            self.n_windows += 1
            return 0
        except Exception as e:  # noqa: BLE001
            self.error = e
            return 1

    def _gather_part(self, user, send, recv, stride, offset, count):
        """rank q's send[offset, offset+count) -> recv[q*stride + offset, ...) on every other rank, on self.side.
        NCCL backend: one batch of isend/irecv (asynchronous, ordered on the side stream).  Other backends (gloo
        rehearsals on one GPU): staged through the host and synchronous -- same data movement, no overlap."""
        try:
            torch, dist = self.torch, self.dist
            mine = self._view(send + 8 * offset, count)
            with torch.cuda.stream(self.side):
                if dist.get_backend(self.group) == "nccl":
                    ops = []
                    for d in range(1, self.world):
                        to, frm = (self.rank + d) % self.world, (self.rank - d) % self.world
                        ops.append(dist.P2POp(dist.isend, mine, to, self.group))
                        ops.append(dist.P2POp(dist.irecv, self._view(recv + 8 * (stride * frm + offset), count), frm, self.group))
                    if ops:
                        dist.batch_isend_irecv(ops)          # completion is ordered on the side stream
                else:
                    self.side.synchronize()                  # the solver made this stream wait for the producer
                    host = mine.cpu()
                    bufs, reqs = {}, []
                    for d in range(1, self.world):
                        to, frm = (self.rank + d) % self.world, (self.rank - d) % self.world
                        bufs[frm] = torch.empty(count, dtype=torch.float64)
                        reqs.append(dist.isend(host, to, group=self.group))
                        reqs.append(dist.irecv(bufs[frm], frm, group=self.group))
                    for r in reqs:
                        r.wait()
                    for frm, t in bufs.items():
                        self._view(recv + 8 * (stride * frm + offset), count).copy_(t)
                    self.side.synchronize()
            self.n_parts += 1
            return 0
        except Exception as e:  # noqa: BLE001
            self.error = e
            return 1

    def _view(self, ptr, count):
        key = (ptr, count)
        t = self._views.get(key)
        if t is None:
            if self.device is not None:
                t = self.torch.as_tensor(_CudaPtr(ptr, count), device=self.device)
            else:
                buf = (C.c_double * count).from_address(ptr)
                t = self.torch.from_numpy(np.frombuffer(buf, dtype=np.float64, count=count))
            self._views[key] = t
        return t

    # The loop enqueues kernels on the context's stream, which must be torch's CURRENT stream
    # on this device (Context(stream=torch.cuda.current_stream().cuda_stream)): the NCCL
    # backend orders its collective after the work already queued on the current stream and
    # makes the current stream wait for it; no host synchronisation happens here.
    def _mark(self, kind, first):
        if self.timing and self.device is not None:
            e = self.torch.cuda.Event(enable_timing=True)
            e.record()
            if first:
                self._ev[kind].append([e, None])
            else:
                self._ev[kind][-1][1] = e

    def reset_timing(self, on=True):
        self.timing = on
        self._ev = {"allgather": [], "allreduce": []}
        self.n_allgather = self.n_allreduce = 0

    def elapsed_ms(self):
        """(all-gather ms, all-reduce ms) spent on the stream since reset_timing(); synchronises"""
        if self.device is not None:
            self.torch.cuda.synchronize(self.device)
        return tuple(sum(a.elapsed_time(b) for a, b in self._ev[k] if b is not None) for k in ("allgather", "allreduce"))

    def _allgather(self, user, send, recv, count):
        try:
            self._mark("allgather", True)
            self.dist.all_gather_into_tensor(self._view(recv, count * self.world), self._view(send, count),
                                             group=self.group)
            self._mark("allgather", False)
            self.n_allgather += 1
            return 0
        except Exception as e:  # noqa: BLE001 - must not unwind through the C frame
            self.error = e
            return 1

    def _allreduce(self, user, buf, count):
        try:
            self._mark("allreduce", True)
            self.dist.all_reduce(self._view(buf, count), op=self.dist.ReduceOp.SUM, group=self.group)
            self._mark("allreduce", False)
            self.n_allreduce += 1
            return 0
        except Exception as e:  # noqa: BLE001
            self.error = e
            return 1


class RcclComm:
    """The library's own communicator (csrc/comm_rccl.hip): RCCL bound inside libcudamat_hip.so, so the solver's
    loop calls ncclAllGather / ncclAllReduce / grouped ncclSend+ncclRecv itself and never re-enters Python.
    Python only carries the 256 id bytes from rank 0 to the others, through `bcast(bytes_or_None) -> bytes`
    (default: torch.distributed.broadcast_object_list on whatever process group exists, e.g. gloo)."""

    def __init__(self, ctx, rank, world, bcast=None):
        from . import _lib
        L = _lib.lib()
        self.rank, self.world = int(rank), int(world)
        ident = None
        if self.rank == 0:
            buf = C.create_string_buffer(RCCL_ID_BYTES)
            _lib.check(L.cudamat_rccl_unique_id(buf))
            ident = buf.raw
        if self.world > 1:
            if bcast is None:
                import torch.distributed as dist

                def bcast(b):
                    box = [b]
                    dist.broadcast_object_list(box, src=0)
                    return box[0]
            ident = bcast(ident)
        assert ident is not None and len(ident) == RCCL_ID_BYTES
        self.struct = Comm()
        _lib.check(L.cudamat_rccl_comm_create(ctx.h, ident, self.rank, self.world, C.byref(self.struct)))
        self.native = True

    def close(self):
        from . import _lib
        if self.struct is not None:
            _lib.check(_lib.lib().cudamat_rccl_comm_destroy(C.byref(self.struct)))
            self.struct = None
