"""Host-side mirror of the reference's interface for the BiCGSTAB path, over the C ABI.

Names and argument meaning follow pbicgstab.h / mmio_wrapper.h of the reference:
  bicgstab(n, nnz, A, iA, jA, b, maxit, tol, debug)                  pbicgstab.h:113
  bicgstab_d(n, nnz, A0, iA0, jA0, d, x0, b, maxit, tol, debug)      pbicgstab.h:116 (overload)
  bicgstab_lu_precond(n, nnz, A, iA, jA, b, maxit, tol, debug)       pbicgstab.h:119
  loadMMSparseMatrix(filename, elem_type, csrFormat)                 mmio_wrapper.h:133
  toDenseVector(n, nnz, A, IA)                                       pbicgstab.cu:1101
Each solver returns (ok, x, dtAlg, stats): `ok` is the reference's bool, x the
solution (written even when not converged), dtAlg the loop's wall seconds.

Context / DeviceArray / Solver expose the HBM-resident operation used by the parity
tests and bench.py.  All compute goes through libcudamat_hip.so; there is no fallback.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import (FLAG_DEBUG, FLAG_NO_EXIT, FLAG_PROFILE, FLAG_X0_ONES, LOOP_PBICGSTAB,
                   LOOP_PBICGSTAB2, LOOP_PIPELINED, PRECOND_BLOCK_ILU0, PRECOND_ILU0, PRECOND_NONE, Comm, CudamatError, Stats, check)


def _np(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def _vp(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


# ----------------------------------------------------------------- drop-in functions
_GPUS = 1


def use_gpus(ngpu):
    """spread the three drop-in solves below over `ngpu` GPUs of the node (uniform row blocks, one host thread and
    one RCCL rank per device: cudamat_solve_sharded); the ILU(0) entry point then factors every GPU's diagonal block
    (block-Jacobi).  Mirrors cudamat_use_gpus of include/pbicgstab.h.  Default 1 = the reference's behaviour."""
    global _GPUS
    _GPUS = max(1, int(ngpu))


def _solve(n, nnz, A, iA, jA, d, x0, b, precond, loop, maxit, tol, debug):
    A, iA, jA, b = _np(A, np.float64), _np(iA, np.int32), _np(jA, np.int32), _np(b, np.float64)
    d = None if d is None else _np(d, np.float64)
    x0 = None if x0 is None else _np(x0, np.float64)
    if len(iA) != n + 1 or len(A) < nnz or len(jA) < nnz or len(b) != n:
        raise ValueError("array sizes do not match n / nnz")
    x = np.zeros(n)
    st = Stats()
    if _GPUS > 1:
        if precond == PRECOND_ILU0:
            precond = PRECOND_BLOCK_ILU0      # ILU(0) of the whole matrix does not shard
        check(_lib.lib().cudamat_solve_sharded(_GPUS, n, nnz, _vp(A), _vp(iA), _vp(jA), _vp(d), _vp(x0), _vp(b), precond,
                                               loop, maxit, tol, int(bool(debug)), _vp(x), C.byref(st)))
    else:
        check(_lib.lib().cudamat_solve(n, nnz, _vp(A), _vp(iA), _vp(jA), _vp(d), _vp(x0), _vp(b), precond,
                                       loop, maxit, tol, int(bool(debug)), _vp(x), C.byref(st)))
    return x, st


def bicgstab(n, nnz, A, iA, jA, b, maxit, tol, debug=False):
    """solve Ax = b, no preconditioner (pbicgstab.h:113).  The reference implementation is
    broken (its `r += b; r0 = r` lines are commented out, pbicgstab.cu:471-478); this is the
    intended maths: the d-variant loop with d = 0 and x0 = 1 (pbicgstab.cu:827-831)."""
    x, st = _solve(n, nnz, A, iA, jA, None, None, b, PRECOND_NONE, LOOP_PBICGSTAB2, maxit, tol, debug)
    return bool(st.converged), x, st.t_solve, st


def bicgstab_d(n, nnz, A0, iA0, jA0, d, x0, b, maxit, tol, debug=False):
    """solve (A0 + I*d) x = b from x0, no preconditioner (pbicgstab.h:116)."""
    x, st = _solve(n, nnz, A0, iA0, jA0, d, x0, b, PRECOND_NONE, LOOP_PBICGSTAB2, maxit, tol, debug)
    return bool(st.converged), x, st.t_solve, st


def bicgstab_lu_precond(n, nnz, A, iA, jA, b, maxit, tol, debug=False):
    """solve Ax = b with the ILU(0) preconditioner; A[i,i] != 0 required (pbicgstab.h:118-119).
    Like the reference (pbicgstab.cu:408) `ok` is True whenever the solve ran; convergence is
    in stats.converged."""
    x, st = _solve(n, nnz, A, iA, jA, None, None, b, PRECOND_ILU0, LOOP_PBICGSTAB, maxit, tol, debug)
    return True, x, st.t_solve, st


def loadMMSparseMatrix(filename, elem_type="d", csrFormat=True):
    """mmio_wrapper.h:133-142: returns (err, m, n, nnz, aVal, aRowInd, aColInd); err 0 ok / 1."""
    if elem_type != "d":
        raise ValueError("only elem_type 'd' is supported on this path")
    m, n, nnz = C.c_int(), C.c_int(), C.c_int()
    v, r, c = C.POINTER(C.c_double)(), C.POINTER(C.c_int)(), C.POINTER(C.c_int)()
    L = _lib.lib()
    err = L.cudamat_load_mtx(str(filename).encode(), int(bool(csrFormat)), C.byref(m), C.byref(n),
                             C.byref(nnz), C.byref(v), C.byref(r), C.byref(c))
    if err:
        return 1, 0, 0, 0, None, None, None
    nr = m.value + 1 if csrFormat else nnz.value
    nc = nnz.value if csrFormat else n.value + 1
    row = np.ctypeslib.as_array(r, shape=(nr,)).astype(np.int32, copy=True)
    col = np.ctypeslib.as_array(c, shape=(nc,)).astype(np.int32, copy=True) if nc else np.zeros(0, np.int32)
    val = (np.ctypeslib.as_array(v, shape=(nnz.value,)).astype(np.float64, copy=True)
           if nnz.value else np.zeros(0))
    for p in (v, r, c):
        L.cudamat_host_free(p)
    return 0, m.value, n.value, nnz.value, val, row, col


def toDenseVector(n, nnz, A, IA):
    A, IA = _np(A, np.float64), _np(IA, np.int32)
    out = np.empty(n)
    _lib.lib().cudamat_to_dense_vector(n, nnz, _vp(A), _vp(IA), _vp(out))
    return out


# ----------------------------------------------------------------- HBM-resident operation
class DeviceArray:
    """a typed allocation in HBM owned by a Context"""

    def __init__(self, ctx, n, dtype):
        self.ctx, self.n, self.dtype = ctx, int(n), np.dtype(dtype)
        p = C.c_void_p()
        check(_lib.lib().cudamat_malloc(ctx.h, self.n * self.dtype.itemsize, C.byref(p)))
        self.ptr = p.value

    @property
    def nbytes(self):
        return self.n * self.dtype.itemsize

    def upload(self, a):
        a = _np(a, self.dtype)
        assert a.size == self.n
        check(_lib.lib().cudamat_h2d(self.ctx.h, self.ptr, _vp(a), self.nbytes))
        return self

    def download(self):
        out = np.empty(self.n, self.dtype)
        check(_lib.lib().cudamat_d2h(self.ctx.h, _vp(out), self.ptr, self.nbytes))
        return out

    def zero(self):
        check(_lib.lib().cudamat_memset(self.ctx.h, self.ptr, 0, self.nbytes))
        return self

    def free(self):
        if self.ptr:
            _lib.lib().cudamat_free(self.ctx.h, self.ptr)
            self.ptr = None


def _ptr(a):
    """device pointer of a DeviceArray, a torch tensor, or a raw int"""
    if a is None:
        return None
    if isinstance(a, DeviceArray):
        return a.ptr
    if hasattr(a, "data_ptr"):
        return a.data_ptr()
    return int(a)


class Context:
    """device + stream (+ reduction workspace).  stream: raw hipStream_t value (e.g.
    torch.cuda.current_stream().cuda_stream) or None for a private stream."""

    def __init__(self, device=0, stream=None):
        h = C.c_void_p()
        check(_lib.lib().cudamat_ctx_create(device, stream, C.byref(h)))
        self.h = h
        self.device = device

    def close(self):
        if self.h:
            _lib.lib().cudamat_ctx_destroy(self.h)
            self.h = None

    def sync(self):
        check(_lib.lib().cudamat_ctx_sync(self.h))

    def set_option(self, name, value):
        """one switch of this context (csrc/config.h; `name` as in options_help(), with or without the CUDAMAT_ prefix).
        A context reads the CUDAMAT_* environment once, when it is created; afterwards only this call changes a switch.
        Whatever runs on the context after the call sees the new value."""
        check(_lib.lib().cudamat_ctx_set_option(self.h, str(name).encode(), str(value).encode()))
        return self

    def reset_options(self):
        """back to what a context created now would hold: the defaults overridden by the CUDAMAT_* environment"""
        check(_lib.lib().cudamat_ctx_reset_options(self.h))
        return self

    def empty(self, n, dtype=np.float64):
        return DeviceArray(self, n, dtype)

    def array(self, a, dtype=None):
        a = np.asarray(a)
        dtype = a.dtype if dtype is None else dtype
        return DeviceArray(self, a.size, dtype).upload(a)

    # elementary kernels (device pointers)
    def spmv(self, n, rowptr, colidx, val, base, x, y, alpha=1.0, beta=0.0, d=None):
        check(_lib.lib().cudamat_spmv(self.h, n, _ptr(rowptr), _ptr(colidx), _ptr(val), base, alpha,
                                      _ptr(x), _ptr(d), beta, _ptr(y)))

    def dot(self, n, x, y):
        out = self.empty(1)
        check(_lib.lib().cudamat_dot(self.h, n, _ptr(x), _ptr(y), out.ptr))
        v = out.download()[0]
        out.free()
        return v

    def nrm2(self, n, x):
        out = self.empty(1)
        check(_lib.lib().cudamat_nrm2(self.h, n, _ptr(x), out.ptr))
        v = out.download()[0]
        out.free()
        return v

    def axpy(self, n, alpha, x, y):
        check(_lib.lib().cudamat_axpy(self.h, n, alpha, _ptr(x), _ptr(y)))

    def scal(self, n, alpha, x):
        check(_lib.lib().cudamat_scal(self.h, n, alpha, _ptr(x)))

    # synthetic inputs generated in HBM
    def gen_rand_rows(self, n, per_row, seed, row0, row1, base, rowptr, colidx, val):
        check(_lib.lib().cudamat_gen_rand_rows(self.h, n, per_row, seed, row0, row1, base, _ptr(rowptr),
                                               _ptr(colidx), _ptr(val)))

    def gen_poisson5(self, nx, ny, row0, row1, base, rowptr, colidx, val):
        check(_lib.lib().cudamat_gen_poisson5(self.h, nx, ny, row0, row1, base, _ptr(rowptr),
                                              _ptr(colidx), _ptr(val)))

    def gen_xstar(self, i0, i1, seed, x):
        check(_lib.lib().cudamat_gen_xstar(self.h, i0, i1, seed, _ptr(x)))

    def timer(self):
        return Timer(self)


class Timer:
    """HIP events on the context's stream"""

    def __init__(self, ctx):
        self.ctx = ctx
        h = C.c_void_p()
        check(_lib.lib().cudamat_timer_create(ctx.h, C.byref(h)))
        self.h = h

    def start(self):
        check(_lib.lib().cudamat_timer_start(self.ctx.h, self.h))

    def stop(self):
        check(_lib.lib().cudamat_timer_stop(self.ctx.h, self.h))

    def elapsed_ms(self):
        ms = C.c_double()
        check(_lib.lib().cudamat_timer_elapsed_ms(self.ctx.h, self.h, C.byref(ms)))
        return ms.value

    def close(self):
        if self.h:
            _lib.lib().cudamat_timer_destroy(self.ctx.h, self.h)
            self.h = None


class Solver:
    """one HBM-resident (row block of a) linear system: cudamat_solver_* of the C ABI"""

    def __init__(self, ctx, n_local, n_cols, nnz, rowptr, colidx, val, base):
        self.ctx = ctx
        self.n, self.n_cols, self.nnz = int(n_local), int(n_cols), int(nnz)
        h = C.c_void_p()
        check(_lib.lib().cudamat_solver_create(ctx.h, self.n, self.n_cols, self.nnz, _ptr(rowptr),
                                               _ptr(colidx), _ptr(val), base, C.byref(h)))
        self.h = h
        self._keep = []

    @classmethod
    def from_host_csr(cls, ctx, rowptr, colidx, val, n_cols=None):
        """cudamat_solver_create_host: the solver's set-up runs beside the upload of the host arrays"""
        rowptr, colidx, val = _np(rowptr, np.int32), _np(colidx, np.int32), _np(val, np.float64)
        n = len(rowptr) - 1
        base = int(rowptr[0])
        nnz = int(rowptr[-1]) - base
        if nnz == 0:
            colidx, val = np.zeros(1, np.int32), np.zeros(1)
        s = cls.__new__(cls)
        s.ctx = ctx
        s.n, s.n_cols, s.nnz = n, int(n if n_cols is None else n_cols), nnz
        h = C.c_void_p()
        check(_lib.lib().cudamat_solver_create_host(ctx.h, s.n, s.n_cols, s.nnz, _vp(rowptr), _vp(colidx), _vp(val), base, C.byref(h)))
        s.h = h
        s._keep = []
        return s

    def close(self):
        if self.h:
            _lib.lib().cudamat_solver_destroy(self.h)
            self.h = None

    def set_shift(self, d):
        self._keep.append(d)
        check(_lib.lib().cudamat_solver_set_shift(self.h, _ptr(d)))

    def set_comm(self, comm):
        """comm: a _lib.Comm (kept alive here) or None"""
        self._keep.append(comm)
        check(_lib.lib().cudamat_solver_set_comm(self.h, None if comm is None else C.byref(comm)))

    def ilu0(self):
        check(_lib.lib().cudamat_solver_ilu0(self.h))

    def trsv_form(self):
        """1: dependency-driven triangular solves, 0: one launch per level"""
        f = C.c_int()
        check(_lib.lib().cudamat_solver_trsv_form(self.h, C.byref(f)))
        return f.value

    def block_ilu0(self):
        """ILU(0) of this rank's diagonal block (block-Jacobi; the only preconditioner of a sharded solver)"""
        check(_lib.lib().cudamat_solver_block_ilu0(self.h))

    def ilu0_values(self):
        cnt = C.c_int64()
        check(_lib.lib().cudamat_solver_ilu0_nnz(self.h, C.byref(cnt)))
        out = self.ctx.empty(max(cnt.value, 1))
        check(_lib.lib().cudamat_solver_ilu0_values(self.h, out.ptr))
        v = out.download()[:cnt.value]
        out.free()
        return v

    def precond_apply(self, vin, vout):
        check(_lib.lib().cudamat_solver_precond_apply(self.h, _ptr(vin), _ptr(vout)))

    def spmv(self, x, y):
        check(_lib.lib().cudamat_solver_spmv(self.h, _ptr(x), _ptr(y)))

    def spmv_mode(self):
        """0: lanes-per-row CSR kernel, 1: blocked two-phase kernels (chosen by the analysis)"""
        m = C.c_int()
        check(_lib.lib().cudamat_solver_spmv_mode(self.h, C.byref(m)))
        return m.value

    def spmv_kernel(self):
        """name(s) of the kernel(s) one SpMV launch runs, as a kernel trace shows them"""
        buf = C.create_string_buffer(96)
        check(_lib.lib().cudamat_solver_spmv_kernel(self.h, buf, 96))
        return buf.value.decode()

    def placement(self):
        """where the blocked copy's arrays went: {"placed": 1 / 0 / -1 (not tried), "slabs", "seconds", "classes"} (include/cudamat.h)"""
        placed, slabs, sec = C.c_int(), C.c_int(), C.c_double()
        buf = C.create_string_buffer(256)
        check(_lib.lib().cudamat_solver_placement(self.h, C.byref(placed), C.byref(slabs), C.byref(sec), buf, 256))
        return {"placed": placed.value, "slabs": slabs.value, "seconds": sec.value, "blocks_by_class": buf.value.decode().strip()}

    def value_dict(self):
        """distinct values when the selected SpMV form reads 8-bit indices into a value dictionary, else 0"""
        m = C.c_int()
        check(_lib.lib().cudamat_solver_value_dict(self.h, C.byref(m)))
        return m.value

    def solve(self, b, x, precond=PRECOND_NONE, loop=LOOP_PBICGSTAB, maxit=2000, tol=1e-8, flags=0):
        st = Stats()
        check(_lib.lib().cudamat_solver_solve(self.h, _ptr(b), _ptr(x), precond, loop, maxit, tol, flags,
                                              C.byref(st)))
        return st

    def history(self, cap=1 << 16):
        buf = np.empty(cap)
        cnt = C.c_int()
        check(_lib.lib().cudamat_solver_history(self.h, _vp(buf), cap, C.byref(cnt)))
        return buf[:cnt.value].copy()
