"""ctypes binding of liboracle.so -- the CPU checker.

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; the product package (cuda_mat_amd) never imports it.
See oracle/oracle.h for what each function restates (reference file:line).
"""
import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
REF_DIR = os.path.join(_HERE, "_ref")

f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


def build(force=False):
    """(re)build liboracle.so and, where /root/reference exists, oracle/_ref."""
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    stale = force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if stale:
        subprocess.run(["make", "-C", _HERE, "liboracle.so"], check=True, capture_output=True)
    subprocess.run(["make", "-C", _HERE, "ref"], check=True, capture_output=True)


class Stats(C.Structure):
    _fields_ = [("iters", C.c_int), ("half_exit", C.c_int), ("converged", C.c_int),
                ("breakdown", C.c_int), ("nrm0", C.c_double), ("nrm", C.c_double)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.orc_spmv.argtypes = [C.c_int, i32p, i32p, f64p, f64p, f64p]
        L.orc_csrmv.argtypes = [C.c_int, i32p, i32p, f64p, C.c_double, f64p, C.c_double, f64p]
        L.orc_dot.argtypes = [C.c_int, f64p, f64p]
        L.orc_dot.restype = C.c_double
        L.orc_nrm2.argtypes = [C.c_int, f64p]
        L.orc_nrm2.restype = C.c_double
        L.orc_bicg.argtypes = [C.c_int, i32p, i32p, f64p, f64p, f64p, C.c_int, C.c_double,
                               C.c_int, C.c_int, C.POINTER(C.c_int)]
        L.orc_bicg_timed.argtypes = [C.c_int, i32p, i32p, f64p, f64p, f64p, C.c_int, C.c_double,
                                     C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_double),
                                     C.POINTER(C.c_double)]
        L.orc_bicg_teams.argtypes = [C.c_int, i32p, i32p, f64p, f64p, f64p, C.c_int, C.c_double,
                                     C.c_int, C.c_int, C.c_int, C.c_int, i32p, i32p, C.POINTER(C.c_double), f64p]
        L.orc_pbicgstab.argtypes = [C.c_int, i32p, i32p, f64p, C.c_void_p, f64p, f64p, C.c_int,
                                    C.c_double, C.c_void_p, C.c_int, C.POINTER(Stats)]
        L.orc_pbicgstab_ex.argtypes = [C.c_int, i32p, i32p, f64p, C.c_void_p, f64p, f64p, C.c_int,
                                       C.c_double, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(Stats)]
        L.orc_pipelined_bicgstab.argtypes = [C.c_int, i32p, i32p, f64p, f64p, f64p, C.c_int, C.c_double,
                                             C.c_void_p, C.c_int, C.POINTER(Stats)]
        L.orc_ppipelined_bicgstab.argtypes = [C.c_int, i32p, i32p, f64p, C.c_void_p, f64p, f64p, C.c_int, C.c_double,
                                              C.c_int, C.c_void_p, C.c_int, C.POINTER(Stats)]
        L.orc_pbicgstab2.argtypes = [C.c_int, i32p, i32p, f64p, C.c_void_p, f64p, f64p, C.c_int,
                                     C.c_double, f64p, C.c_void_p, C.c_int, C.POINTER(Stats)]
        L.orc_ilu0.argtypes = [C.c_int, i32p, i32p, f64p]
        L.orc_trsv_lower_unit.argtypes = [C.c_int, i32p, i32p, f64p, f64p, f64p]
        L.orc_trsv_upper.argtypes = [C.c_int, i32p, i32p, f64p, f64p, f64p]
        L.orc_levels.argtypes = [C.c_int, i32p, i32p, C.c_int, i32p]
        L.orc_mix64.argtypes = [C.c_uint64]
        L.orc_mix64.restype = C.c_uint64
        L.orc_poisson5_nnz.argtypes = [C.c_int, C.c_int]
        L.orc_poisson5_nnz.restype = C.c_int64
        L.orc_poisson5.argtypes = [C.c_int, C.c_int, C.c_int, i32p, i32p, f64p]
        L.orc_rand_row_nnz.argtypes = [C.c_int64, C.c_int]
        L.orc_rand_rows.argtypes = [C.c_int64, C.c_int, C.c_uint64, C.c_int64, C.c_int64, C.c_int,
                                    i32p, i32p, f64p]
        L.orc_xstar.argtypes = [C.c_int64, C.c_int64, C.c_uint64, f64p]
        L.orc_example_count.argtypes = [C.c_int, C.c_double, C.c_uint]
        L.orc_example_count.restype = C.c_int64
        L.orc_example_system.argtypes = [C.c_int, C.c_double, C.c_double, C.c_uint, i32p, i32p, f64p, f64p]
        L.orc_example_system.restype = C.c_int64
        L.orc_mtx_load.argtypes = [C.c_char_p, C.c_int] + [C.POINTER(C.c_int)] * 3 + [
            C.POINTER(C.POINTER(C.c_double)), C.POINTER(C.POINTER(C.c_int)),
            C.POINTER(C.POINTER(C.c_int))]
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_to_dense_vector.argtypes = [C.c_int, C.c_int, f64p, i32p, f64p]
        L.orc_num_threads.restype = C.c_int
        L.orc_set_num_threads.argtypes = [C.c_int]
        _lib = L
        # A GPU box exposes every host core but grants a CPU share of about 16: an OpenMP team
        # of 128 spinning threads makes the small solves of the test-suite crawl.
        if "OMP_NUM_THREADS" not in os.environ:
            L.orc_set_num_threads(default_threads())
    return _lib


@dataclass
class Csr:
    """host CSR; index base = rowptr[0] as in the reference (pbicgstab.cu:201)."""
    n: int
    rowptr: np.ndarray
    colidx: np.ndarray
    val: np.ndarray
    m: int = None

    @property
    def nnz(self):
        return int(self.rowptr[-1] - self.rowptr[0])

    @property
    def base(self):
        return int(self.rowptr[0])

    def rebased(self, base):
        sh = base - self.base
        return Csr(self.n, (self.rowptr + sh).astype(np.int32), (self.colidx + sh).astype(np.int32),
                   self.val.copy(), self.m)

    def to_scipy(self):
        import scipy.sparse as sp
        b = self.base
        return sp.csr_matrix((self.val, self.colidx - b, self.rowptr - b),
                             shape=(self.n, self.m if self.m is not None else self.n))


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def spmv(A, x):
    y = np.empty(A.n)
    lib().orc_spmv(A.n, A.rowptr, A.colidx, A.val, _f(x), y)
    return y


def csrmv(A, alpha, x, beta, y):
    y = _f(y).copy()
    lib().orc_csrmv(A.n, A.rowptr, A.colidx, A.val, alpha, _f(x), beta, y)
    return y


def dot(a, b):
    return lib().orc_dot(len(a), _f(a), _f(b))


def nrm2(a):
    return lib().orc_nrm2(len(a), _f(a))


def bicg(A, b, maxit=2000, eps=1e-6, int_transpose=False, parallel_vec=False):
    x = np.empty(A.n)
    it = C.c_int(0)
    lib().orc_bicg(A.n, A.rowptr, A.colidx, A.val, _f(b), x, maxit, eps, int(int_transpose),
                   int(parallel_vec), C.byref(it))
    return x, it.value


def bicg_timed(A, b, maxit=2000, eps=1e-6, int_transpose=False, parallel_vec=False, fast_transpose=True):
    """bicg() with the seconds of the transposition and of the iteration loop (bicstab.cpp:146-182) apart; the
    transposition may use every thread (same arrays, see oracle_solvers.c transpose2_par).
    Returns (x, iterations, t_transpose, t_loop)."""
    x = np.empty(A.n)
    it = C.c_int(0)
    tt, tl = C.c_double(0.0), C.c_double(0.0)
    lib().orc_bicg_timed(A.n, A.rowptr, A.colidx, A.val, _f(b), x, maxit, eps, int(int_transpose),
                         int(parallel_vec), C.byref(it), int(fast_transpose), C.byref(tt), C.byref(tl))
    return x, it.value, tt.value, tl.value


def bicg_teams(A, b, teams, maxit=2, eps=0.0, parallel_vec=False):
    """the BiCG program once per OpenMP team size in `teams` after one (threaded) transposition: returns
    (x of the last run, [iterations], t_transpose, [loop seconds]) -- bench.py's CPU baseline"""
    x = np.empty(A.n)
    teams = np.ascontiguousarray(teams, dtype=np.int32)
    its = np.zeros(len(teams), np.int32)
    tl = np.zeros(len(teams))
    tt = C.c_double(0.0)
    lib().orc_bicg_teams(A.n, A.rowptr, A.colidx, A.val, _f(b), x, maxit, eps, 0, int(parallel_vec), 1,
                         len(teams), teams, its, C.byref(tt), tl)
    return x, [int(i) for i in its], tt.value, [float(t) for t in tl]


def ilu0(A):
    vm = A.val.copy()
    err = lib().orc_ilu0(A.n, A.rowptr, A.colidx, vm)
    if err:
        raise ZeroDivisionError("zero/missing pivot in row %d" % (err - 1))
    return vm


def trsv_lower_unit(A, vm, rhs):
    out = np.empty(A.n)
    lib().orc_trsv_lower_unit(A.n, A.rowptr, A.colidx, _f(vm), _f(rhs), out)
    return out


def trsv_upper(A, vm, rhs):
    out = np.empty(A.n)
    lib().orc_trsv_upper(A.n, A.rowptr, A.colidx, _f(vm), _f(rhs), out)
    return out


def levels(A, upper=False):
    lev = np.zeros(A.n, dtype=np.int32)
    nlev = lib().orc_levels(A.n, A.rowptr, A.colidx, int(upper), lev)
    return nlev, lev


def pbicgstab(A, f, x0=None, vm=None, maxit=2000, tol=1e-6, want_hist=False, want_trace=False):
    """pbicgstab.cu:45-154 (x0 defaults to ones, :306-308).
    want_trace: also return the loop's scalars, one row per iteration started:
    (rho, sum|rw_j r_j|, rw.v, sum|rw_j v_j|, alpha, t.r, t.t, omega) -- tests/nondominant.py noise_breakdown() reads it."""
    x = np.ones(A.n) if x0 is None else _f(x0).copy()
    st = Stats()
    hist = np.full(2 * maxit, np.nan) if (want_hist or want_trace) else None
    vm_keep = None if vm is None else _f(vm)
    vmp = None if vm_keep is None else vm_keep.ctypes.data_as(C.c_void_p)
    trace = np.full((maxit, 8), np.nan) if want_trace else None
    lib().orc_pbicgstab_ex(A.n, A.rowptr, A.colidx, A.val, vmp, _f(f), x, maxit, tol,
                           None if hist is None else hist.ctypes.data_as(C.c_void_p),
                           0 if hist is None else len(hist),
                           None if trace is None else trace.ctypes.data_as(C.c_void_p), maxit if want_trace else 0, C.byref(st))
    if want_trace:
        started = min(maxit, st.iters + (1 if st.half_exit else 0))
        return x, st, hist, trace[:started]
    return (x, st, hist) if want_hist else (x, st)


PIPE_RR = 32        # residual replacement period of the pipelined loop (csrc/solver.hip kPipeRR)


def pipelined_bicgstab(A, f, x0=None, maxit=2000, tol=1e-6, want_hist=False, verify=True, vm=None, rr=None):
    """pipelined BiCGStab (Cools & Vanroose 2017, Alg. 4), stopping rules of pbicgstab.cu:116,147; x0 defaults to ones.
    vm: ILU(0) values on A's pattern = right preconditioner applied where pbicgstab.cu:92-98,121-127 apply it (None: M = I);
    rr: residual replacement every rr iterations (0: never; default PIPE_RR, the product's default).
    verify (the product's rule, csrc/solver.hip cudamat_solver_solve): an iterate the loop calls converged is checked
    against its TRUE residual; beyond twice the target the loop is restarted from it towards the same absolute target,
    at most three times within maxit.  st.iters is the total; the history is that of the first segment."""
    x = np.ones(A.n) if x0 is None else _f(x0).copy()
    rr = PIPE_RR if rr is None else rr
    st = Stats()
    hist = np.full(2 * maxit, np.nan) if want_hist else None
    fb = _f(f)
    vm_keep = None if vm is None else _f(vm)
    vmp = None if vm_keep is None else vm_keep.ctypes.data_as(C.c_void_p)
    lib().orc_ppipelined_bicgstab(A.n, A.rowptr, A.colidx, A.val, vmp, fb, x, maxit, tol, int(rr),
                                  None if hist is None else hist.ctypes.data_as(C.c_void_p),
                                  0 if hist is None else len(hist), C.byref(st))
    st.restarts = 0
    if verify and st.converged and st.nrm0 > 0.0:
        target = tol * st.nrm0
        for _ in range(3):
            if not st.converged or st.iters >= maxit:
                break
            true = float(np.linalg.norm(fb - spmv(A, x)))
            if true <= 2.0 * target:
                st.nrm = true
                break
            st2 = Stats()
            lib().orc_ppipelined_bicgstab(A.n, A.rowptr, A.colidx, A.val, vmp, fb, x, maxit - st.iters, target / true, int(rr),
                                          None, 0, C.byref(st2))
            st.restarts += 1
            st.iters += st2.iters
            st.converged, st.half_exit, st.nrm = st2.converged, st2.half_exit, st2.nrm
    return (x, st, hist) if want_hist else (x, st)


def pbicgstab2(A0, b, d=None, x0=None, maxit=2000, tol=1e-6, want_hist=False):
    """pbicgstab.cu:581-754 (d=None: intended maths of :425-578, x0 = 1 :827-831)."""
    x0 = np.ones(A0.n) if x0 is None else _f(x0)
    x = np.zeros(A0.n)
    st = Stats()
    hist = np.full(maxit, np.nan) if want_hist else None
    d_keep = None if d is None else _f(d)
    dp = None if d_keep is None else d_keep.ctypes.data_as(C.c_void_p)
    ok = lib().orc_pbicgstab2(A0.n, A0.rowptr, A0.colidx, A0.val, dp, x0, _f(b), maxit, tol, x,
                              None if hist is None else hist.ctypes.data_as(C.c_void_p),
                              0 if hist is None else len(hist), C.byref(st))
    return (bool(ok), x, st, hist) if want_hist else (bool(ok), x, st)


def poisson5(nx, ny, base=0):
    n = nx * ny
    nnz = lib().orc_poisson5_nnz(nx, ny)
    rp = np.empty(n + 1, np.int32)
    ci = np.empty(nnz, np.int32)
    v = np.empty(nnz)
    lib().orc_poisson5(nx, ny, base, rp, ci, v)
    return Csr(n, rp, ci, v, n)


def rand_rows(n, per_row, seed, row0=0, row1=None, base=0):
    """rows [row0,row1) of the synthetic random matrix (global column ids)."""
    row1 = n if row1 is None else row1
    rn = lib().orc_rand_row_nnz(n, per_row)
    nr = row1 - row0
    rp = np.empty(nr + 1, np.int32)
    ci = np.empty(nr * rn, np.int32)
    v = np.empty(nr * rn)
    lib().orc_rand_rows(n, per_row, seed, row0, row1, base, rp, ci, v)
    return Csr(nr, rp, ci, v, n)


def xstar(n, seed, i0=0, i1=None):
    i1 = n if i1 is None else i1
    x = np.empty(i1 - i0)
    lib().orc_xstar(i0, i1, seed, x)
    return x


def example_system(dim=10000, p_zero=0.99, p_zero_vec=0.2, seed=1):
    """the system the reference CLI solves when no -M is given (example.cpp:274-288,339): returns (A 1-based, b).
    seed 1 = the reference's own rand() stream (it never calls srand)."""
    nnz = lib().orc_example_count(dim, p_zero, seed)
    rp = np.empty(dim + 1, np.int32)
    ci = np.empty(max(nnz, 1), np.int32)
    v = np.empty(max(nnz, 1))
    b = np.empty(dim)
    got = lib().orc_example_system(dim, p_zero, p_zero_vec, seed, rp, ci, v, b)
    assert got == nnz
    return Csr(dim, rp, ci[:nnz].copy(), v[:nnz].copy(), dim), b


def _take(ptr, count, dtype):
    a = np.ctypeslib.as_array(ptr, shape=(max(count, 0),)).astype(dtype, copy=True)
    return a


def mtx_load(path, csr=True):
    m, n, nnz = C.c_int(), C.c_int(), C.c_int()
    v = C.POINTER(C.c_double)()
    r = C.POINTER(C.c_int)()
    c = C.POINTER(C.c_int)()
    err = lib().orc_mtx_load(path.encode(), int(csr), C.byref(m), C.byref(n), C.byref(nnz),
                             C.byref(v), C.byref(r), C.byref(c))
    if err:
        raise IOError("orc_mtx_load failed for %s" % path)
    nr = (m.value if csr else nnz.value)
    ncol = (nnz.value if csr else n.value)
    row = _take(r, nr + (1 if csr else 0), np.int32)
    col = _take(c, ncol + (0 if csr else 1), np.int32)
    val = _take(v, nnz.value, np.float64)
    for p in (v, r, c):
        lib().orc_free(p)
    if csr:
        return Csr(m.value, row, col, val, n.value)
    return m.value, n.value, row, col, val


def to_dense_vector(A):
    """pbicgstab.cu:1101-1115 on an n x 1 CSR."""
    out = np.empty(A.n)
    lib().orc_to_dense_vector(A.n, A.nnz, A.val, A.rowptr, out)
    return out


def default_threads():
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    return max(1, min(16, avail))


def num_threads():
    return lib().orc_num_threads()


def set_num_threads(n):
    lib().orc_set_num_threads(int(n))


# ---- the unmodified reference (oracle/_ref), only where it was built --------
def ref_available():
    return os.path.exists(os.path.join(REF_DIR, "bicstab_ref"))


def write_ref_text_format(A, b, mat_path, vec_path):
    """bicstab.cpp:198-227 input format: 'NZ N', NZ x 'value col', N+1 row ptrs; 0-based."""
    A0 = A.rebased(0)
    with open(mat_path, "w") as f:
        f.write("%d %d\n" % (A0.nnz, A0.n))
        for v, c in zip(A0.val, A0.colidx):
            f.write("%.17g %d\n" % (v, c))
        f.write("\n".join(str(int(r)) for r in A0.rowptr) + "\n")
    with open(vec_path, "w") as f:
        f.write("%d\n" % len(b))
        f.write("\n".join("%.17g" % t for t in b) + "\n")


def run_ref_bicg(A, b, workdir, threads=4):
    """run the unmodified reference program; returns (x at its 6-digit print
    precision, iteration count)."""
    mat, vec = os.path.join(workdir, "mat.txt"), os.path.join(workdir, "vec.txt")
    write_ref_text_format(A, b, mat, vec)
    env = dict(os.environ, OMP_NUM_THREADS=str(threads))
    out = subprocess.run([os.path.join(REF_DIR, "bicstab_ref")], input="%s\n%s\n" % (mat, vec),
                         capture_output=True, text=True, env=env, check=True).stdout
    lines = out.split("\n")
    iters = int(lines[2].split(":")[1])
    x = np.array([float(t) for t in lines[4].split()])
    return x, iters
