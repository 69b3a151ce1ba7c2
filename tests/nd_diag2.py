"""GPU box, by hand: the reference loop (pbicgstab.cu:45-154, M = I) three ways on one stagnating system -- the oracle, the
library's fused loop, and the SAME call sequence as the reference issued one primitive at a time through the C ABI
(cudamat_spmv / dot / axpy / scal on device pointers, scalars on the host) -- with the loop's scalars side by side."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cuda_mat_amd as cm
from oracle import oracle as O
from tests import nondominant as ND
O.set_num_threads(1)
ctx = cm.Context(0)
name = sys.argv[1] if len(sys.argv) > 1 else "example2000"
A, b = ND.FAMILY[name](O)
n = A.n
MAXIT = 400


def loop(spmv, dot, axpy, scal, copy, ones):
    """the reference's call sequence; returns per-iteration rows (rho, rw.v, alpha, nrm_half, t.r, t.t, omega, nrm_full, beta)"""
    x = ones(); r = spmv(x); scal(-1.0, r); axpy(1.0, b_dev(), r); rw = copy(r); p = copy(r)
    nrm0 = np.sqrt(dot(r, r)); rows = []; rho = 0.0; alpha = omega = 0.0; beta = np.nan
    for i in range(MAXIT):
        rhop = rho; rho = dot(rw, r)
        if i > 0:
            beta = (rho / rhop) * (alpha / omega)
            axpy(-omega, v, p); scal(beta, p); axpy(1.0, r, p)
        v = spmv(p); rwv = dot(rw, v); alpha = rho / rwv
        axpy(-alpha, v, r); axpy(alpha, p, x); nh = np.sqrt(dot(r, r))
        t = spmv(r); tr = dot(t, r); tt = dot(t, t); omega = tr / tt
        axpy(omega, r, x); axpy(-omega, t, r); nf = np.sqrt(dot(r, r))
        rows.append((rho, rwv, alpha, nh, tr, tt, omega, nf, beta))
        if not np.isfinite(nf): break
    return nrm0, rows


# --- host (oracle primitives)
_b = b.copy()
def b_dev(): return _b
def h_axpy(a, x, y): y += a * x          # one rounding per product and per sum, like orc_axpy
def h_scal(a, x): x *= a
nrm0_o, rows_o = loop(lambda x: O.spmv(A, x), lambda a, c: O.dot(a, c), h_axpy, h_scal, lambda a: a.copy(), lambda: np.ones(n))

# --- GPU primitives through the C ABI
drp, dci, dva = ctx.array(A.rowptr), ctx.array(A.colidx), ctx.array(A.val)
db = ctx.array(b)
def b_dev(): return db
def g_spmv(x):
    y = ctx.empty(n); ctx.spmv(n, drp, dci, dva, A.base, x, y); return y
def g_copy(a):
    c = ctx.empty(n); c.upload(a.download()); return c
nrm0_g, rows_g = loop(g_spmv, lambda a, c: ctx.dot(n, a, c), lambda a, x, y: ctx.axpy(n, a, x, y), lambda a, x: ctx.scal(n, a, x), g_copy, lambda: ctx.array(np.ones(n)))

# --- the library's fused loop
s = cm.Solver.from_host_csr(ctx, A.rowptr, A.colidx, A.val)
dx = ctx.array(np.ones(n))
st = s.solve(db, dx, precond=0, loop=0, maxit=MAXIT, tol=1e-30)
hg = s.history()
print("%s: oracle-primitive loop ends after %d iterations, GPU-primitive loop after %d, fused GPU loop after %d (breakdown %d)" % (name, len(rows_o), len(rows_g), st.iters, st.breakdown))
print("nrm0", nrm0_o, nrm0_g, st.nrm0)
hdr = "it  who      rho        rw.v       alpha      |r|half    t.r        t.t        omega      |r|full    beta"
print(hdr)
last = max(len(rows_g), st.iters)
for i in list(range(0, 4)) + list(range(max(4, last - 14), last + 1)):
    for who, rows in (("orc", rows_o), ("gpuP", rows_g)):
        if i < len(rows): print("%3d %-5s " % (i, who) + " ".join("%10.3e" % v for v in rows[i]))
    if 2 * i + 1 < len(hg): print("%3d fused %32s %10.3e %32s %10.3e" % (i, "", hg[2 * i], "", hg[2 * i + 1]))
