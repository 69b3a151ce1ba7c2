#!/usr/bin/env python3
"""Regenerate tests/golden/*.npz from the UNMODIFIED reference, run in place.

Needs /root/reference (this container only): builds oracle/_ref via
oracle/Makefile (bicstab_omp/bicstab.cpp at -O0, mmio.c + mmio_wrapper.h) and
records what the reference itself produces:

  loader_<name>.npz : m, n, nnz, rowptr, colidx, val returned by the reference
                      loadMMSparseMatrix (mmio_wrapper.h:133) for each shipped
                      .mtx fixture (copied next to this script as data).
  bicg_<case>.npz   : b, the solution printed by the reference CPU program
                      (bicstab.cpp:255-256, 6 significant digits) and its
                      iteration count (bicstab.cpp:250), hard-coded eps 1e-6.

The .npz files are data (inputs + expected outputs); no reference source is
stored.  Run:  python tests/golden/make_golden.py
"""
import ctypes as C
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

FIXTURES = ["mat3", "mat3_A0", "vec3", "vec3_d", "mat900", "mat10000"]
RAND_N, RAND_PER_ROW, RAND_SEED = 20000, 50, 0x5EED


def ref_load(path, csr=True):
    L = C.CDLL(os.path.join(O.REF_DIR, "libref_mmio.so"))
    L.ref_loadMMSparseMatrix.argtypes = [C.c_char_p, C.c_int] + [C.POINTER(C.c_int)] * 3 + [
        C.POINTER(C.POINTER(C.c_double)), C.POINTER(C.POINTER(C.c_int)), C.POINTER(C.POINTER(C.c_int))]
    m, n, nnz = C.c_int(), C.c_int(), C.c_int()
    v, r, c = C.POINTER(C.c_double)(), C.POINTER(C.c_int)(), C.POINTER(C.c_int)()
    err = L.ref_loadMMSparseMatrix(path.encode(), int(csr), C.byref(m), C.byref(n), C.byref(nnz),
                                   C.byref(v), C.byref(r), C.byref(c))
    assert err == 0, path
    nr = m.value + 1 if csr else nnz.value
    nc = nnz.value if csr else n.value + 1
    return dict(m=m.value, n=n.value, nnz=nnz.value,
                rowptr=np.ctypeslib.as_array(r, (nr,)).astype(np.int32),
                colidx=np.ctypeslib.as_array(c, (nc,)).astype(np.int32),
                val=np.ctypeslib.as_array(v, (nnz.value,)).astype(np.float64))


def main():
    O.build()
    assert O.ref_available(), "oracle/_ref not built (needs /root/reference)"
    mats = {}
    for name in FIXTURES:
        g = ref_load(os.path.join(HERE, name + ".mtx"))
        np.savez_compressed(os.path.join(HERE, "loader_%s.npz" % name), **g)
        mats[name] = O.Csr(g["m"], g["rowptr"], g["colidx"], g["val"], g["n"])
        print("loader", name, g["m"], g["n"], g["nnz"], "base", g["rowptr"][0])
    g = ref_load(os.path.join(HERE, "mat900.mtx"), csr=False)
    np.savez_compressed(os.path.join(HERE, "loader_mat900_csc.npz"), **g)

    cases = {}
    A = mats["mat900"]
    cases["mat900_urand"] = (A, np.random.default_rng(0).uniform(1.0, 5.0, A.n))
    cases["mat900_sin"] = (A, O.spmv(A, 1.0 + np.sin(np.arange(A.n))))
    A = mats["mat10000"]
    cases["mat10000_sin"] = (A, O.spmv(A, 1.0 + np.sin(np.arange(A.n))))
    A = O.rand_rows(RAND_N, RAND_PER_ROW, RAND_SEED)
    cases["rand20000x50"] = (A, O.spmv(A, O.xstar(RAND_N, RAND_SEED + 1)))
    with tempfile.TemporaryDirectory() as td:
        for name, (A, b) in cases.items():
            x, iters = O.run_ref_bicg(A, b, td, threads=4)
            np.savez_compressed(os.path.join(HERE, "bicg_%s.npz" % name), b=b, x=x, iters=iters)
            print("bicg", name, "iters", iters)


if __name__ == "__main__":
    main()
