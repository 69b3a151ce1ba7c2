"""Property test of the Matrix Market loader: random valid coordinate files -- general /
symmetric / skew-symmetric, real / integer, 0- or 1-based, comments and blank lines, CSR and CSC --
through the product loader (cudamat_load_mtx), the oracle restatement and, where oracle/_ref was built
from /root/reference in this container, the UNMODIFIED reference loader itself.  CPU only."""
import ctypes as C
import os

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libref_mmio.so")


def _ref_load(path, csr):
    L = C.CDLL(REF_SO)
    L.ref_loadMMSparseMatrix.argtypes = [C.c_char_p, C.c_int] + [C.POINTER(C.c_int)] * 3 + [
        C.POINTER(C.POINTER(C.c_double)), C.POINTER(C.POINTER(C.c_int)), C.POINTER(C.POINTER(C.c_int))]
    m, n, nnz = C.c_int(), C.c_int(), C.c_int()
    v, r, c = C.POINTER(C.c_double)(), C.POINTER(C.c_int)(), C.POINTER(C.c_int)()
    err = L.ref_loadMMSparseMatrix(path.encode(), int(csr), C.byref(m), C.byref(n), C.byref(nnz), C.byref(v),
                                   C.byref(r), C.byref(c))
    if err:
        return None
    nr = m.value + 1 if csr else nnz.value
    nc = nnz.value if csr else n.value + 1
    return (m.value, n.value, nnz.value, np.ctypeslib.as_array(v, (nnz.value,)).copy() if nnz.value else np.zeros(0),
            np.ctypeslib.as_array(r, (nr,)).copy() if nr else np.zeros(0, np.int32),
            np.ctypeslib.as_array(c, (nc,)).copy() if nc else np.zeros(0, np.int32))


@st.composite
def mtx_files(draw):
    sym = draw(st.sampled_from(["general", "symmetric", "skew-symmetric"]))
    field = draw(st.sampled_from(["real", "integer"]))
    m = draw(st.integers(1, 12))
    n = m if sym != "general" else draw(st.integers(1, 12))
    base = draw(st.integers(0, 1))
    cells = [(i, j) for i in range(m) for j in range(n)]
    if sym == "symmetric":
        cells = [(i, j) for i, j in cells if i >= j]
    elif sym == "skew-symmetric":
        cells = [(i, j) for i, j in cells if i > j]
    picked = draw(st.lists(st.sampled_from(cells), unique=True, min_size=0 if cells else 0,
                           max_size=min(len(cells), 40))) if cells else []
    # the reference detects the base from the data: make the detection unambiguous and equal to `base`
    if base == 1 and picked and not any(i == m - 1 or j == n - 1 for i, j in picked):
        picked.append((m - 1, 0) if (m - 1, 0) in cells and (m - 1, 0) not in picked else picked[0])
    if base == 0 and picked and not any(i == 0 or j == 0 for i, j in picked):
        return draw(mtx_files())            # ambiguous (would be read as base 0 with shifted indices anyway)
    picked = list(dict.fromkeys(picked))
    if base == 1 and picked and not any(i == m - 1 or j == n - 1 for i, j in picked):
        return draw(mtx_files())
    order = draw(st.permutations(picked)) if picked else []
    vals = [draw(st.integers(-9, 9)) if field == "integer" else draw(st.floats(-1e3, 1e3, allow_nan=False,
            allow_infinity=False, width=32)) for _ in order]
    lines = ["%%MatrixMarket matrix coordinate " + field + " " + sym]
    if draw(st.booleans()):
        lines.append("% a comment")
    if draw(st.booleans()):
        lines.append("")
    lines.append("%d %d %d" % (m, n, len(order)))
    for (i, j), v in zip(order, vals):
        lines.append("%d %d %s" % (i + base, j + base, repr(v) if field == "real" else str(v)))
    return "\n".join(lines) + "\n"


@settings(max_examples=120, deadline=None, suppress_health_check=[HealthCheck.too_slow, HealthCheck.function_scoped_fixture])
@given(text=mtx_files(), csr=st.booleans())
def test_loader_agrees_with_oracle_and_reference(text, csr, oracle, tmp_path_factory):
    import cuda_mat_amd as cm
    path = str(tmp_path_factory.mktemp("mtx") / "m.mtx")
    with open(path, "w") as f:
        f.write(text)
    got = cm.loadMMSparseMatrix(path, "d", csr)
    ref = _ref_load(path, csr) if os.path.exists(REF_SO) else None
    try:
        if csr:
            A = oracle.mtx_load(path, csr=True)
            orc = (A.n, A.m, A.nnz, A.val, A.rowptr, A.colidx)
        else:
            m, n, row, col, val = oracle.mtx_load(path, csr=False)
            orc = (m, n, len(val), val, row, col)
    except IOError:
        orc = None
    if orc is None:
        assert got[0] == 1 and ref is None
        return
    assert got[0] == 0, text
    assert (got[1], got[2], got[3]) == orc[:3]
    np.testing.assert_array_equal(got[4], orc[3])
    np.testing.assert_array_equal(got[5], orc[4])
    np.testing.assert_array_equal(got[6], orc[5])
    if ref is not None:
        assert ref[:3] == orc[:3]
        np.testing.assert_array_equal(ref[3], orc[3])
        np.testing.assert_array_equal(ref[4], orc[4])
        np.testing.assert_array_equal(ref[5], orc[5])
