"""The C++ drop-in library (libcuda_mat.so: pbicgstab.h / mmio_wrapper.h entry points) and the
`example` CLI (reference example.cpp:168-378)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "cuda_mat_amd", "host")
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def built():
    import cuda_mat_amd
    cuda_mat_amd.lib()
    subprocess.run(["make", "-C", HOST], check=True, capture_output=True)
    return os.path.join(HOST, "example")


def test_host_library_exports_the_reference_entry_points(built):
    """same C++ signatures as pbicgstab.h:113,116,119 / mmio_wrapper.h:133 (mangled names match)"""
    syms = subprocess.run(["nm", "-DC", os.path.join(HOST, "libcuda_mat.so")], capture_output=True,
                          text=True, check=True).stdout
    for want in [
        "bicgstab(int, int, double*, int*, int*, double*, int, double, bool, double*, double*)",
        "bicgstab(int, int, double*, int*, int*, double*, double*, double*, int, double, bool, double*, double*)",
        "bicgstab_lu_precond(int, int, double*, int*, int*, double*, int, double, bool, double*, double*)",
        "loadMMSparseMatrix(char*, char, bool, int*, int*, int*, double**, int**, int**)",
        "toDenseVector(int, int, double*, int*, double*)",
        "gen_rand_vector(int, double*, double, double, double)",
        "rand_float(double, double)", "rand_float_0_1()", "second()",
    ]:
        assert " T " + want in syms, want


def test_cli_rejects_unknown_switch_and_missing_file(built):
    r = subprocess.run([built, "-Z"], capture_output=True, text=True)
    assert r.returncode != 0 and "Unknown switch '-Z'" in r.stderr
    import cuda_mat_amd as cm
    if cm.device_count() > 0:
        r = subprocess.run([built, "-M/nonexistent.mtx"], capture_output=True, text=True)
        assert r.returncode != 0 and "FAILED" in r.stderr


@pytest.mark.gpu
def test_cli_solves_the_shipped_fixtures(built):
    # the reference's own usage lines (example.cpp:188-190)
    r = subprocess.run([built, "-M" + os.path.join(GOLD, "mat10000.mtx")], capture_output=True, text=True)
    assert r.returncode == 0 and "nnz=49600" in r.stdout and "success" in r.stdout
    assert "algorithm delta time = " in r.stdout and "total delta time = " in r.stdout
    r = subprocess.run([built, "-M" + os.path.join(GOLD, "mat3.mtx"), "-V" + os.path.join(GOLD, "vec3.mtx"), "-D", "-P",
                        "-C1", "-T1e-9"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "initial norm = " in r.stdout and "k = 0, norm = " in r.stdout
    assert "(1.166667 5.666667 -3.833333 )" in r.stdout
    # mat3 has no (2,2) entry: the ILU(0) path must fail loudly, not crash (pbicgstab.h:118)
    r = subprocess.run([built, "-M" + os.path.join(GOLD, "mat3.mtx"), "-V" + os.path.join(GOLD, "vec3.mtx")],
                       capture_output=True, text=True)
    assert r.returncode != 0 and "no diagonal entry" in r.stderr and "method failed" in r.stderr
    # random matrix path (-N -R), ILU(0) with the debug trace of pbicgstab.cu:77,114,145,204
    # (such a matrix, entries in [1,10] and no dominance, need not converge: only the trace is checked)
    r = subprocess.run([built, "-N40", "-R0.5", "-D", "-S1", "-I50"], capture_output=True, text=True)
    assert r.returncode in (0, 1), r.stdout + r.stderr
    assert "N=40, nnz=" in r.stdout and "gpu, init residual:norm" in r.stdout
    assert "residual norm (before precond)" in r.stdout
    assert ("success" in r.stdout) == (r.returncode == 0)
    r = subprocess.run([built, "-M" + os.path.join(GOLD, "mat900.mtx"), "-C0", "-T1e-8"], capture_output=True, text=True)
    assert r.returncode == 0 and "iterations = " in r.stdout
