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
def test_cli_draws_the_reference_random_systems(built):
    """no -M: the CLI builds the reference's random system (example.cpp:274-288) from libc rand() -- unseeded, i.e. the
    srand(1) stream, as upstream -- and prints `nnz=` before it solves: the counts are those of the oracle's restatement of
    the recipe (oracle.example_system), which the GPU parity tests solve.  (Like upstream, the CLI looks for its device
    first: without one it ends before the matrix is drawn.)"""
    from oracle import oracle as O
    for args, dim, pz in ((["-N40", "-R0.5"], 40, 0.5), ([], 10000, 0.99)):
        r = subprocess.run([built] + args + ["-I3"], capture_output=True, text=True)
        A, b = O.example_system(dim, pz, 0.2, 1)
        assert "nnz=%d\n" % A.nnz in r.stdout, r.stdout[-300:]
    assert A.nnz == 1007629


@pytest.mark.gpu
def test_cli_default_workload_ends_as_the_reference_loop_would(built):
    """`example` with no arguments (example.cpp:173-180: n = 10000, P(0) = 0.99, ILU(0), maxit 2000, tol 1e-6): the factors
    blow the first direction up and the residual is NaN from the second iteration on (tests/test_gpu_nondominant.py has the
    oracle comparison).  Upstream spins to maxit on NaNs and prints "success" whatever happened (pbicgstab.cu:408); here
    the run stops at the NaN and says "method failed" with a non-zero exit status."""
    r = subprocess.run([built, "-D"], capture_output=True, text=True)
    assert "nnz=1007629" in r.stdout and "N=10000, nnz=1007629" in r.stdout
    assert r.returncode != 0 and "method failed" in r.stderr and "success" not in r.stdout
    assert "i = 0, residual norm (before precond)" in r.stdout and "i = 2," not in r.stdout


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


@pytest.mark.gpu
def test_cli_row_shards_over_gpus(built):
    """-G<n>: one host thread per rank, uniform row blocks, the sharded C++ loop (csrc/sharded.cpp).  A one-GPU box
    runs the ranks on device 0 with host-synchronised copies in place of RCCL (CUDAMAT_SHARDED_ONE_DEVICE=1); with
    enough devices the same command line uses one GPU per rank over RCCL."""
    import re
    import cuda_mat_amd as cm
    mat = "-M" + os.path.join(GOLD, "mat10000.mtx")
    ref = subprocess.run([built, mat, "-C0", "-T1e-8", "-P", "-S3"], capture_output=True, text=True)
    assert ref.returncode == 0, ref.stderr
    runs = [(dict(os.environ, CUDAMAT_SHARDED_ONE_DEVICE="1"), n) for n in (2, 3)]
    # (the real thing, RCCL between two GPUs, runs last of all: tests/test_zz_multi_gpu.py)
    for env, n in runs:
        r = subprocess.run([built, mat, "-C0", "-T1e-8", "-P", "-S3", "-G%d" % n], capture_output=True, text=True, env=env)
        assert r.returncode == 0 and "success" in r.stdout and "Using %d GPUs" % n in r.stdout, r.stdout[-500:] + r.stderr[-2000:]
        it0 = int(re.search(r"iterations = (\d+)", ref.stdout).group(1))
        it1 = int(re.search(r"iterations = (\d+)", r.stdout).group(1))
        assert abs(it0 - it1) <= max(2, it0 // 10)
        x0 = [float(v) for v in re.search(r"result:\s*\(([^)]*)\)", ref.stdout).group(1).split()]
        x1 = [float(v) for v in re.search(r"result:\s*\(([^)]*)\)", r.stdout).group(1).split()]
        assert len(x0) == len(x1) == 10000 and max(abs(a - b) for a, b in zip(x0, x1)) <= 2e-5
    # the preconditioned entry point shards as block-Jacobi and says so
    env = dict(os.environ, CUDAMAT_SHARDED_ONE_DEVICE="1")
    r = subprocess.run([built, mat, "-T1e-8", "-G2"], capture_output=True, text=True, env=env)
    assert r.returncode == 0 and "block-Jacobi" in r.stderr and "success" in r.stdout
    # more ranks than devices without the emulation switch is refused up front
    if cm.device_count() < 7:
        r = subprocess.run([built, mat, "-G7"], capture_output=True, text=True)
        assert r.returncode != 0 and "only" in r.stderr


def test_host_selftest_under_address_sanitizer():
    """SURVEY section 5 (sanitizers for the host side): the loader, toDenseVector and Matrix.h built with
    -fsanitize=address,undefined and run on the shipped fixtures and on hostile files (no GPU involved)"""
    r = subprocess.run(["make", "-C", HOST, "asan-check"], capture_output=True, text=True)
    assert r.returncode == 0 and "SELFTEST_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr
