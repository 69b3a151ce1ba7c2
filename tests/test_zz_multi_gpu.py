"""Paths that need TWO OR MORE real GPUs (skipped on a one-GPU box).  They are the first executions of the
inter-GPU exchanges (grouped ncclSend/ncclRecv pieces, RCCL collectives between devices) wherever the suite meets such a
box, so they live in a file that sorts last: a failure here cannot hide a result of the single-GPU parity tests."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "cuda_mat_amd", "host")
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def cm():
    import cuda_mat_amd as cm
    if cm.device_count() < 2:
        pytest.skip("needs two GPUs")
    return cm


def test_bench_two_gpus_over_rccl(cm):
    """two real GPUs (skipped on a one-GPU box): bench.py under torch.distributed.run with the library's RCCL
    binding, the gather in pieces behind phase 1, and the same run with the plain all-gather -- both must pass
    bench.py's own gate at the first form they are given and produce the same solution bit for bit"""
    import json
    import socket
    import subprocess
    root = ROOT
    outs = {}
    for form in ("rccl:1", "rccl:0"):
        sk = socket.socket()
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
        sk.close()
        env = dict(os.environ, CUDAMAT_BENCH_FORMS=form, CUDAMAT_SPMV_MODE="pb")
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"),
                            "--gpus", "2", "--rows", "2000000", "--steps", "10", "--warmup", "2", "--cpu-baseline", "off"],
                           capture_output=True, text=True, timeout=300, env=env, cwd=root)
        assert r.returncode == 0, r.stderr[-3000:]
        out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
        assert out["n_gpus"] == 2 and out["comm"]["gate"] == [{"form": {"comm": "rccl", "overlap": form.endswith("1")},
                                                                 "failed_ranks": 0, "rank0_failure": None}]
        outs[form] = out
    assert outs["rccl:1"]["comm"]["form"]["gather"].startswith("in pieces")
    assert outs["rccl:1"]["config"]["gate_x_sha256"] == outs["rccl:0"]["config"]["gate_x_sha256"]


def test_cli_row_shards_over_two_gpus(cm):
    """example -G2 with one GPU per rank: the in-process RCCL communicators of cudamat_solve_sharded"""
    subprocess.run(["make", "-C", HOST], check=True, capture_output=True)
    built = os.path.join(HOST, "example")
    mat = "-M" + os.path.join(GOLD, "mat10000.mtx")
    ref = subprocess.run([built, mat, "-C0", "-T1e-8", "-P"], capture_output=True, text=True, timeout=300)
    r = subprocess.run([built, mat, "-C0", "-T1e-8", "-P", "-G2"], capture_output=True, text=True, timeout=300)
    assert ref.returncode == 0 and r.returncode == 0 and "Using 2 GPUs" in r.stdout, r.stdout[-500:] + r.stderr[-2000:]
    x0 = [float(v) for v in re.search(r"result:\s*\(([^)]*)\)", ref.stdout).group(1).split()]
    x1 = [float(v) for v in re.search(r"result:\s*\(([^)]*)\)", r.stdout).group(1).split()]
    assert len(x0) == len(x1) == 10000 and max(abs(a - b) for a, b in zip(x0, x1)) <= 2e-5


def test_a_failing_rank_aborts_the_rccl_communicators(cm):
    """two real GPUs, in-process RCCL ranks (cudamat_solve_sharded): rank 1's 7th all-reduce fails (injected) while rank 0
    sits in the matching collective -- ncclCommAbort on every rank's communicators must bring the call back with an error"""
    import threading
    import numpy as np
    from oracle import oracle as O
    A = O.mtx_load(os.path.join(GOLD, "mat10000.mtx"))
    b = O.spmv(A, 1.0 + np.sin(np.arange(A.n)))
    os.environ["CUDAMAT_TEST_COMM_FAIL"] = "1:7"
    box = {}

    def call():
        try:
            cm.use_gpus(2)
            box["res"] = cm.bicgstab(A.n, A.nnz, A.val, A.rowptr, A.colidx, b, 2000, 1e-8)
        except Exception as e:  # noqa: BLE001
            box["err"] = e
        finally:
            cm.use_gpus(1)

    try:
        t = threading.Thread(target=call, daemon=True)
        t.start()
        t.join(180)
    finally:
        del os.environ["CUDAMAT_TEST_COMM_FAIL"]
    assert not t.is_alive(), "cudamat_solve_sharded did not return after rank 1 failed"
    assert "err" in box and "rank 1 of 2" in str(box["err"]), box
