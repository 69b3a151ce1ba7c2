"""Row-sharded path on the GPU box (one GPU): the real C++ sharded loop driven by W emulated ranks
(threads with their own contexts/streams; collectives = host-synchronised copies, tests/dist_sim.py)
and by torch.distributed's NCCL(=RCCL) backend at world size 1 with the sharded path forced."""
import os
import sys
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(autouse=True)
def _serial_oracle(oracle):
    """one OpenMP thread for the checker: its `reduction(+)` dots are then summed in one fixed order, so oracle
    iteration counts (and with them the last digits of its solution) do not wander from run to run"""
    before = oracle.num_threads()
    oracle.set_num_threads(1)
    yield
    oracle.set_num_threads(before)


@pytest.fixture(scope="module")
def cm():
    import cuda_mat_amd as cm
    assert cm.device_count() > 0
    return cm


def _run_rank(cm, group, rank, n, A, b, out, pieces=False, side_reduce=False, windows=False, **kw):
    import dist_sim
    from cuda_mat_amd.dist import shard_rows
    try:
        ctx = cm.Context(0)
        row0, row1, per = shard_rows(n, group.world, rank)
        rp = (A.rowptr[row0:row1 + 1] - A.rowptr[row0]).astype(np.int32)
        k0, k1 = A.rowptr[row0], A.rowptr[row1]
        s = cm.Solver.from_host_csr(ctx, rp, A.colidx[k0:k1], A.val[k0:k1], n_cols=n)
        comm = dist_sim.ThreadComm(cm, group, rank, ctx, pieces=pieces, side_reduce=side_reduce, windows=windows)
        s.set_comm(comm.struct)
        db, dx = ctx.array(b[row0:row1]), ctx.array(np.ones(row1 - row0))
        # y = A x through the sharded SpMV entry point
        dy = ctx.empty(row1 - row0)
        s.spmv(db, dy)
        y = dy.download()
        st = s.solve(db, dx, **kw)
        out[rank] = (row0, row1, dx.download(), st.as_dict(), s.history(), y, comm.n_allgather, comm.n_allreduce,
                     comm.n_parts, comm.n_side, comm.n_windows, comm.window_doubles)
        s.close()
        comm.close()
        ctx.close()
    except Exception as e:  # noqa: BLE001
        group.barrier.abort()
        out[rank] = e


@pytest.mark.parametrize("world,n,per_row", [(2, 20000, 50), (3, 10007, 20), (4, 4096, 8)])
def test_cpp_sharded_loop_with_emulated_ranks(cm, oracle, world, n, per_row):
    import dist_sim
    A = oracle.rand_rows(n, per_row, 0x5EED)
    xs = oracle.xstar(n, 0x5EEE)
    b = oracle.spmv(A, xs)
    group = dist_sim.ThreadGroup(world)
    out = [None] * world
    th = [threading.Thread(target=_run_rank, args=(cm, group, r, n, A, b, out),
                           kwargs=dict(loop=cm.LOOP_PBICGSTAB, maxit=200, tol=1e-8)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
    for o in out:
        assert not isinstance(o, Exception) and o is not None, o
    x = np.concatenate([o[2] for o in out])
    y = np.concatenate([o[5] for o in out])
    np.testing.assert_array_equal(y, oracle.spmv(A, b))          # integer-valued A, b in 1/8ths: exact
    st0 = out[0][3]
    for o in out:                                                # every rank took the same decisions
        assert (o[3]["iters"], o[3]["half_exit"], o[3]["converged"]) == (st0["iters"], st0["half_exit"], st0["converged"])
        np.testing.assert_array_equal(o[4], out[0][4])
    xo, so, ho = oracle.pbicgstab(A, b, maxit=200, tol=1e-8, want_hist=True)
    assert st0["converged"] and abs(st0["iters"] - so.iters) <= 1
    assert np.linalg.norm(x - xo) / np.linalg.norm(xo) <= 1e-8
    np.testing.assert_allclose(x, xs, rtol=1e-7)
    k = min(len(out[0][4]), 6)
    np.testing.assert_allclose(out[0][4][:k], ho[:k], rtol=1e-9)
    # single-GPU solve of the same system gives the same answer to rounding
    ctx = cm.Context(0)
    s = cm.Solver.from_host_csr(ctx, A.rowptr, A.colidx, A.val)
    db, dx = ctx.array(b), ctx.array(np.ones(n))
    st1 = s.solve(db, dx, loop=cm.LOOP_PBICGSTAB, maxit=200, tol=1e-8)
    assert abs(st1.iters - st0["iters"]) <= 1
    assert np.linalg.norm(dx.download() - x) / np.linalg.norm(x) <= 1e-8
    s.close()
    ctx.close()
    # collective counts: init (1 gather for x0, 1 for the spmv call, 1 reduce + 2 setup agreements: the spmv call and
    # the solve), then 2 + 3 per iteration, plus whatever the host enqueued past the stopping point (at most kLag + 1
    # iterations)
    it, half = st0["iters"], st0["half_exit"]
    ran = it + (1 if half else 0)
    assert 2 + 2 * ran <= out[0][6] <= 2 + 2 * (ran + 3)
    assert 3 + 3 * ran <= out[0][7] <= 3 + 3 * (ran + 3)


def test_sharded_rejects_mismatched_blocks_and_ilu(cm, oracle):
    import dist_sim
    A = oracle.rand_rows(1000, 10, 1)
    ctx = cm.Context(0)
    s = cm.Solver.from_host_csr(ctx, A.rowptr[:401] - A.rowptr[0], A.colidx[:A.rowptr[400]], A.val[:A.rowptr[400]],
                                n_cols=1000)
    comm = dist_sim.ThreadComm(cm, dist_sim.ThreadGroup(2), 0, ctx)
    with pytest.raises(cm.CudamatError) as e:      # rank 0 of 2 must own 500 rows, has 400
        s.set_comm(comm.struct)
    assert e.value.code == 2
    s.close()
    ctx.close()


def test_torch_nccl_world1_forced_sharded():
    """TorchComm over the NCCL (= RCCL) backend on the real GPU: pointer->tensor views, stream
    ordering against the context's stream (= torch's current stream) and the collective calls.
    Runs in a subprocess that imports torch first (one HIP runtime per process)."""
    import socket
    import subprocess
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_nccl_world1_worker.py")
    r = subprocess.run([sys.executable, worker, str(port)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "NCCL_WORLD1_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_cpp_sharded_loop_with_blocked_spmv(cm, oracle, monkeypatch):
    """the row-sharded loop with the blocked two-phase SpMV forced on every (rectangular) shard:
    n_local x n column blocks over the gathered vector, 8 emulated ranks, uneven last block"""
    import dist_sim
    monkeypatch.setenv("CUDAMAT_SPMV_MODE", "pb")
    world, n, per_row = 8, 30011, 24
    A = oracle.rand_rows(n, per_row, 0xBEEF)
    xs = oracle.xstar(n, 0x5EEE)
    b = oracle.spmv(A, xs)
    group = dist_sim.ThreadGroup(world)
    out = [None] * world
    th = [threading.Thread(target=_run_rank, args=(cm, group, r, n, A, b, out),
                           kwargs=dict(loop=cm.LOOP_PBICGSTAB, maxit=200, tol=1e-8)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=180)
    for o in out:
        assert not isinstance(o, Exception) and o is not None, o
    x = np.concatenate([o[2] for o in out])
    y = np.concatenate([o[5] for o in out])
    np.testing.assert_array_equal(y, oracle.spmv(A, b))
    xo, so = oracle.pbicgstab(A, b, maxit=200, tol=1e-8)
    assert out[0][3]["converged"] and abs(out[0][3]["iters"] - so.iters) <= 1
    assert np.linalg.norm(x - xo) / np.linalg.norm(xo) <= 1e-8


@pytest.mark.parametrize("world,n,per_row,chunks", [(8, 30011, 24, 4), (3, 50021, 40, 3), (2, 40000, 30, 1)])
def test_overlapped_gather_is_bit_identical(cm, oracle, monkeypatch, world, n, per_row, chunks):
    """the gather in pieces on the communicator's stream with phase 1 of the blocked SpMV chasing the pieces
    (solver.hip spmv_local) against the plain all-gather + SpMV on the same shards: same y, same iterates, same
    residual history, bit for bit; and the pieces really were exchanged piecewise (no plain all-gather)"""
    import dist_sim
    monkeypatch.setenv("CUDAMAT_SPMV_MODE", "pb")
    monkeypatch.setenv("CUDAMAT_OVERLAP_CHUNKS", str(chunks))
    A = oracle.rand_rows(n, per_row, 0xBEEF)
    xs = oracle.xstar(n, 0x5EEE)
    b = oracle.spmv(A, xs)
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("CUDAMAT_OVERLAP", mode)
        group = dist_sim.ThreadGroup(world)
        out = [None] * world
        th = [threading.Thread(target=_run_rank, args=(cm, group, r, n, A, b, out),
                               kwargs=dict(pieces=True, loop=cm.LOOP_PBICGSTAB, maxit=200, tol=1e-8)) for r in range(world)]
        for t in th:
            t.start()
        for t in th:
            t.join(timeout=180)
        for o in out:
            assert not isinstance(o, Exception) and o is not None, o
        res[mode] = out
    on, off = res["1"], res["0"]
    assert all(o[8] > 0 and o[6] == 0 for o in on), "overlap on: pieces only"
    assert all(o[8] == 0 and o[6] > 0 for o in off), "overlap off: plain all-gathers only"
    for a, c in zip(on, off):
        np.testing.assert_array_equal(a[5], c[5])             # y = A b
        np.testing.assert_array_equal(a[2], c[2])             # x
        np.testing.assert_array_equal(a[4], c[4])             # residual history
        assert a[3]["iters"] == c[3]["iters"] and a[3]["converged"]
    y = np.concatenate([o[5] for o in on])
    np.testing.assert_array_equal(y, oracle.spmv(A, b))
    x = np.concatenate([o[2] for o in on])
    np.testing.assert_allclose(x, xs, rtol=1e-7)


@pytest.mark.parametrize("world,n,per_row", [(4, 30011, 24), (2, 20000, 50)])
def test_pipelined_loop_sharded(cm, oracle, monkeypatch, world, n, per_row):
    """CUDAMAT_LOOP_PIPELINED row-sharded: the reductions of an iteration on the communicator's reduce stream
    (beside the SpMV) or on the solver's stream give the same iterates bit for bit; the sharded solve agrees with
    the oracle's pipelined restatement (history 1e-9, same exit) and returns x*; 2 all-gathers and 2 all-reduces per
    iteration (the standard loop needs 3)"""
    import dist_sim
    monkeypatch.setenv("CUDAMAT_SPMV_MODE", "pb")
    A = oracle.rand_rows(n, per_row, 0xBEEF)
    xs = oracle.xstar(n, 0x5EEE)
    b = oracle.spmv(A, xs)
    res = {}
    for side in (True, False):
        group = dist_sim.ThreadGroup(world)
        out = [None] * world
        th = [threading.Thread(target=_run_rank, args=(cm, group, r, n, A, b, out),
                               kwargs=dict(pieces=True, side_reduce=side, loop=cm.LOOP_PIPELINED, maxit=200, tol=1e-8))
              for r in range(world)]
        for t in th:
            t.start()
        for t in th:
            t.join(timeout=180)
        for o in out:
            assert not isinstance(o, Exception) and o is not None, o
        res[side] = out
    for a, c in zip(res[True], res[False]):
        np.testing.assert_array_equal(a[2], c[2])
        np.testing.assert_array_equal(a[4], c[4])
        assert a[3]["iters"] == c[3]["iters"] and a[3]["half_exit"] == c[3]["half_exit"] and a[3]["converged"]
    o0 = res[True][0]
    it, half = o0[3]["iters"], o0[3]["half_exit"]
    ran = it + (1 if half else 0)
    assert 2 * ran <= o0[9] <= 2 * (ran + 3) and res[False][0][9] == 0            # side all-reduces: 2 per iteration
    xo, so, ho = oracle.pipelined_bicgstab(A, b, maxit=200, tol=1e-8, want_hist=True)
    assert (it, half) == (so.iters, so.half_exit)
    k = min(len(o0[4]), 6)
    np.testing.assert_allclose(o0[4][:k], ho[:k], rtol=1e-9)
    x = np.concatenate([o[2] for o in res[True]])
    np.testing.assert_allclose(x, xs, rtol=1e-7)
    assert np.linalg.norm(x - xo) / np.linalg.norm(xo) <= 1e-8


@pytest.mark.parametrize("world,name", [(2, "poisson60x50"), (4, "mat10000"), (3, "rand9001x20")])
def test_preconditioned_pipelined_loop_sharded(cm, oracle, golden_dir, world, name):
    """SURVEY 8 f4, both halves together: the pipelined loop (2 hidden all-reduces per iteration) with block-Jacobi
    ILU(0) of each rank's diagonal block, row-sharded over emulated ranks.  Oracle: the preconditioned pipelined
    restatement with M = blockdiag(ILU0(A_rr)): iteration count +-10 % (>= +-2), history 1e-7, solution 1e-6; and the
    reference's loop with the same M: solution 1e-5, iteration count +-15 % (>= +-2) -- a different algorithm in floating
    point, and on the 4-block mat10000 the reference loop's OWN count moves between 42 and 45 when b is perturbed by
    1e-15 relative, the pipelined restatement's between 37 and 49 over replacement periods 8...64 (measured with the
    oracle), so the cross-algorithm band is wider than the like-for-like one."""
    import dist_sim
    if name.startswith("poisson"):
        A = oracle.poisson5(60, 50)
    elif name.startswith("rand"):
        A = oracle.rand_rows(9001, 20, 0xC0FFEE)
    else:
        err, m, n_, nnz, v, ia, ja = cm.loadMMSparseMatrix(os.path.join(golden_dir, name + ".mtx"))
        assert err == 0
        A = oracle.Csr(m, ia, ja, v, m)
    n = A.n
    xs = 1.0 + np.sin(np.arange(n))
    b = oracle.spmv(A, xs)
    vm = _block_jacobi_vm(oracle, A, world)
    group = dist_sim.ThreadGroup(world)
    out = [None] * world
    th = [threading.Thread(target=_run_rank_bj, args=(cm, group, r, n, A, b, out),
                           kwargs=dict(loop=cm.LOOP_PIPELINED, maxit=500, tol=1e-8)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=180)
    for o in out:
        assert not isinstance(o, Exception) and o is not None, o
    x = np.concatenate([o[2] for o in out])
    st0 = out[0][3]
    for o in out:
        assert (o[3]["iters"], o[3]["half_exit"], o[3]["converged"]) == (st0["iters"], st0["half_exit"], st0["converged"])
        np.testing.assert_array_equal(o[4], out[0][4])
    xo, so, ho = oracle.pipelined_bicgstab(A, b, vm=vm, maxit=500, tol=1e-8, want_hist=True)
    xr, sr = oracle.pbicgstab(A, b, vm=vm, maxit=500, tol=1e-8)
    assert st0["converged"] and so.converged and st0["restarts"] == 0
    assert abs(st0["iters"] - so.iters) <= max(2, so.iters // 10), (st0["iters"], so.iters)
    assert abs(st0["iters"] - sr.iters) <= max(2, 0.15 * sr.iters), (st0["iters"], sr.iters)
    k = min(len(out[0][4]), 4)
    np.testing.assert_allclose(out[0][4][:k], ho[:k], rtol=1e-7)
    assert np.linalg.norm(x - xo) / np.linalg.norm(xo) <= 1e-6
    assert np.linalg.norm(x - xr) / np.linalg.norm(xr) <= 1e-5
    np.testing.assert_allclose(x, xs, rtol=0, atol=1e-4)


@pytest.mark.parametrize("world,name", [(4, "poisson"), (3, "band"), (2, "rand")])
def test_windowed_gather_halo(cm, oracle, monkeypatch, world, name):
    """a banded matrix needs a halo, not the whole gather.  With gather_window in the communicator the ranks
    exchange, once, which part of every other slice their rows reference; when every rank needs at most half of the
    whole gather, only those windows travel.  Same iterates as the whole gather bit for bit; a scattered matrix keeps
    the whole gather."""
    import dist_sim
    monkeypatch.setenv("CUDAMAT_SPMV_MODE", "csr")     # one SpMV form in both runs (the tuner's pick may differ by timing)
    if name == "poisson":
        A = oracle.poisson5(100, 120)
    elif name == "band":
        import scipy.sparse as sp
        rng = np.random.default_rng(4)
        n0 = 30011
        rows = np.repeat(np.arange(n0), 21)
        cols = np.clip(rows + rng.integers(-900, 901, rows.size), 0, n0 - 1)
        S = sp.csr_matrix((np.ones(rows.size), (rows, cols)), shape=(n0, n0))
        S.sum_duplicates()
        S.data[:] = -1.0
        S.setdiag(25.0)
        S.sort_indices()
        A = oracle.Csr(n0, S.indptr.astype(np.int32), S.indices.astype(np.int32), S.data.astype(np.float64), n0)
    else:
        A = oracle.rand_rows(20000, 30, 0xBEEF)
    n = A.n
    xs = oracle.xstar(n, 0x5EEE)
    b = oracle.spmv(A, xs)
    res = {}
    for windows in (True, False):
        group = dist_sim.ThreadGroup(world)
        out = [None] * world
        th = [threading.Thread(target=_run_rank, args=(cm, group, r, n, A, b, out),
                               kwargs=dict(windows=windows, loop=cm.LOOP_PBICGSTAB, maxit=1000, tol=1e-8)) for r in range(world)]
        for t in th:
            t.start()
        for t in th:
            t.join(timeout=180)
        for o in out:
            assert not isinstance(o, Exception) and o is not None, o
        res[windows] = out
    for a, c in zip(res[True], res[False]):
        np.testing.assert_array_equal(a[5], c[5])             # y = A b
        np.testing.assert_array_equal(a[2], c[2])             # x
        np.testing.assert_array_equal(a[4], c[4])             # residual history
    y = np.concatenate([o[5] for o in res[True]])
    np.testing.assert_array_equal(y, oracle.spmv(A, b))
    on = res[True]
    per = (n + world - 1) // world
    if name == "rand":
        assert all(o[10] == 0 and o[3]["overlapped"] == 0 and o[3]["gather_fraction"] == 1.0 for o in on), "scattered columns: whole gather"
    else:
        assert all(o[10] > 0 and o[3]["overlapped"] == 2 for o in on)
        assert all(o[6] == 1 for o in on), "one all-gather: the exchange of the windows themselves"
        # the doubles that travelled are the halo, a small fraction of (world - 1) slices per gather
        for o in on:
            assert o[11] <= 0.1 * o[10] * (world - 1) * per
            assert 0 < o[3]["gather_fraction"] <= 0.1
    assert on[0][3]["converged"]
    x = np.concatenate([o[2] for o in on])
    np.testing.assert_allclose(x, xs, rtol=5e-5)      # tol 1e-8 on the residual of a Laplacian (kappa ~ 1e3-1e4)


def test_bench_two_processes_share_the_gpu_over_gloo():
    """the REAL multi-process path of bench.py (torch.distributed.run, one rank per process, row shards
    generated per rank, TorchComm on device buffers, lagged stop decisions across processes) with two
    ranks on this box's single GPU.  RCCL refuses two ranks on one device, so the rehearsal uses gloo;
    bench.py's own gate (solve converges to x*, recursive = true residual) must pass on both ranks."""
    import json
    import socket
    import subprocess
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    # the gather in pieces (phase 1 of the blocked SpMV per piece, pieces exchanged with isend/irecv) and the plain
    # all-gather: two real processes, the same iterates bit for bit
    for form in ("torch:1", "torch:0"):
        env = dict(os.environ, CUDAMAT_BENCH_ONE_DEVICE="1", CUDAMAT_BENCH_BACKEND="gloo", CUDAMAT_BENCH_FORMS=form,
                   CUDAMAT_SPMV_MODE="pb")           # (the pieces ride on the blocked SpMV; at this size the tuner may pick CSR)
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"),
                            "--gpus", "2", "--rows", "600000", "--steps", "6", "--warmup", "1", "--cpu-baseline", "off"],
                           capture_output=True, text=True, timeout=600, env=env, cwd=root)
        assert r.returncode == 0, r.stderr[-3000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, r.stdout[-2000:]
        out = json.loads(lines[0])
        assert out["n_gpus"] == 2 and out["value"] > 0 and out["config"]["converges_in_iters"] <= 10
        assert out["scaling"] == "strong" and "roofline" in out
        assert out["comm"]["gate"][-1]["failed_ranks"] == 0 and len(out["comm"]["gate"]) == 1
        outs[form] = out
    assert outs["torch:1"]["comm"]["form"]["gather"].startswith("in pieces")
    assert outs["torch:0"]["comm"]["form"]["gather"] == "plain all-gather"
    assert outs["torch:1"]["config"]["gate_x_sha256"] == outs["torch:0"]["config"]["gate_x_sha256"]
    assert outs["torch:1"]["config"]["converges_in_iters"] == outs["torch:0"]["config"]["converges_in_iters"]


def test_bench_two_processes_pipelined_loop_with_block_jacobi():
    """SURVEY 8 f4 end to end in the multi-process bench: two ranks (one GPU, gloo), the pipelined loop with block-Jacobi
    ILU(0) -- bench.py's own gate (the solve converges to x*, the loop's residual is the true residual on every rank) must
    pass, and the standard loop with the same preconditioner must converge to the same solution (the digests of the two
    gates differ only if the iterates do: both are checked against x* to 2e-5 by the gate itself)"""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for loop in ("pipelined", "pbicgstab"):
        env = dict(os.environ, CUDAMAT_BENCH_ONE_DEVICE="1", CUDAMAT_BENCH_BACKEND="gloo", CUDAMAT_BENCH_FORMS="torch:0",
                   CUDAMAT_SPMV_MODE="pb")
        for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k, None)
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rows", "600000", "--steps", "6",
                            "--warmup", "1", "--cpu-baseline", "off", "--precond", "bjilu0", "--loop", loop],
                           capture_output=True, text=True, timeout=900, env=env, cwd=root)
        assert r.returncode == 0, r.stderr[-3000:]
        out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
        assert out["n_gpus"] == 2 and out["value"] > 0 and out["comm"]["gate"][-1]["failed_ranks"] == 0
        assert out["config"]["converges_in_iters"] is not None and out["config"]["converges_in_iters"] <= 6
        assert "trsv_ms_per_apply" in out and out["trsv_ms_per_apply"] > 0
        outs[loop] = out
    assert outs["pipelined"]["config"]["loop"].startswith("pipelined") and outs["pbicgstab"]["config"]["loop"].startswith("pbicgstab.cu")
    assert abs(outs["pipelined"]["config"]["converges_in_iters"] - outs["pbicgstab"]["config"]["converges_in_iters"]) <= 1


def test_bench_line_contract_on_one_gpu():
    """the one-GPU bench line (a small C4-shaped system): metric / value / roofline / side figures as DESIGN section 6
    describes them -- the judged numbers are measured on fp64 values (value_dictionary == 0 in the timed region), the
    dictionary run is the labelled side figure, the drop-in section times two host-pointer calls (the second reuses the
    first one's plan), stdout carries exactly one line"""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "CUDAMAT_BENCH_FORMS", "CUDAMAT_VALUE_DICT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--rows", "3000000", "--steps", "10", "--warmup", "2",
                        "--cpu-baseline", "off"], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["steps"] == 10 and out["warmup"] == 2 and out["unit"] == "iter/s" and out["dtype"] == "f64"
    assert out["higher_is_better"] is True and out["vs_baseline"] is None and out["value"] > 0
    assert abs(out["value"] - 1e3 / out["ms_per_step"]) <= 1e-6 * out["value"]
    rf = out["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert rf["launches_timed"] == 20 and rf["avg_launch_ms"] > 0 and "k_pb_phase1 +" in rf["kernel"]
    assert out["value_dictionary"] == 0                               # the timed kernels read fp64 values
    side = out["with_value_dictionary"]
    assert side["value_dictionary"] == 39 and side["value"] > 0 and "effective_frac" in side and "frac" not in side
    d = out["drop_in"]
    assert d["first_call"]["plan_reused"] == 0 and d["second_call_same_matrix"]["plan_reused"] == 1
    assert d["first_call"]["converged"] and d["first_call"]["max_abs_err"] < 2e-5 and d["first_call"]["tune_s"] >= 0
    assert d["second_call_same_matrix"]["setup_s"] == 0


def test_bench_plain_command_launches_its_ranks_and_survives_a_hanging_form():
    """`python bench.py --gpus 2` with no launcher around it (how the driver produced BENCH): bench.py starts the ranks itself
    as child processes, prints one line with n_gpus = 2 -- and when the first exchange form never completes (forced: the
    workers sleep in front of its gate) the supervisors end that set of workers at the time limit and the second form is
    timed by fresh processes.  Two ranks on this box's one GPU, gloo exchanges (RCCL refuses two ranks per device)."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CUDAMAT_BENCH_ONE_DEVICE="1", CUDAMAT_BENCH_BACKEND="gloo", CUDAMAT_SPMV_MODE="pb",
               CUDAMAT_BENCH_TEST_HANG="torch:1", CUDAMAT_BENCH_FORM_TIMEOUT="90")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "CUDAMAT_BENCH_FORMS"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rows", "600000", "--steps", "6",
                        "--warmup", "1", "--cpu-baseline", "off"], capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["value"] > 0
    log = out["comm"]["launcher"]
    assert [e["form"] for e in log] == ["torch:1", "torch:0"] and [e["ok"] for e in log] == [False, True]
    assert "time limit" in log[0]["ranks"]
    assert out["comm"]["form"]["gather"] == "plain all-gather" and out["comm"]["gate"][-1]["failed_ranks"] == 0


def _block_jacobi_vm(oracle, A, world):
    """ILU(0) of every rank's diagonal block, written on A's pattern with zeros outside the blocks: with these
    values the oracle's triangular solves over A's pattern ARE the block-Jacobi preconditioner"""
    import scipy.sparse as sp
    from cuda_mat_amd.dist import shard_rows
    base = int(A.rowptr[0])
    S = sp.csr_matrix((A.val, A.colidx - base, A.rowptr - base), shape=(A.n, A.n))
    row_of = np.repeat(np.arange(A.n), np.diff(A.rowptr))
    col = A.colidx - base
    vm = np.zeros(A.val.size)
    for r in range(world):
        r0, r1, _ = shard_rows(A.n, world, r)
        B = S[r0:r1, r0:r1].tocsr()
        B.sort_indices()
        lu = oracle.ilu0(oracle.Csr(r1 - r0, B.indptr.astype(np.int32), B.indices.astype(np.int32), B.data.copy(), r1 - r0))
        inside = (row_of >= r0) & (row_of < r1) & (col >= r0) & (col < r1)
        assert inside.sum() == lu.size
        vm[inside] = lu                     # same (row, column) order in both
    return vm


def _run_rank_bj(cm, group, rank, n, A, b, out, **kw):
    import dist_sim
    from cuda_mat_amd.dist import shard_rows
    try:
        ctx = cm.Context(0)
        row0, row1, per = shard_rows(n, group.world, rank)
        base = int(A.rowptr[0])
        rp = (A.rowptr[row0:row1 + 1] - A.rowptr[row0]).astype(np.int32)
        k0, k1 = A.rowptr[row0] - base, A.rowptr[row1] - base
        s = cm.Solver.from_host_csr(ctx, rp, A.colidx[k0:k1] - base, A.val[k0:k1], n_cols=n)
        comm = dist_sim.ThreadComm(cm, group, rank, ctx)
        s.set_comm(comm.struct)
        with pytest.raises(cm.CudamatError):          # ILU(0) of the whole matrix does not shard
            s.ilu0()
        s.block_ilu0()
        lu = s.ilu0_values()
        db, dx = ctx.array(b[row0:row1]), ctx.array(np.ones(row1 - row0))
        dz = ctx.empty(row1 - row0)
        s.precond_apply(db, dz)
        st = s.solve(db, dx, precond=cm.PRECOND_BLOCK_ILU0, **kw)
        out[rank] = (row0, row1, dx.download(), st.as_dict(), s.history(), lu, dz.download(), comm.n_allgather, comm.n_allreduce)
        s.close()
        ctx.close()
    except Exception as e:  # noqa: BLE001
        group.barrier.abort()
        out[rank] = e


@pytest.mark.parametrize("world,name", [(2, "poisson60x50"), (4, "poisson60x50"), (3, "rand9001x20"), (4, "mat10000")])
def test_block_jacobi_ilu0_sharded_vs_oracle(cm, oracle, golden_dir, world, name):
    """the sharded loop with ILU(0) of each rank's diagonal block as preconditioner (no collective
    in the preconditioner).  Parity statement: NOT the reference's maths for world > 1 -- the oracle is the same
    restated loop (pbicgstab.cu:45-154) with M = blockdiag(ILU0(A_rr)); block factors rtol 1e-12, M^-1 b rtol
    1e-10, iteration counts equal +-1, solution 1e-7 relative."""
    import dist_sim
    if name.startswith("poisson"):
        A = oracle.poisson5(60, 50)
    elif name.startswith("rand"):
        A = oracle.rand_rows(9001, 20, 0xC0FFEE)
    else:
        err, m, n_, nnz, v, ia, ja = cm.loadMMSparseMatrix(os.path.join(golden_dir, name + ".mtx"))
        assert err == 0
        A = oracle.Csr(m, ia, ja, v, m)
    n = A.n
    xs = 1.0 + np.sin(np.arange(n))
    b = oracle.spmv(A, xs)
    vm = _block_jacobi_vm(oracle, A, world)
    group = dist_sim.ThreadGroup(world)
    out = [None] * world
    th = [threading.Thread(target=_run_rank_bj, args=(cm, group, r, n, A, b, out),
                           kwargs=dict(loop=cm.LOOP_PBICGSTAB, maxit=500, tol=1e-8)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=180)
    for o in out:
        assert not isinstance(o, Exception) and o is not None, o
    x = np.concatenate([o[2] for o in out])
    # factors of every block and one application of the preconditioner
    base = int(A.rowptr[0])
    row_of = np.repeat(np.arange(n), np.diff(A.rowptr))
    col = A.colidx - base
    for o in out:
        r0, r1 = o[0], o[1]
        inside = (row_of >= r0) & (row_of < r1) & (col >= r0) & (col < r1)
        np.testing.assert_allclose(o[5], vm[inside], rtol=1e-12, atol=1e-14)
    z = np.concatenate([o[6] for o in out])
    zo = oracle.trsv_upper(A, vm, oracle.trsv_lower_unit(A, vm, b))
    np.testing.assert_allclose(z, zo, rtol=1e-10, atol=1e-12)
    # the solve
    xo, so, ho = oracle.pbicgstab(A, b, vm=vm, maxit=500, tol=1e-8, want_hist=True)
    st0 = out[0][3]
    for o in out:
        assert (o[3]["iters"], o[3]["half_exit"], o[3]["converged"]) == (st0["iters"], st0["half_exit"], st0["converged"])
    assert st0["converged"] and abs(st0["iters"] - so.iters) <= max(1, so.iters // 10)
    # 1e-6: both sides are deterministic (fixed-order HIP reductions, single-threaded oracle, see the fixture above)
    assert np.linalg.norm(x - xo) / np.linalg.norm(xo) <= 1e-6
    np.testing.assert_allclose(x, xs, rtol=0, atol=1e-4)       # tol 1e-8 on the residual; x* = 1 + sin(i)
    k = min(len(out[0][4]), 4)
    np.testing.assert_allclose(out[0][4][:k], ho[:k], rtol=1e-8)
    # collectives: none added by the preconditioner (2 gathers + 3 reduces per iteration as without it;
    # +1 gather... none for precond_apply)
    it, half = st0["iters"], st0["half_exit"]
    ran = it + (1 if half else 0)
    assert 1 + 2 * ran <= out[0][7] <= 1 + 2 * (ran + 3)
    assert 2 + 3 * ran <= out[0][8] <= 2 + 3 * (ran + 3)             # (+1: the setup agreement of the solve)


def test_bench_two_processes_poisson_exchanges_a_halo():
    """bench.py --workload poisson5 on two ranks (gloo rehearsal on one GPU): the ranks find that their rows reference
    only a band of the other slice, exchange windows instead of gathering, pass the gate, and say so in the JSON"""
    import json
    import socket
    import subprocess
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CUDAMAT_BENCH_ONE_DEVICE="1", CUDAMAT_BENCH_BACKEND="gloo", CUDAMAT_BENCH_FORMS="torch:0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"),
                        "--gpus", "2", "--workload", "poisson5", "--rows", "400000", "--nx", "800", "--steps", "6", "--warmup", "1",
                        "--cpu-baseline", "off"], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert out["n_gpus"] == 2 and out["comm"]["gate"][-1]["failed_ranks"] == 0
    assert out["comm"]["form"]["gather"].startswith("windows only"), out["comm"]["form"]
