"""world_size-2 gloo test (CPU) of the row-sharded path: shard_rows(), the cudamat_comm callbacks of
TorchComm (ctypes function pointers -> torch.distributed on the raw buffers) and the sequence of
collectives the C++ loop issues (restated in tests/dist_sim.py with the oracle as compute)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_rows_covers_everything():
    from cuda_mat_amd.dist import shard_rows
    for n in (1, 7, 8, 9, 1000, 10_000_000):
        for world in (1, 2, 3, 8):
            blocks = [shard_rows(n, world, r) for r in range(world)]
            per = blocks[0][2]
            assert per * world >= n and per == -(-n // world)
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            for r in range(world):
                r0, r1, p = blocks[r]
                assert p == per and r0 == min(n, r * per) and 0 <= r1 - r0 <= per
                if r:
                    assert r0 == blocks[r - 1][1]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, per_row, q):
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        os.environ["OMP_NUM_THREADS"] = "2"
        import torch.distributed as dist
        from cuda_mat_amd.dist import TorchComm, shard_rows
        from oracle import oracle as O
        import dist_sim
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
        comm = TorchComm(device=None)
        assert (comm.rank, comm.world) == (rank, world)
        row0, row1, per = shard_rows(n, world, rank)
        A_loc = O.rand_rows(n, per_row, 0x5EED, row0, row1)
        A = O.rand_rows(n, per_row, 0x5EED)
        b = O.spmv(A, O.xstar(n, 0x5EEE))
        x, it, half, conv, hist = dist_sim.sharded_pbicgstab(O, comm.struct, A_loc, n, per, b[row0:row1],
                                                            200, 1e-8)
        assert comm.error is None
        q.put((rank, row0, row1, x, it, half, conv, hist, comm.n_allgather, comm.n_allreduce))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, "error", traceback.format_exc()))
        raise e


@pytest.mark.parametrize("world,n", [(2, 3001), (2, 4000)])
def test_sharded_loop_over_gloo_matches_single_process(oracle, world, n):
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    for attempt in range(3):
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, world, port, n, 20, q)) for r in range(world)]
        for p in procs:
            p.start()
        res = [q.get(timeout=180) for _ in range(world)]
        for p in procs:
            p.join(timeout=60)
        # the port found free above can be taken by another process before rank 0 binds it: only that is retried
        rendezvous = [r for r in res if r[1] == "error" and any(k in r[2] for k in ("in use", "onnect", "imed out"))]
        if not rendezvous:
            break
    for r in res:
        assert r[1] != "error", r[2]
    res.sort(key=lambda t: t[0])
    x = np.concatenate([r[3] for r in res])
    A = oracle.rand_rows(n, 20, 0x5EED)
    xs = oracle.xstar(n, 0x5EEE)
    b = oracle.spmv(A, xs)
    xo, so, ho = oracle.pbicgstab(A, b, maxit=200, tol=1e-8, want_hist=True)
    # every rank took the same decisions
    assert len({(r[4], r[5], r[6]) for r in res}) == 1
    it, half, conv = res[0][4:7]
    assert conv and so.converged and abs(it - so.iters) <= 1
    np.testing.assert_allclose(x, xo, rtol=1e-8)
    np.testing.assert_allclose(x, xs, rtol=1e-7)
    k = min(len(res[0][7]), 6)
    np.testing.assert_allclose(res[0][7][:k], ho[:k], rtol=1e-9)
    # collectives per iteration: 2 all-gathers, 3 all-reduces (+1 gather, +1 reduce for r0)
    # an iteration left through the half-step test still did its 2 gathers and 2 of its reduces
    assert res[0][8] == 1 + 2 * it + (2 if half else 0)
    assert res[0][9] == 1 + 3 * it + (2 if half else 0)
