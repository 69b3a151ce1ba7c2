"""subprocess body of tests/test_gpu_dist.py::test_torch_nccl_world1_forced_sharded.
torch is imported FIRST: torch wheels bundle their own HIP runtime, and a process must not
initialise two of them (load order: torch, then libcudamat_hip.so, which then binds to it)."""
import os
import sys

import torch  # noqa: F401  (first!)
import torch.distributed as dist
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["CUDAMAT_FORCE_SHARDED"] = "1"

import cuda_mat_amd as cm  # noqa: E402
from cuda_mat_amd.dist import TorchComm  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main():
    n = 20000
    A = O.rand_rows(n, 50, 0x5EED)
    xs = O.xstar(n, 0x5EEE)
    b = O.spmv(A, xs)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%s" % sys.argv[1], rank=0, world_size=1,
                            device_id=dev)
    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        ctx = cm.Context(0, stream=stream.cuda_stream)
        comm = TorchComm(device=dev)
        s = cm.Solver.from_host_csr(ctx, A.rowptr, A.colidx, A.val)
        s.set_comm(comm.struct)
        tb = torch.from_numpy(b).to(dev)
        tx = torch.ones(n, dtype=torch.float64, device=dev)
        st = s.solve(tb, tx, loop=cm.LOOP_PBICGSTAB, maxit=200, tol=1e-8)
        stream.synchronize()
        assert comm.error is None, comm.error
        assert comm.n_allgather >= 1 + 2 * st.iters and comm.n_allreduce >= 1 + 3 * st.iters
        x = tx.cpu().numpy()
        s.close()
        # the library's own RCCL binding (csrc/comm_rccl.hip) at world size 1: librccl is bound with dlopen (the copy
        # this process has already loaded), a communicator pair is created, and the loop's all-gathers / all-reduces
        # are real ncclAllGather / ncclAllReduce calls made from C++
        from cuda_mat_amd.dist import RcclComm
        assert cm.lib().cudamat_rccl_available() == 1
        rc = RcclComm(ctx, 0, 1)
        s = cm.Solver.from_host_csr(ctx, A.rowptr, A.colidx, A.val)
        s.set_comm(rc.struct)
        tx2 = torch.ones(n, dtype=torch.float64, device=dev)
        st2 = s.solve(tb, tx2, loop=cm.LOOP_PBICGSTAB, maxit=200, tol=1e-8, flags=cm.FLAG_PROFILE)
        stream.synchronize()
        assert (st2.iters, st2.converged, st2.half_exit) == (st.iters, st.converged, st.half_exit)
        assert torch.equal(tx2, tx), "native RCCL collectives must reproduce the torch.distributed run bit for bit"
        assert st2.n_gather >= 1 + 2 * st.iters and st2.n_allreduce >= 1 + 3 * st.iters and st2.ms_gather > 0
        assert st2.overlapped == 0 and abs(st2.ms_gather - st2.ms_gather_exposed) < 1e-9
        # the pipelined loop: its reductions are ncclAllReduce calls on the communicator's third stream
        tx3 = torch.ones(n, dtype=torch.float64, device=dev)
        st3 = s.solve(tb, tx3, loop=cm.LOOP_PIPELINED, maxit=200, tol=1e-8, flags=cm.FLAG_PROFILE)
        stream.synchronize()
        assert st3.converged and abs(st3.iters - st.iters) <= 1 and st3.n_allreduce >= 2 * st3.iters
        assert float((tx3 - tx).abs().max()) <= 1e-7
        s.close()
        rc.close()
        ctx.close()
    dist.destroy_process_group()
    xo, so = O.pbicgstab(A, b, maxit=200, tol=1e-8)
    assert st.converged and abs(st.iters - so.iters) <= 1, (st.iters, so.iters)
    err = np.linalg.norm(x - xo) / np.linalg.norm(xo)
    assert err <= 1e-8, err
    print("NCCL_WORLD1_OK iters=%d allgather=%d allreduce=%d err=%.2e" % (st.iters, comm.n_allgather,
                                                                          comm.n_allreduce, err))


if __name__ == "__main__":
    main()
