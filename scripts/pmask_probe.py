"""timing-only probe (results are wrong by construction): C4's blocked SpMV with the product stream folded into a
window of CUDAMAT_PB_PMASK_MB megabytes -- the upper bound of what an Infinity-Cache-resident product ring could give
the REAL phase-1 / phase-2 kernels (run under rocprofv3 --kernel-trace --stats for the per-kernel split).
The knob exists only in commits d474d75 / 7505629 (reverted); logs: profiles/r02_probes/pmask*_probe.log"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cuda_mat_amd as cm
os.environ["CUDAMAT_SPMV_MODE"] = "pb"
n, per = 10_000_000, 50
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev)
with torch.cuda.stream(stream):
    ctx = cm.Context(0, stream=stream.cuda_stream)
    rn = cm.lib().cudamat_rand_row_nnz(n, per)
    nnz = n * rn
    rp = torch.empty(n + 1, dtype=torch.int32, device=dev)
    ci = torch.empty(nnz, dtype=torch.int32, device=dev)
    va = torch.empty(nnz, dtype=torch.float64, device=dev)
    ctx.gen_rand_rows(n, per, 0x5EED, 0, n, 0, rp, ci, va)
    s = cm.Solver(ctx, n, n, nnz, rp, ci, va, 0)
    del rp, ci, va
    torch.cuda.empty_cache()
    x = torch.ones(n, dtype=torch.float64, device=dev)
    y = torch.empty(n, dtype=torch.float64, device=dev)
    for _ in range(3):
        s.spmv(x, y)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    K = 20
    for _ in range(K):
        s.spmv(x, y)
    e1.record(stream)
    e1.synchronize()
    print("PMASK_MB=%s  %.3f ms per SpMV" % (os.environ.get("CUDAMAT_PB_PMASK_MB", "-"), e0.elapsed_time(e1) / K), flush=True)
