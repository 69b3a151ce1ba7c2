import torch, time
dev = torch.device("cuda", 0)
n = 1 << 28
x = torch.empty(n, dtype=torch.float64, device=dev); y = torch.empty(n, dtype=torch.float64, device=dev)
def t(f, reps=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps
ms = t(lambda: x.fill_(1.5)); print("pure write  %.3f ms  %.0f GB/s" % (ms, 8 * n / ms / 1e6))
ms = t(lambda: x.sum()); print("pure read   %.3f ms  %.0f GB/s" % (ms, 8 * n / ms / 1e6))
ms = t(lambda: y.copy_(x)); print("copy 1R+1W  %.3f ms  %.0f GB/s total" % (ms, 16 * n / ms / 1e6))
ms = t(lambda: y.add_(x)); print("axpy 2R+1W  %.3f ms  %.0f GB/s total" % (ms, 24 * n / ms / 1e6))
