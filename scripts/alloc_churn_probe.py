"""GPU box: does hipMalloc cost depend on what the device memory held before?  Ten 8 GB allocations (never touched), freed,
ten again, then the same after writing to every byte once; per-call milliseconds."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cuda_mat_amd as cm
ctx = cm.Context(0)
n = int(8e9 / 8)
def round_(touch):
    t_m, bufs = [], []
    for _ in range(10):
        t0 = time.perf_counter(); a = ctx.empty(n); ctx.sync(); t_m.append((time.perf_counter() - t0) * 1e3); bufs.append(a)
    if touch:
        for a in bufs: a.zero()
        ctx.sync()
    t_f = []
    for a in bufs:
        t0 = time.perf_counter(); a.free(); ctx.sync(); t_f.append((time.perf_counter() - t0) * 1e3)
    return t_m, t_f
for name, touch in (("untouched", False), ("untouched again", False), ("written", True), ("after written", False), ("after written 2", False)):
    m, f = round_(touch)
    print("%-16s malloc ms: %s | free ms: %s" % (name, " ".join("%.1f" % v for v in m), " ".join("%.1f" % v for v in f)), flush=True)
