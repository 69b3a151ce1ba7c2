"""GPU box: do two streams whose addresses are equal modulo a large power of two (every hipMalloc is 2 MB aligned) fight for
the same HBM banks?  512 MiB copy (1 read : 1 write) and y += a x (2 reads : 1 write) with the destination shifted by a few
offsets; median of 9 HIP-event timings each.   usage: python scripts/align_probe.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import cuda_mat_amd as cm
from cuda_mat_amd import _lib

ctx = cm.Context(0)
n = 1 << 26
pad = 64 << 20
x = ctx.empty(n + pad // 8)
y = ctx.empty(n + pad // 8)
x.zero(); y.zero()
t = ctx.timer()


def med(fn):
    ts = []
    for i in range(11):
        t.start(); fn(); t.stop()
        if i >= 2:
            ts.append(t.elapsed_ms())
    return sorted(ts)[len(ts) // 2]


for off in (0, 128, 256, 1024, 4096, 4096 + 256, 65536, 65536 + 4096 + 256, 1 << 20, (1 << 20) + 4096 + 256, (3 << 20) + 12288 + 768):
    yp = y.ptr + off
    ms_c = med(lambda: _lib.check(_lib.lib().cudamat_d2d(ctx.h, yp, x.ptr, 8 * n)))
    ms_a = med(lambda: _lib.check(_lib.lib().cudamat_axpy(ctx.h, n, 0.5, x.ptr, yp)))
    print("dst offset %9d B: copy %.1f GB/s   y += a x %.1f GB/s" % (off, 16.0 * n / ms_c / 1e6, 24.0 * n / ms_a / 1e6), flush=True)
