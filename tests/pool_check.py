"""Run by tests/test_gpu_parity.py::test_device_memory_pool_recycles_and_trims in a process of its own: the recycling pool of
csrc/pool.cpp seen through cudamat_malloc / cudamat_free / cudamat_pool_trim / cudamat_mem_info."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cuda_mat_amd as cm  # noqa: E402


def main():
    ctx = cm.Context(0)
    lib = cm.lib()
    GB = 1 << 30

    def free_now(pool=False):
        ctx.sync()
        f, t, pf = C.c_size_t(), C.c_size_t(), C.c_size_t()
        assert lib.cudamat_mem_info(0, C.byref(f), C.byref(t), C.byref(pf)) == 0
        return pf.value if pool else f.value

    def same(a, b):
        return abs(a - b) <= (32 << 20)          # (the runtime's own staging buffers for the uploads below come and go: 2 MB seen)

    assert lib.cudamat_pool_trim() == 0
    f0 = free_now()
    a = ctx.empty(GB // 8)                       # 1 GB
    pa = a.ptr
    assert free_now() <= f0 - GB + (64 << 20)
    a.free()
    f1 = free_now()
    assert f1 <= f0 - GB + (64 << 20) and free_now(pool=True) >= GB            # still with the pool
    b = ctx.empty(GB // 8)
    assert b.ptr == pa and same(free_now(), f1)      # recycled, no driver call
    b.free()
    # two pieces out of the one free block, written and read back: distinct memory
    c, d = ctx.empty(GB // 32), ctx.empty(GB // 32)          # 256 MB each: the 1 GB block is split (>= 64 MB left over)
    assert same(free_now(), f1) and c.ptr == pa and d.ptr == pa + (GB // 4)
    c.upload(np.full(GB // 32, 1.5))
    d.upload(np.full(GB // 32, -2.5))
    assert c.download()[-1] == 1.5 and d.download()[0] == -2.5
    c.free()
    d.free()
    e = ctx.empty(GB // 8)                       # the pieces merged again: the whole 1 GB fits where it was
    assert e.ptr == pa and same(free_now(), f1)
    e.free()
    # a small request is not pooled
    s = ctx.empty(1000)
    s.free()
    assert lib.cudamat_pool_trim() == 0
    assert free_now() >= f0 - (64 << 20) and free_now(pool=True) == 0
    ctx.close()
    print("pool_check ok")


if __name__ == "__main__":
    main()
