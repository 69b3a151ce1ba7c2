"""Follow-up of placement_probe.py (the loop's speed alternated 187 / 179 it/s with every re-creation of the solver in one
process, whatever the padding).  E3: re-create WITHOUT handing the memory back to the driver (the pool serves the same blocks
again): does the alternation stop?  E1: two instances alive at once, timed in turn; the first freed, a third created.

    python scripts/placement_probe2.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("CUDAMAT_VALUE_DICT", "0")
import cuda_mat_amd as cm  # noqa: E402
from placement_probe import timed  # noqa: E402


class Inst:
    def __init__(self, ctx, n, per, rp, ci, va, xs):
        self.s = cm.Solver(ctx, n, n, n * per, rp, ci, va, 0)
        self.b, self.x = ctx.empty(n), ctx.empty(n)
        self.s.spmv(xs, self.b)
        self.ctx = ctx

    def time(self, tag):
        step_ms, spmv_ms = timed(self.ctx, self.s, self.b, self.x)
        print("%-46s %.3f ms/step (%.1f it/s)  spmv %.3f ms   b at %#x" % (tag, step_ms, 1e3 / step_ms, spmv_ms, self.b.ptr), flush=True)

    def close(self):
        self.b.free()
        self.x.free()
        self.s.close()


def main():
    ctx = cm.Context(0)
    n, per = 10_000_000, 50
    rp, ci, va = ctx.empty(n + 1, np.int32), ctx.empty(n * per, np.int32), ctx.empty(n * per)
    ctx.gen_rand_rows(n, per, 7, 0, n, 0, rp, ci, va)
    xs = ctx.empty(n)
    ctx.gen_xstar(0, n, 8, xs)
    args = (ctx, n, per, rp, ci, va, xs)
    print("E3: re-created on the pool's blocks (no trim)")
    for i in range(5):
        a = Inst(*args)
        a.time("  instance %d" % i)
        a.close()
    cm.lib().cudamat_pool_trim()
    print("E3b: re-created after a trim (fresh driver allocations)")
    for i in range(4):
        a = Inst(*args)
        a.time("  instance %d" % i)
        a.close()
        cm.lib().cudamat_pool_trim()
    print("E1: two alive at once")
    a = Inst(*args)
    b = Inst(*args)
    a.time("  A")
    b.time("  B (created while A lives)")
    a.time("  A again")
    a.close()
    cm.lib().cudamat_pool_trim()
    c = Inst(*args)
    c.time("  C (after A was freed and trimmed)")
    b.time("  B again")
    b.close()
    c.close()


if __name__ == "__main__":
    main()
