#!/usr/bin/env python3
"""Why does the adjoint stream at ~6.9 TB/s on 512 rows and ~6.5 TB/s on 1024 rows of 256^3 Float32?  Same slabs, operators
built over the first K rows (views): is it the working set, the allocation, or the launch length?"""
import gc
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jets_jl_amd as J

J.init(0)
edge, N = 256, 1024
n = edge ** 3
blk = J.JetSpace("float32", edge, edge, edge)
coeff = J.rand(J.JetBSpace([blk] * N), seed=1, stream=0)
d = J.rand(J.JetBSpace([blk] * N), seed=3, stream=0)
mt = J.zeros(blk)
m = J.rand(blk, seed=2, stream=0)


def timed(fn, reps=6):
    fn(); fn()
    best, tot = 1e9, 0.0
    for _ in range(reps):
        e0 = J.Event().record(); fn(); e1 = J.Event().record()
        t = e0.elapsed_ms(e1); best = min(best, t); tot += t
    return best, tot / reps


for K, first in ((1024, 0), (512, 0), (512, 512), (256, 0), (256, 768), (768, 0), (896, 0), (640, 0)):
    A = J.blockop([[J.JopDiagonal(c)] for c in coeff.arrays[first:first + K]])
    # range vector over the same K blocks of d: a wrapped view of the slab
    from jets_jl_amd import arrays as _a
    import ctypes as C
    from jets_jl_amd._ffi import lib, check
    h = C.c_void_p()
    check(lib.jh_bvec_view(d.handle, first, K, C.byref(h)))
    dK = _a.BlockArray(h, [blk] * K, d.dtype, owner=d)
    nb = (2 * K * n + n) * 4
    ba, aa = timed(lambda: J.mul_(mt, A.H, dK))
    bf, af = timed(lambda: J.mul_(dK, A, m))
    print(f"rows [{first:4d}, {first + K:4d})  adjoint min {ba:7.3f} avg {aa:7.3f} ms  {nb / ba / 1e6:7.1f} GB/s | forward min {bf:7.3f} avg {af:7.3f} ms  {nb / bf / 1e6:7.1f} GB/s", flush=True)
    J.close(A)
    del A, dK
    gc.collect()
