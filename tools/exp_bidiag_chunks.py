#!/usr/bin/env python3
"""Cost of running the one-pass step in chunks of the domain (no exchange), 1024 x 256^3 and 128 x 256^3."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jets_jl_amd as J
from jets_jl_amd._ffi import lib, check
from jets_jl_amd import jetblock as _blk

J.init(0)
edge = 256
n = edge ** 3
for nblocks in (1024, 128):
    blk = J.JetSpace("float32", edge, edge, edge)
    coeff = J.rand(J.JetBSpace([blk] * nblocks), seed=1, stream=0)
    A = J.blockop([[J.JopDiagonal(c)] for c in coeff.arrays])
    nat = _blk._native_op(A.jet.s["_native"], A.jet.s["ops"], A.jet.rng.eltype())
    u, v, w = J.rand(J.range(A), seed=3, stream=0), J.rand(J.domain(A), seed=2, stream=0), J.zeros(J.domain(A))
    out = C.c_double(0)
    b3 = (3 * nblocks * n + 2 * n) * 4

    def run(nchunks, sync):
        step = -(-n // nchunks)
        step = -(-step // 16384) * 16384
        lo = 0
        while lo < n:
            cnt = min(step, n - lo)
            check(lib.jh_blockop_bidiag_step_range(nat.handle, u.handle, v.handle, w.handle, 1.0, -0.5, lo, cnt, C.byref(out) if sync else None))
            lo += cnt

    for sync in (True, False):
        for nchunks in (1, 2, 4, 8):
            run(nchunks, sync); run(nchunks, sync)
            best = 1e9
            for _ in range(4):
                e0 = J.Event().record(); run(nchunks, sync); e1 = J.Event().record()
                best = min(best, e0.elapsed_ms(e1))
            print(f"{nblocks} x {edge}^3  {nchunks} chunk(s), normsq readback per chunk: {sync!s:5s}  {best:8.3f} ms  {b3 / best / 1e6:7.1f} GB/s", flush=True)
    J.close(A); del A, coeff, u, v, w, nat
    import gc; gc.collect()
