#!/usr/bin/env python3
"""The fused forward update d = alpha * (A m) (+ beta * d) -- the one pass of `(a * A) * m` -- in its two walks against column bands.
    python tools/exp_update_fwd.py [ROWSxEDGE ...]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jets_jl_amd as J
from jets_jl_amd import jetblock as _blk
from jets_jl_amd._ffi import check, lib

J.init(0)
J.tune(autotune=0)
for nrow, edge in [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or ((128, 256), (256, 256), (1024, 128), (1024, 256)):
    spc = J.JetSpace("float32", edge, edge, edge)
    n = edge ** 3
    coeff = J.rand(J.JetBSpace([spc] * nrow), seed=1, stream=0)
    A = J.blockop([[J.JopDiagonal(c)] for c in coeff.arrays])
    m = J.rand(spc, seed=2, stream=0)
    d = J.zeros(J.range(A))
    J.mul_(d, A, m)
    nat = _blk._tall_native(A)

    def timed(beta):
        fn = lambda: check(lib.jh_blockop_mul_axpby(nat.handle, d.handle, m.handle, 0.75, beta, None))
        for _ in range(3):
            fn()
        ts = []
        for _ in range(8):
            e0 = J.Event().record()
            fn()
            e1 = J.Event().record()
            ts.append(e0.elapsed_ms(e1))
        return min(ts)

    for beta in (0.0, 0.5):
        b = ((2 if beta == 0 else 3) * nrow + 1) * n * 4
        for name, kw in (("default", {}), ("256 x 4, order 1", dict(fwd_order=1)), ("256 x 1, 2 rows, bands of 32", dict(fwd_wg=256, fwd_unroll=1, fwd_group=2, fwd_ctiles=32)),
                         ("256 x 1, 1 row, bands of 32", dict(fwd_wg=256, fwd_unroll=1, fwd_group=1, fwd_ctiles=32)),
                         ("256 x 1, 2 rows, bands of 64", dict(fwd_wg=256, fwd_unroll=1, fwd_group=2, fwd_ctiles=64)),
                         ("256 x 4, 2 rows, bands of 8", dict(fwd_wg=256, fwd_unroll=4, fwd_group=2, fwd_ctiles=8)), ("default", {})):
            J.tune(fwd_wg=0, fwd_unroll=0, fwd_group=0, fwd_order=-1, fwd_ctiles=-1)
            J.tune(**kw)
            t = timed(beta)
            print(f"{nrow} x {edge}^3 update beta={beta}, {name:30s}: {t:7.3f} ms {b / t / 1e6:7.1f} GB/s", flush=True)
    J.tune(fwd_wg=0, fwd_unroll=0, fwd_group=0, fwd_order=-1, fwd_ctiles=-1)
    J.close(A)
