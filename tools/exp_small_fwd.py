#!/usr/bin/env python3
"""The tall all-diagonal forward BELOW the lazily measured regime (operators under 8 GiB streamed, or fewer than 64 rows): the size-based default shape
against one pack per lane in column bands.   python tools/exp_small_fwd.py [ROWSxEDGE ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jets_jl_amd as J

J.init(0)
J.tune(autotune=0)
for nrow, edge in [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or ((1024, 64), (64, 128), (16, 256), (32, 256), (256, 128), (4096, 32)):
    spc = J.JetSpace("float32", edge, edge, edge)
    n = edge ** 3
    coeff = J.rand(J.JetBSpace([spc] * nrow), seed=1, stream=0)
    A = J.blockop([[J.JopDiagonal(c)] for c in coeff.arrays])
    m = J.rand(spc, seed=2, stream=0)
    d = J.zeros(J.range(A))

    def timed():
        for _ in range(3):
            J.mul_(d, A, m)
        ts = []
        for _ in range(8):
            e0 = J.Event().record()
            J.mul_(d, A, m)
            e1 = J.Event().record()
            ts.append(e0.elapsed_ms(e1))
        return min(ts)

    b = (2 * nrow + 1) * n * 4
    for name, kw in (("default", {}), ("256 x 1, 2 rows, bands of 32", dict(fwd_wg=256, fwd_unroll=1, fwd_group=2, fwd_ctiles=32)),
                     ("256 x 1, 1 row, bands of 32", dict(fwd_wg=256, fwd_unroll=1, fwd_group=1, fwd_ctiles=32)),
                     ("256 x 1, 2 rows, bands of 64", dict(fwd_wg=256, fwd_unroll=1, fwd_group=2, fwd_ctiles=64)),
                     ("256 x 1, 2 rows, all rows concurrent", dict(fwd_wg=256, fwd_unroll=1, fwd_group=2, fwd_order=1)), ("default", {})):
        J.tune(fwd_wg=0, fwd_unroll=0, fwd_group=0, fwd_order=-1, fwd_ctiles=-1)
        J.tune(**kw)
        t = timed()
        print(f"{nrow} x {edge}^3 all-diagonal forward, {name:38s}: {t:7.3f} ms {b / t / 1e6:7.1f} GB/s  (rows/wg {J.tune_get('last_fwd_rows_per_wg')}, walk {J.tune_get('last_fwd_walk')})", flush=True)
    J.tune(fwd_wg=0, fwd_unroll=0, fwd_group=0, fwd_order=-1, fwd_ctiles=-1)
    J.close(A)
