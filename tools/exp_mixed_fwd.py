import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
import jets_jl_amd as J
J.init(0)
for nrow, edge in [tuple(int(v) for v in a.split('x')) for a in sys.argv[1:]] or ((256, 256), (1024, 128), (128, 256)):
    spc = J.JetSpace("float32", edge, edge, edge)
    n = edge ** 3
    coeff = J.rand(J.JetBSpace([spc] * nrow), seed=1, stream=0)
    rows = [[J.JopDiagonal(c)] for c in coeff.arrays]
    rows[nrow // 2] = [J.JopIdentity(spc)]
    A = J.blockop(rows)
    m = J.rand(spc, seed=2, stream=0)
    d = J.zeros(J.range(A))
    def timed():
        for _ in range(3): J.mul_(d, A, m)
        ts = []
        for _ in range(8):
            e0 = J.Event().record(); J.mul_(d, A, m); e1 = J.Event().record(); ts.append(e0.elapsed_ms(e1))
        return min(ts)
    b = (2 * nrow - 1) * n * 4
    for order, grp, un in ((-1, 0, 0), (32, 1, 1), (32, 2, 1), (32, 4, 1), (16, 2, 1), (-1, 0, 0)):
        J.tune(fwd_order=order, fwd_group=grp, fwd_unroll=un)
        t = timed()
        print(f"{nrow} x {edge}^3 mixed forward, packs/lane {un or 4} ctiles {order} rows/wg {grp or 4}: {t:7.3f} ms {b / t / 1e6:7.1f} GB/s", flush=True)
    J.tune(fwd_order=-1, fwd_group=0, fwd_unroll=0)
    J.close(A)
