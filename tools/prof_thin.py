#!/usr/bin/env python3
"""For tools/prof_any.sh: THIN tall operators -- 262144 rows of 513 Float32 (traces rather than volumes: adjoint and fused A'A through the
split-row walk + fold) and a 256-row operator of 2 MiB rows with an identity row in four (the MIXED fused A'A).
    TAG=thin_r06 REGEX='k_tall_diag_adj|k_fold_parts' bash tools/prof_any.sh tools/prof_thin.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jets_jl_amd as J

J.init(0)


def timed(fn, reps=10):
    fn(); fn(); J.synchronize()
    e0 = J.Event().record()
    for _ in range(reps):
        fn()
    e1 = J.Event().record()
    return e0.elapsed_ms(e1) / reps


for nrow, n, mixed in ((262144, 513, False), (262144, 512, False), (256, 524288, True), (256, 524288, False)):
    spc = J.JetSpace("float32", n)
    slab = J.rand(J.JetBSpace([spc] * nrow), seed=1, stream=0).arrays
    A = J.blockop([[J.JopDiagonal(slab[i]) if (not mixed or i % 4) else J.JopIdentity(spc)] for i in range(nrow)])
    m = J.rand(spc, seed=2, stream=0); d = J.rand(J.range(A), seed=3, stream=0); mt = J.zeros(spc)
    N = J.compose(A.H, A)
    ncoef = nrow if not mixed else nrow - (nrow + 3) // 4
    ta = timed(lambda: J.mul_(mt, A.H, d)); parts_a = J.tune_get("last_adj_parts")
    tn = timed(lambda: J.mul_(mt, N, m)); parts_n = J.tune_get("last_adj_parts")
    print(f"{nrow:7d} x {n:7d} {'MIXED' if mixed else 'diag '}: adjoint {ta:7.3f} ms {(ncoef + nrow + 1) * n * 4 / ta / 1e6:6.0f} GB/s ({parts_a} parts) | "
          f"A'A {tn:7.3f} ms {(ncoef + 2) * n * 4 / tn / 1e6:6.0f} GB/s ({parts_n} parts)", flush=True)
    J.close(A); del A, slab, d
