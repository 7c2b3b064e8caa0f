#!/usr/bin/env python3
"""The fused normal operator of N x K grids of diagonals (jh_grid_normal.hip) for tools/prof_any.sh: prints the ALGO lines its summary needs.

    TAG=grid_normal_r06 REGEX='k_grid_normal' bash tools/prof_any.sh tools/prof_grid_normal.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jets_jl_amd as J

J.init(0)
reps = 10
cases = [(64, 4, 256), (128, 2, 256), (96, 3, 256)]
for N, K, e in cases:
    print(rf"ALGO k_grid_normal<float,\s1,\s4,\s{K}, {(N * K + 2 * K) * e ** 3 * 4}")
for N, K, e in cases:
    spc = J.JetSpace(np.float32, e, e, e)
    coeff = J.rand(J.JetBSpace([spc] * (N * K)), seed=1, stream=0)
    A = J.blockop([[J.JopDiagonal(coeff.arrays[i * K + j]) for j in range(K)] for i in range(N)])
    m, y = J.rand(J.domain(A), seed=2, stream=0), J.zeros(J.domain(A))
    NA = J.compose(A.H, A)
    J.mul_(y, NA, m)
    J.synchronize()
    e0 = J.Event().record()
    for _ in range(reps):
        J.mul_(y, NA, m)
    e1 = J.Event().record()
    ms = e0.elapsed_ms(e1) / reps
    nbytes = (N * K + 2 * K) * e ** 3 * 4
    print(f"A' o A on {N} x {K} of {e}^3 Float32: {ms:8.3f} ms  {nbytes / ms / 1e6:8.1f} GB/s over {nbytes} algorithmic bytes", flush=True)
    J.close(A)
    del coeff, A, m, y, NA
