#!/usr/bin/env python3
"""Operators that mix BIG dense children (beyond the one-launch loop's 256 KiB) with diagonal / zero blocks: the per-block loop (one child launch
+ one accumulate launch per non-zero block) against round 3's per-column batches + one combine launch (knob dense_mixed).
    python tools/bench_dense_mixed.py [M] [K] [N]      M x K grid of N x N Float32 blocks, dense on a checkerboard, diagonals elsewhere"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import jets_jl_amd as J

M = int(sys.argv[1]) if len(sys.argv) > 1 else 8
K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
N = int(sys.argv[3]) if len(sys.argv) > 3 else 384
J.init(0)
spc = J.JetSpace("float32", N)
mat = J.JetSpace("float32", N, N)
rows, ndense = [], 0
for i in range(M):
    row = []
    for j in range(K):
        if (i + j) % 2 == 0:
            row.append(J.JopDense(J.rand(mat, seed=7, stream=i * K + j)))
            ndense += 1
        elif (i + j) % 5 == 0:
            row.append(J.JopZeroBlock(spc, spc))
        else:
            row.append(J.JopDiagonal(J.rand(spc, seed=8, stream=i * K + j)))
    rows.append(row)
A = J.blockop(rows)
m = J.rand(J.domain(A), seed=2, stream=0)
d = J.zeros(J.range(A))
mt = J.zeros(J.domain(A))


def timed(fn, reps=20, warm=5):
    for _ in range(warm):
        fn()
    J.synchronize()
    e0 = J.Event().record()
    for _ in range(reps):
        fn()
    e1 = J.Event().record()
    return e0.elapsed_ms(e1) / reps * 1e3


mat_bytes = ndense * N * N * 4
for knob in (1, 0, 1, 0):
    J.tune(dense_mixed=knob, graphs=1)
    tf = timed(lambda: J.mul_(d, A, m))
    lf = J.tune_get("last_launches") if knob else None
    ta = timed(lambda: J.mul_(mt, A.H, d))
    print(f"{M} x {K} grid of {N}^2 Float32 blocks ({ndense} dense = {mat_bytes / 2**20:.0f} MiB of matrices), dense_mixed={knob}: forward {tf:8.1f} us ({mat_bytes / tf / 1e6:6.2f} TB/s of matrices)"
          f"{f' in {lf} launches' if lf else ''} | adjoint {ta:8.1f} us ({mat_bytes / ta / 1e6:6.2f} TB/s)", flush=True)
