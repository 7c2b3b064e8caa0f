#!/usr/bin/env python3
"""Tall adjoint and fused A'A walked in several launches over row ranges (knob adj_rows_per_launch; same bits as one launch):
is a 128 GiB operator faster in pieces?  1024 x 256^3 Float32, interleaved rounds."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jets_jl_amd as J

J.init(0)
edge, N = 256, int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n = edge ** 3
blk = J.JetSpace(np.float32, edge, edge, edge)
coeff = J.rand(J.JetBSpace([blk] * N), seed=1, stream=0)
A = J.blockop([[J.JopDiagonal(c)] for c in coeff.arrays])
m = J.rand(blk, seed=2, stream=0)
d = J.rand(J.range(A), seed=3, stream=0)
mt, ref = J.zeros(blk), J.zeros(blk)
J.mul_(ref, A.H, d)
nb = (2 * N * n + n) * 4
res = {}
for rnd in range(3):
    for rows in (0, 512, 256, 128, 64, 32):
        if rows >= N:
            continue
        J.tune(adj_rows_per_launch=rows)
        J.mul_(d, A, m)                         # the solver alternates
        J.mul_(mt, A.H, d)
        e0 = J.Event().record()
        J.mul_(mt, A.H, d)
        e1 = J.Event().record()
        res.setdefault(rows, []).append(e0.elapsed_ms(e1))
for rows, ts in res.items():
    print(f"{N} x {edge}^3 adjoint, rows per launch {rows or N:5d}: min {min(ts):7.3f} ms  med {sorted(ts)[1]:7.3f} ms  {nb / min(ts) / 1e6:7.1f} GB/s", flush=True)
J.tune(adj_rows_per_launch=256)
J.mul_(d, A, m)
J.mul_(ref, A.H, d)
J.tune(adj_rows_per_launch=0)
J.mul_(mt, A.H, d)
print("bit-identical:", np.array_equal(mt.to_numpy(), ref.to_numpy()))
