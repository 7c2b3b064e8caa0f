#!/usr/bin/env python3
"""For tools/prof_any.sh: the forward of a tall operator of ODD blocks (diagonals in one slab) on element-indexed packs (k_tall_diag_fwd<MIXED>, fwd_anchor = 0)
and on lanes anchored to each row's own 16-byte grid (k_tall_fwd_anchored, round 6) -- one profile shows both kernels' time and HBM traffic.
    TAG=anchor255 REGEX='k_tall_(diag_fwd|fwd_anchored)' bash tools/prof_any.sh tools/prof_anchor.py 256 255"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jets_jl_amd as J

J.init(0)
nrow, e = int(sys.argv[1]), int(sys.argv[2])
spc = J.JetSpace("float32", e, e, e)
n = e ** 3
diags = J.rand(J.JetBSpace([spc] * nrow), seed=1, stream=0).arrays
A = J.blockop([[J.JopDiagonal(g)] for g in diags])
m = J.rand(spc, seed=2, stream=0)
d = J.rand(J.range(A), seed=3, stream=0)
by = (2 * nrow + 1) * n * 4
print(f"ALGO k_tall_(diag_fwd|fwd_anchored) {by}")
for k in (0, -1, 0, -1):
    J.tune(fwd_anchor=k)
    J.synchronize()
    e0 = J.Event().record()
    for _ in range(6):
        J.mul_(d, A, m)
    e1 = J.Event().record()
    ms = e0.elapsed_ms(e1) / 6
    print(f"{nrow} x {e}^3 Float32 forward, fwd_anchor = {k:2d}: {ms:8.3f} ms  {by / ms / 1e6:8.1f} GB/s", flush=True)
