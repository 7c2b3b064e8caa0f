#!/usr/bin/env python3
"""For tools/prof_any.sh: forward and adjoint of a tall operator of ODD blocks with streaming (ua_nt = 1) and temporal (ua_nt = 0) loads -- the two
policies are different instantiations (NT = true / false), so one profile shows both kernels' HBM traffic against the algorithmic bytes.
    TAG=ua255 REGEX='k_tall_diag_(fwd|adj)' bash tools/prof_any.sh tools/prof_unaligned.py 256 255"""
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
mt = J.zeros(spc)
by = (2 * nrow + 1) * n * 4
print(f"ALGO k_tall_diag_(fwd|adj) {by}")
for k in (1, 0):
    J.tune(ua_nt=k)
    for _ in range(6):
        J.mul_(d, A, m)
    for _ in range(6):
        J.mul_(mt, A.H, d)
J.synchronize()
print(f"{nrow} x {e}^3 Float32, diagonals in one slab: 6 forwards + 6 adjoints with ua_nt = 1 (NT = true), then with ua_nt = 0 (NT = false)")
