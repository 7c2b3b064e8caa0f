#!/usr/bin/env python3
"""Tall operators with ONE identity row among the diagonals on rows of 1-4 MiB -- the adjoint and the fused A'A, which run the chain kernels with empty stage lists
since round 6 -- and a regularised 64 x 4 grid's fused A'A, for tools/prof_any.sh: prints the ALGO lines its summary needs.

    TAG=mixed_mid_r06 REGEX='k_chain_adj|k_grid_normal|k_fold_parts' bash tools/prof_any.sh tools/prof_mixed_mid.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jets_jl_amd as J

J.init(0)
reps = 20


def timed(tag, fn, nbytes):
    fn(); J.synchronize()
    e0 = J.Event().record()
    for _ in range(reps):
        fn()
    e1 = J.Event().record()
    ms = e0.elapsed_ms(e1) / reps
    print(f"{tag:70s} {ms:8.3f} ms  {nbytes / ms / 1e6:8.1f} GB/s", flush=True)


for nrow, n in ((int(os.environ.get("NROW", "256")), int(os.environ.get("NLEN", "524288"))),):
    spc = J.JetSpace(np.float32, n)
    R = J.JetBSpace([spc] * nrow)
    rows = [[J.JopDiagonal(c)] for c in J.rand(R, seed=1, stream=0).arrays]
    rows[3] = [J.JopIdentity(spc)]
    A = J.blockop(rows)
    m, y, d = J.rand(spc, seed=2, stream=0), J.zeros(spc), J.rand(R, seed=3, stream=0)
    NA = J.compose(A.H, A)
    Nn = (nrow - 1) * n * 4
    print(f"ALGO k_chain_adj<float,\\s1,\\s4,\\s1,\\s4,\\s(true|false),\\s1,\\s256,\\s0> {Nn + 2 * n * 4}")
    print(f"ALGO k_chain_adj<float,\\s1,\\s4,\\s1,\\s4,\\s(true|false),\\s0,\\s256,\\s0> {Nn + nrow * n * 4 + n * 4}")
    timed(f"{nrow} x {n} Float32, one identity row: fused A'A", lambda: J.mul_(y, NA, m), Nn + 2 * n * 4)
    timed(f"{nrow} x {n} Float32, one identity row: adjoint", lambda: J.mul_(y, A.H, d), Nn + nrow * n * 4 + n * 4)
    J.close(A)
    del A, R, m, y, d, NA, rows
N, K, e = 64, 4, 256
spc = J.JetSpace(np.float32, e, e, e)
coeff = J.rand(J.JetBSpace([spc] * (N * K)), seed=1, stream=0)
rows = [[J.JopDiagonal(coeff.arrays[i * K + j]) for j in range(K)] for i in range(N)]
lam = lambda: J.JopLn(dom=spc, rng=spc, df=J.constdiag_df, df_adj=J.constdiag_df_adj, s={"a": 0.5})
rows += [[lam() if j == k else J.JopZeroBlock(spc, spc) for j in range(K)] for k in range(K)]
A = J.blockop(rows)
m, y = J.rand(J.domain(A), seed=2, stream=0), J.zeros(J.domain(A))
NA = J.compose(A.H, A)
nb = (N * K + 2 * K) * e ** 3 * 4
print(f"ALGO k_grid_normal_mixed<float,\\s1,\\s4,\\s4, {nb}")
timed(f"{N} + {K} regularisation rows x {K} of {e}^3 Float32: fused A'A", lambda: J.mul_(y, NA, m), nb)
