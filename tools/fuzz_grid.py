#!/usr/bin/env python3
"""Fuzz of the routes for M x K operators of plain diagonals: the register-tiled k_grid_tile with 2 / 4 / 8 lines per workgroup (knob
grid_tile; round 3), k_grid_diag with 1 / 2 / 4 packs per lane (knob grid_diag), wide
operators through their tall twin in both directions (knob wide_twin = 2) -- random shapes, block lengths (16-byte multiples), four
eltypes, dirty outputs; forward and adjoint (and, round 6, the fused A'A of N x (2 .. 4) grids) bit-exact vs the CPU oracle's loops.      python tools/fuzz_grid.py NCASES [SEED0]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import jets_jl_amd as J
from oracle import jets_oracle as oracle
from tests.helpers import DTYPES, assert_bits_equal, u01

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
J.init(0)
if os.environ.get("GENERAL_BAND"):
    J.tune(general_band=int(os.environ["GENERAL_BAND"]))               # tiles per band of the grid kernels' decode (8 / 16 / 32 / 64)
t0 = time.time()
stats = {"grid": 0, "wide": 0, "tall": 0}
for case in range(seed0, seed0 + ncases):
    rng = np.random.default_rng(91_000 + case)
    dt = DTYPES[rng.integers(len(DTYPES))]
    per16 = 16 // np.dtype(dt).itemsize
    M, K = int(rng.integers(1, 13)), int(rng.integers(1, 13))
    if M == 1 and K == 1:
        K = 2
    n = int(rng.choice([1, 2, 5, 64, 255, 256, 257, 1024, 3000])) * per16
    spc = J.JetSpace(dt, n)
    coeff = [[J.rand(spc, seed=300 + case, stream=i * K + j) for j in range(K)] for i in range(M)]
    A = J.blockop([[J.JopDiagonal(c) for c in row] for row in coeff])
    ops = [[oracle.Block("diag", n, coeff=u01(oracle, dt, 300 + case, i * K + j, n)) for j in range(K)] for i in range(M)]
    hm = [u01(oracle, dt, 1, case * 16 + j, n) for j in range(K)]
    hd = [u01(oracle, dt, 2, case * 16 + i, n) for i in range(M)]
    hmt = [u01(oracle, dt, 3, case * 16 + j, n) for j in range(K)]
    want_d = oracle.block_df(ops, [b.copy() for b in hd], hm)
    want_m = oracle.block_df_adj(ops, [b.copy() for b in hmt], want_d)
    stats["grid" if (M > 1 and K > 1) else ("wide" if M == 1 else "tall")] += 1
    for gd, wt, gt in ((1, 1, 1), (1, 1, (2, 4, 8)[case % 3]), (1, 1, 0), (2, 2, 0), (4, 2, 0), (0, 0, 0)):   # gt: k_grid_tile with automatic / forced lines per workgroup
        J.tune(grid_diag=gd, wide_twin=wt, grid_tile=gt, adj_split=0)
        m = J.from_numpy(np.concatenate(hm), J.domain(A))
        d = J.from_numpy(np.concatenate(hd), J.range(A))
        J.mul_(d, A, m)
        mt = J.from_numpy(np.concatenate(hmt), J.domain(A))
        J.mul_(mt, A.H, d)
        tag = f"case {case}: {np.dtype(dt).name} {M}x{K} n={n} grid_diag={gd} wide_twin={wt} grid_tile={gt}"
        assert_bits_equal(d.to_numpy(), np.concatenate(want_d), tag + " forward")
        assert_bits_equal(mt.to_numpy().ravel(order="F") if K == 1 else mt.to_numpy(), np.concatenate(want_m), tag + " adjoint")
    if M >= 2 and 2 <= K <= 4:                                        # round 6: the fused A'A of N x (2 .. 4) grids (jh_grid_normal.hip): the two stages' bits
        J.tune(grid_diag=1, wide_twin=1, grid_tile=1, adj_split=0)
        t = oracle.block_df(ops, [np.zeros(n, dt) for _ in range(M)], hm)
        want_y = oracle.block_df_adj(ops, [np.zeros(n, dt) for _ in range(K)], t)
        m = J.from_numpy(np.concatenate(hm), J.domain(A))
        y = J.mul_(J.from_numpy(np.concatenate(hmt), J.domain(A)), J.compose(A.H, A), m)      # (a dirty output)
        assert_bits_equal(y.to_numpy(), np.concatenate(want_y), f"case {case}: {np.dtype(dt).name} {M}x{K} n={n} fused A'A")
        stats["normal"] = stats.get("normal", 0) + 1
    J.tune(grid_diag=1, wide_twin=1, grid_tile=1, adj_split=-1)
    J.close(A)
    if (case - seed0 + 1) % 200 == 0:
        print(f"{case - seed0 + 1} cases, {time.time() - t0:.0f} s, {stats}", flush=True)
print(f"fuzz_grid: {ncases} cases from seed {seed0}: all bit-exact under every route; {stats}; {time.time() - t0:.0f} s")
