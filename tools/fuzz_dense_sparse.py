#!/usr/bin/env python3
"""Fuzz of the list route for block-SPARSE operators of dense children (k_gemv_rows_list / k_gemv_cols_list + the combine over step lists): random M x K
grids up to 30 x 30, a random 3 ... 40 % of the blocks non-zero -- dense, adjointed dense, and (where row and column lengths agree) diagonal / identity --,
ragged row / column lengths incl. empty ones, four eltypes, dirty outputs; forward and adjoint within 2e-6 / 1e-13 of the CPU oracle's loops under the
list route (both lane layouts), the grid route of round 3 and the per-block loop; with columns in order and no adjointed child the forward bit for bit.
    python tools/fuzz_dense_sparse.py NCASES [SEED0]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import jets_jl_amd as J
from oracle import jets_oracle as oracle
from tests.helpers import DTYPES, assert_bits_equal, u01
from tests.test_gpu_dense_lists import _build, _err, _tol

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
J.init(0)
t0 = time.time()
POOL = [0, 1, 3, 4, 8, 20, 33, 64, 100, 130, 256]
stats = {"bits": 0, "children": 0}
for case in range(seed0, seed0 + ncases):
    rng = np.random.default_rng(55_000 + case)
    dt = DTYPES[rng.integers(len(DTYPES))]
    M, K = int(rng.integers(2, 31)), int(rng.integers(2, 31))
    if rng.random() < 0.5:
        n = int(rng.choice(POOL[1:]))
        row_len, col_len = [n] * M, [n] * K
    else:
        row_len, col_len = [int(rng.choice(POOL)) for _ in range(M)], [int(rng.choice(POOL)) for _ in range(K)]
    if sum(row_len) == 0 or sum(col_len) == 0:
        row_len[0] = col_len[0] = 8
    fill = float(rng.choice([0.03, 0.08, 0.2, 0.4]))
    kinds, nd, nadj = [], 0, 0
    for i in range(M):
        row = []
        for j in range(K):
            k = "zero"
            if rng.random() < fill or (i == j and rng.random() < 0.7):
                k = "dense" if rng.random() < 0.75 else "dense_adj"
                if row_len[i] == col_len[j] and rng.random() < 0.15:
                    k = "diag" if rng.random() < 0.5 else "id"
                nd += k.startswith("dense"); nadj += k == "dense_adj"
            row.append(k)
        kinds.append(row)
    if nd == 0:
        kinds[0][0] = "dense"; nd = 1
    A, ops = _build(J, oracle, dt, kinds, row_len, col_len, seed=2000 + case)
    stats["children"] += nd
    hm = [u01(oracle, dt, 1, case * 64 + j, col_len[j]) for j in range(K)]
    hd = [u01(oracle, dt, 2, case * 64 + i, row_len[i]) for i in range(M)]
    hmt = [u01(oracle, dt, 3, case * 64 + j, col_len[j]) for j in range(K)]
    want_d = oracle.block_df(ops, [b.copy() for b in hd], hm)
    want_m = oracle.block_df_adj(ops, [b.copy() for b in hmt], want_d)
    wd, wm = np.concatenate(want_d), np.concatenate(want_m)
    tag = f"case {case}: {np.dtype(dt).name} {M}x{K} rows={row_len[:6]} cols={col_len[:6]} fill={fill} dense={nd} adjointed={nadj}"
    for route in ("lists", "lists-combine", "lists-in-order", "grid", "loop"):    # lists-combine: never the one-launch direct mode of block-diagonal operators
        J.tune(small_loop_max_kib=0, dense_list=0 if route == "grid" else 1, dense_list_split=0 if route == "lists-in-order" else 1,
               dense_mixed=0 if route == "loop" else 1, small_loop=0 if route == "loop" else 1, dense_direct=0 if route == "lists-combine" else 1)
        d = J.from_numpy(np.concatenate(hd), J.range(A))
        J.mul_(d, A, J.from_numpy(np.concatenate(hm), J.domain(A)))
        mt = J.from_numpy(np.concatenate(hmt), J.domain(A))
        J.mul_(mt, A.H, J.from_numpy(wd, J.range(A)))
        assert _err(d.to_numpy(), wd) < _tol(dt), f"forward, {route}, {tag}: {_err(d.to_numpy(), wd)}"
        assert _err(mt.to_numpy(), wm) < _tol(dt), f"adjoint, {route}, {tag}: {_err(mt.to_numpy(), wm)}"
        if route == "lists-in-order" and nadj == 0:
            assert_bits_equal(d.to_numpy(), wd, "forward with columns in order, " + tag)
            stats["bits"] += 1
    J.tune(small_loop_max_kib=512, dense_list=1, dense_list_split=1, dense_mixed=1, small_loop=1, dense_direct=1)
    J.close(A)
    if (case - seed0 + 1) % 100 == 0:
        print(f"{case - seed0 + 1} cases, {time.time() - t0:.0f} s, {stats}", flush=True)
print(f"fuzz_dense_sparse: {ncases} cases from seed {seed0} ok under the four routes; {stats}; {time.time() - t0:.0f} s")
