#!/usr/bin/env python3
"""Fuzz of the step-list walk of the register-tiled general kernel (k_general_tile LIST): random M x K grids (up to 40 x 40) of equal blocks with random
sparsity (structured: block-diagonal / banded / arrow / lower-triangular, or a random fill of 2 ... 70 %), kinds drawn from every elementwise kind, four
eltypes, block lengths that are 16-byte multiples (ragged tiles), dirty outputs; the per-line lists (general_list = 3) and the four-line lists (2) always, by the automatic rule (1) and never
(0), the XCD-aware decode on and off, every band width -- forward and adjoint bit-exact vs the CPU oracle's loops.
    python tools/fuzz_grid_sparse.py NCASES [SEED0]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import jets_jl_amd as J
from oracle import jets_oracle as oracle
from tests.helpers import DTYPES, assert_bits_equal, u01
from tests.test_gpu_random_differential import _build

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
J.init(0)
t0 = time.time()
NAMES = ["identity", "scale", "diag", "diag", "diag_adj"]
stats = {"list_fwd": 0, "list_adj": 0, "auto_fwd": 0, "auto_adj": 0}
for case in range(seed0, seed0 + ncases):
    rng = np.random.default_rng(77_000 + case)
    dt = DTYPES[rng.integers(len(DTYPES))]
    per16 = 16 // np.dtype(dt).itemsize
    M, K = int(rng.integers(2, 41)), int(rng.integers(2, 41))
    n = int(rng.choice([1, 2, 5, 64, 255, 256, 257, 600])) * per16
    pat = ["diag", "band", "arrow", "lower", "rand"][rng.integers(5)]
    fill = float(rng.choice([0.02, 0.05, 0.1, 0.25, 0.5, 0.7]))
    w = int(rng.integers(0, 4))
    kinds = []
    for i in range(M):
        row = []
        for j in range(K):
            on = {"diag": i == j, "band": abs(i - j) <= w, "arrow": i == j or i == 0 or j == 0, "lower": j <= i and rng.random() < 0.6,
                  "rand": rng.random() < fill}[pat]
            row.append(NAMES[rng.integers(len(NAMES))] if on else "zero")
        kinds.append(row)
    A, ops = _build(J, oracle, dt, [n] * M, [n] * K, kinds, seed=900 + case)
    hm = [u01(oracle, dt, 1, case * 64 + j, n) for j in range(K)]
    hd = [u01(oracle, dt, 2, case * 64 + i, n) for i in range(M)]
    hmt = [u01(oracle, dt, 3, case * 64 + j, n) for j in range(K)]
    want_d = oracle.block_df(ops, [b.copy() for b in hd], hm)
    want_m = oracle.block_df_adj(ops, [b.copy() for b in hmt], want_d)
    for gl in (3, 2, 1, 0):
        J.tune(general_list=gl, general_xcd=int(rng.integers(0, 3)), general_band=int(rng.choice([8, 16, 32, 64])), adj_split=0)
        m = J.from_numpy(np.concatenate(hm), J.domain(A))
        d = J.from_numpy(np.concatenate(hd), J.range(A))
        J.mul_(d, A, m)
        lf = J.tune_get("last_general_list")
        mt = J.from_numpy(np.concatenate(hmt), J.domain(A))
        J.mul_(mt, A.H, d)
        la = J.tune_get("last_general_list")
        tag = f"case {case}: {np.dtype(dt).name} {M}x{K} n={n} {pat} general_list={gl}"
        assert_bits_equal(d.to_numpy(), np.concatenate(want_d), tag + " forward")
        assert_bits_equal(mt.to_numpy(), np.concatenate(want_m), tag + " adjoint")
        if gl == 2:
            stats["list_fwd"] += lf; stats["list_adj"] += la
        elif gl == 1:
            stats["auto_fwd"] += int(lf > 0); stats["auto_adj"] += int(la > 0)
    J.tune(general_list=1, general_xcd=1, general_band=8, adj_split=-1)
    J.close(A)
    if (case - seed0 + 1) % 100 == 0:
        print(f"{case - seed0 + 1} cases, {time.time() - t0:.0f} s, {stats}", flush=True)
print(f"fuzz_grid_sparse: {ncases} cases from seed {seed0}: all bit-exact under every route; launches on the lists {stats}; {time.time() - t0:.0f} s")
