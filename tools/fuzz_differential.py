#!/usr/bin/env python3
"""One-off fuzz campaign: the randomized differential test of tests/test_gpu_random_differential.py with wider draws
(up to 40 x 6 blocks, block lengths up to 70 001, tall operators with hundreds of rows) and many more seeds, HIP path vs
the CPU oracle bit for bit (adj_split=0: the ordered walk) -- plus, for tall all-diagonal draws, the fused normal operator,
the ranged adjoint and the split-row walk against a wide host sum.

    python tools/fuzz_differential.py NCASES [SEED0]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import jets_jl_amd as J
from oracle import jets_oracle as oracle
from tests.helpers import DTYPES, assert_bits_equal, u01
from tests.test_gpu_random_differential import KINDS, _build, _split

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 500
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
J.init(0)
if os.environ.get("GENERAL_BAND"):
    J.tune(general_band=int(os.environ["GENERAL_BAND"]))               # tiles per band of the general kernels' decode (8 / 16 / 32 / 64)
POOL = [0, 1, 2, 3, 4, 5, 7, 8, 12, 16, 31, 64, 100, 256, 257, 1000, 1024, 4100, 16384, 70001]


def draw(rng):
    dt = DTYPES[rng.integers(len(DTYPES))]
    shape = rng.random()
    if shape < 0.35:
        nrow, ncol = int(rng.integers(1, 41)), 1                       # tall
    elif shape < 0.5:
        nrow, ncol = 1, int(rng.integers(1, 7))                        # wide
    else:
        nrow, ncol = int(rng.integers(1, 7)), int(rng.integers(1, 7))
    if rng.random() < 0.6:
        n = int(rng.choice(POOL[1:]))
        lens = [n] * max(nrow, ncol)
    else:
        lens = [int(rng.choice(POOL[:17])) for _ in range(max(nrow, ncol))]
    len_r, len_c = lens[:nrow], lens[:ncol]
    mostly = rng.random()
    kinds = []
    for i in range(nrow):
        row = []
        for j in range(ncol):
            if len_r[i] != len_c[j]:
                row.append("zero")
            elif mostly < 0.3:
                row.append("diag")                                     # all-diagonal: the tall fast path when ncol == 1
            elif mostly < 0.4:
                row.append("diag" if rng.random() < 0.8 else ["identity", "scale"][rng.integers(2)])   # densified scalar rows
            else:
                row.append(KINDS[rng.integers(len(KINDS))] if rng.random() < 0.85 else "zero")
        kinds.append(row)
    return dt, nrow, ncol, len_r, len_c, kinds


def wide_err(a, b):
    a, b = np.asarray(a, dtype=np.clongdouble).ravel(), np.asarray(b, dtype=np.clongdouble).ravel()
    den = np.linalg.norm(np.abs(b).astype(np.longdouble))
    return float(np.linalg.norm(np.abs(a - b).astype(np.longdouble)) / (den if den else 1.0))


t0 = time.time()
done = fused = split = 0
if os.environ.get("FUZZ_WIDE_TWIN"):                        # e.g. 2: wide operators' forward through the tall twin at every block size
    J.tune(wide_twin=int(os.environ["FUZZ_WIDE_TWIN"]))
for case in range(seed0, seed0 + ncases):
    rng = np.random.default_rng(777_000 + case)
    dt, nrow, ncol, len_r, len_c, kinds = draw(rng)
    while sum(len_r) == 0 or sum(len_c) == 0:
        dt, nrow, ncol, len_r, len_c, kinds = draw(rng)
    J.tune(adj_split=0)
    A, ops = _build(J, oracle, dt, len_r, len_c, kinds, seed=500 + case)
    tag = f"case {case}: {np.dtype(dt).name} {nrow}x{ncol} rows={len_r[:6]} cols={len_c[:6]} kinds={[r[:6] for r in kinds[:6]]}"
    NR, NC = sum(len_r), sum(len_c)
    try:
        m = J.rand(J.domain(A), seed=1, stream=case)
        d = J.rand(J.range(A), seed=2, stream=case)
        hm, hd = u01(oracle, dt, 1, case, NC), u01(oracle, dt, 2, case, NR)
        J.mul_(d, A, m)
        ref_d = oracle.block_df(ops, _split(hd, len_r), _split(hm, len_c))
        assert_bits_equal(d.to_numpy(), np.concatenate(ref_d), "forward, " + tag)
        dd = J.rand(J.range(A), seed=3, stream=case)
        mt = J.rand(J.domain(A), seed=4, stream=case)
        hdd, hmt = u01(oracle, dt, 3, case, NR), u01(oracle, dt, 4, case, NC)
        J.mul_(mt, A.H, dd)
        ref_m = oracle.block_df_adj(ops, _split(hmt, len_c), _split(hdd, len_r))
        assert_bits_equal(mt.to_numpy().ravel(order="F"), np.concatenate(ref_m), "adjoint, " + tag)
        # composite A'A: fused where the operator allows, chained otherwise -- either way the oracle's chain, bitwise
        y = J.mul(A.H @ A, m)
        tmp = oracle.block_df(ops, [np.zeros(n, dtype=dt) for n in len_r], _split(hm, len_c))
        ref_y = oracle.block_df_adj(ops, [np.zeros(n, dtype=dt) for n in len_c], tmp)
        assert_bits_equal(y.to_numpy().ravel(order="F"), np.concatenate(ref_y), "A'A, " + tag)
        fused += 1
        # every operator again with the rows / columns cut into k parts (split walk, tall and general kernels): tolerance parity
        if max(nrow, ncol) >= 4:
            J.tune(adj_split=int(rng.integers(2, 9)))
            d2 = J.rand(J.range(A), seed=2, stream=case)
            J.mul_(d2, A, m)
            pf = J.tune_get("last_adj_parts")
            mt2 = J.rand(J.domain(A), seed=4, stream=case)
            J.mul_(mt2, A.H, dd)
            pa = J.tune_get("last_adj_parts")
            single = np.dtype(dt).itemsize // (2 if np.dtype(dt).kind == "c" else 1) == 4
            tol = 2e-5 if single else 1e-13                      # against the ORDERED sum, whose own rounding error is in there
            e1 = wide_err(d2.to_numpy(), np.concatenate(ref_d))
            e2 = wide_err(mt2.to_numpy().ravel(order="F"), np.concatenate(ref_m))
            assert e1 < tol and e2 < tol, f"split walk: forward {e1:.2e} (parts {pf}), adjoint {e2:.2e} (parts {pa}), " + tag
            gsplit = globals().get("gsplit", 0) + (pf > 1) + (pa > 1)
            J.tune(adj_split=0)
        if ncol == 1 and nrow >= 4 and all(k == "diag" for r in kinds for k in r) and len_r[0] > 0:
            J.tune(adj_split=int(rng.integers(2, 9)))
            J.mul_(mt, A.H, dd)
            if J.tune_get("last_adj_parts") > 1:
                split += 1
                ha = np.stack([o[0].coeff for o in ops]).astype(np.clongdouble)
                truth = np.sum(np.conj(ha) * np.stack(_split(hdd, len_r)).astype(np.clongdouble), axis=0)
                tol = 1e-6 if np.dtype(dt).itemsize // (2 if np.dtype(dt).kind == "c" else 1) == 4 else 1e-14
                e = wide_err(mt.to_numpy().ravel(order="F"), truth)
                assert e < tol, f"split adjoint rel err {e:.2e}, " + tag
    except Exception as e:
        print("FAIL", tag)
        print(repr(e)[:2000])
        raise SystemExit(1)
    done += 1
    if done % 100 == 0:
        print(f"{done} cases ok ({fused} composites, {split} split-row checks), {time.time() - t0:.0f} s", flush=True)
J.tune(adj_split=-1)
print(f"fuzz: {done} cases, all bit-exact vs the oracle ({fused} composites, {split} split-row checks vs a wide sum, {globals().get('gsplit', 0)} split-walk launches within tolerance of the ordered result), {time.time() - t0:.0f} s")
