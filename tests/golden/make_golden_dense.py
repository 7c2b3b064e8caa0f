#!/usr/bin/env python3
"""Generates tests/golden/jets_dense_blocks_v1.npz: tall (uniform and ragged), wide and grid block operators of DENSE children
(the reference's JopBaz fixture, test/runtests.jl:27-33, in the shapes of its test sets 720-758), seeded inputs and the CPU
oracle's outputs.  The reference itself cannot run here (Julia), so the expected outputs come from the oracle, which is pinned
against the reference's test identities in tests/test_oracle_pinning.py.

    python tests/golden/make_golden_dense.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import jets_oracle as jo  # noqa: E402

CASES = {   # tag -> (dtype, row counts, column counts): block (i, j) is rows[i] x cols[j]
    "tall_f32": (np.float32, [8] * 6, [12]),
    "tall_c64": (np.complex128, [5] * 4, [5]),
    "ragged_f64": (np.float64, [4, 9, 2, 16, 7], [6]),
    "wide_f64": (np.float64, [10], [4] * 5),
    "wide_c32": (np.complex64, [8], [8] * 3),
    "grid_f32": (np.float32, [6] * 3, [4] * 2),
}


def build(tag):
    dt, rows, cols = CASES[tag]
    seed = 7000 + sorted(CASES).index(tag)
    mats = [[np.asfortranarray(jo.rng_u01(dt, seed, 100 * i + j, 0, rows[i] * cols[j]).reshape((rows[i], cols[j]), order="F"))
             for j in range(len(cols))] for i in range(len(rows))]
    ops = [[jo.Block("dense", rows[i], cols[j], coeff=mats[i][j]) for j in range(len(cols))] for i in range(len(rows))]
    m = jo.rng_u01(dt, seed, 9001, 0, sum(cols))
    d = jo.rng_u01(dt, seed, 9002, 0, sum(rows))
    d0 = jo.rng_u01(dt, seed, 9003, 0, sum(rows))
    return dt, rows, cols, mats, ops, m, d, d0


def split(v, lens):
    off = np.cumsum([0] + list(lens))
    return [np.ascontiguousarray(v[off[k]:off[k + 1]]) for k in range(len(lens))]


def main():
    out = {}
    for tag in CASES:
        dt, rows, cols, mats, ops, m, d, d0 = build(tag)
        out[f"{tag}_A"] = np.concatenate([mats[i][j].ravel(order="F") for i in range(len(rows)) for j in range(len(cols))])
        out[f"{tag}_m"], out[f"{tag}_d"], out[f"{tag}_d0"] = m, d, d0
        out[f"{tag}_fwd_dirty"] = np.concatenate(jo.block_df(ops, split(d0.copy(), rows), split(m, cols)))      # accumulates into d as found when ncol > 1
        out[f"{tag}_adj"] = np.concatenate(jo.block_df_adj(ops, [np.zeros(c, dtype=dt) for c in cols], split(d, rows)))
    np.savez_compressed(os.path.join(HERE, "jets_dense_blocks_v1.npz"), **out)
    print("wrote", os.path.join(HERE, "jets_dense_blocks_v1.npz"), {k: v.shape for k, v in out.items() if k.endswith("_adj")})


if __name__ == "__main__":
    main()
