#!/usr/bin/env python3
"""Generates tests/golden/jets_nonlinear_v1.npz: fixtures for the nonlinear block path (JetBlock_f!, block point!, Jacobian)
and for one Golub-Kahan step (forward update followed by the adjoint), produced by the CPU oracle (see make_golden.py for
the provenance note: the reference cannot run here and stores no vectors; the oracle is pinned to its test identities).
Data only.

    python tests/golden/make_golden_nonlinear.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import jets_oracle as jo  # noqa: E402
from tests.golden.make_golden import signed  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "jets_nonlinear_v1.npz")
NL_KINDS = [["square", "diag", "zero"], ["identity", "square", "square"]]


def nl_blocks(dt, n, coeffs, mo):
    rows = []
    for i, row in enumerate(NL_KINDS):
        r = []
        for j, k in enumerate(row):
            if k == "zero":
                r.append(jo.Block("zero", n, n))
            elif k == "identity":
                r.append(jo.Block("identity", n))
            elif k == "square":
                r.append(jo.Block("square", n, coeff=np.ascontiguousarray(mo[j])))
            else:
                r.append(jo.Block("diag", n, coeff=np.ascontiguousarray(coeffs[i, j])))
        rows.append(r)
    return rows


def nonlinear_case(dt, n, tag, out):
    coeffs = np.stack([np.stack([signed(dt, 41, 10 * i + j, n) for j in range(3)]) for i in range(2)])
    mo = np.stack([signed(dt, 42, j, n) for j in range(3)])
    dm = np.stack([signed(dt, 43, j, n) for j in range(3)])
    d0 = np.stack([signed(dt, 44, i, n) for i in range(2)])
    dd = np.stack([signed(dt, 45, i, n) for i in range(2)])
    ops = nl_blocks(dt, n, coeffs, mo)
    f = jo.block_f(ops, [d0[i].copy() for i in range(2)], [np.ascontiguousarray(mo[j]) for j in range(3)])          # dirty d: accumulated into
    jv = jo.block_df(ops, [d0[i].copy() for i in range(2)], [np.ascontiguousarray(dm[j]) for j in range(3)])
    jt = jo.block_df_adj(ops, [np.zeros(n, dtype=dt) for _ in range(3)], [np.ascontiguousarray(dd[i]) for i in range(2)])
    out.update({f"{tag}_coeffs": coeffs, f"{tag}_mo": mo, f"{tag}_dm": dm, f"{tag}_d0": d0, f"{tag}_dd": dd,
                f"{tag}_f": np.stack(f), f"{tag}_jv": np.stack(jv), f"{tag}_jt": np.stack(jt)})


def bidiag_case(dt, nrow, n, tag, out):
    a = np.stack([signed(dt, 51, i, n) for i in range(nrow)])
    v = signed(dt, 52, 0, n)
    u = np.stack([signed(dt, 53, i, n) for i in range(nrow)])
    alpha, beta = 0.75, -1.375
    ops = [[jo.Block("diag", n, coeff=np.ascontiguousarray(a[i]))] for i in range(nrow)]
    tmp = jo.block_df(ops, [np.zeros(n, dtype=dt) for _ in range(nrow)], [v])
    unew = jo.barr_lincomb([np.empty(n, dtype=dt) for _ in range(nrow)], [alpha, beta], [tmp, [np.ascontiguousarray(u[i]) for i in range(nrow)]])
    w = jo.block_df_adj(ops, [np.zeros(n, dtype=dt)], unew)[0]
    out.update({f"{tag}_a": a, f"{tag}_v": v, f"{tag}_u": u, f"{tag}_alpha_beta": np.array([alpha, beta]), f"{tag}_unew": np.stack(unew), f"{tag}_w": w})


def main():
    out = {}
    nonlinear_case(np.float64, 37, "nl_f64", out)
    nonlinear_case(np.complex64, 20, "nl_c32", out)
    bidiag_case(np.float32, 5, 76, "gk_f32", out)
    bidiag_case(np.complex128, 3, 18, "gk_c64", out)
    np.savez_compressed(OUT, **out)
    print(f"wrote {OUT}: {len(out)} arrays, {os.path.getsize(OUT)} bytes")


if __name__ == "__main__":
    main()
