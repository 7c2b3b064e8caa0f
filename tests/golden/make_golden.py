#!/usr/bin/env python3
"""Generates tests/golden/jets_block_path_v1.npz: small input/expected-output vectors for the block
mul! path.

Provenance: the reference (Julia) cannot be executed in this image and stores no golden vectors of its
own, so these fixtures are produced by the CPU oracle (oracle/jets_oracle.c), which is itself pinned
against the reference's test identities by tests/test_oracle_pinning.py.  They freeze today's
bit patterns: a later change to the oracle or to a HIP kernel that alters any result bit fails
tests/test_golden.py.  Data only -- inputs and expected outputs, no reference source text.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import jets_oracle as jo  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "jets_block_path_v1.npz")


def signed(dt, seed, stream, n):
    """U[0,1) -> (-1,1) (and a complex spread) with exactly representable steps, so inputs have both signs."""
    x = jo.rng_u01(dt, seed, stream, 0, n)
    return (x * 2 - (1 + 1j if np.dtype(dt).kind == "c" else 1)).astype(dt)


def tall_case(dt, nrow, n, tag, out):
    a = [signed(dt, 11, i, n) for i in range(nrow)]
    m = signed(dt, 12, 0, n)
    d = [signed(dt, 13, i, n) for i in range(nrow)]
    ops = [[jo.Block("diag", n, coeff=g)] for g in a]
    fwd = jo.block_df(ops, [np.zeros(n, dtype=dt) for _ in range(nrow)], [m])
    adj = jo.block_df_adj(ops, [np.zeros(n, dtype=dt)], d)
    nrm = jo.normal_df(ops, [np.zeros(n, dtype=dt)], [m])
    out.update({f"{tag}_a": np.stack(a), f"{tag}_m": m, f"{tag}_d": np.stack(d), f"{tag}_fwd": np.stack(fwd),
                f"{tag}_adj": adj[0], f"{tag}_normal": nrm[0]})


MIXED_KINDS = [["diag", "identity", "diag", "scale"], ["diag_adj", "zero", "diag", "diag"], ["scale", "diag", "diag_adj", "zero"]]


def mixed_blocks(dt, n, coeffs):
    rows = []
    for i, row in enumerate(MIXED_KINDS):
        r = []
        for j, k in enumerate(row):
            if k == "zero":
                r.append(jo.Block("zero", n, n))
            elif k == "identity":
                r.append(jo.Block("identity", n))
            elif k == "scale":
                r.append(jo.Block("scale", n, scale=(0.5 + i) - (0.25j * (j + 1) if np.dtype(dt).kind == "c" else 0)))
            else:
                r.append(jo.Block("diag", n, coeff=coeffs[i, j], adjoint=(k == "diag_adj")))
        rows.append(r)
    return rows


def mixed_case(dt, n, tag, out):
    coeffs = np.stack([np.stack([signed(dt, 21, 10 * i + j, n) for j in range(4)]) for i in range(3)])
    ops = mixed_blocks(dt, n, coeffs)
    m = [signed(dt, 22, j, n) for j in range(4)]
    d = [signed(dt, 23, i, n) for i in range(3)]
    d0 = [signed(dt, 24, i, n) for i in range(3)]
    fwd = jo.block_df(ops, [np.zeros(n, dtype=dt) for _ in range(3)], m)
    fwd_dirty = jo.block_df(ops, [x.copy() for x in d0], m)
    adj = jo.block_df_adj(ops, [np.zeros(n, dtype=dt) for _ in range(4)], d)
    out.update({f"{tag}_coeffs": coeffs, f"{tag}_m": np.stack(m), f"{tag}_d": np.stack(d), f"{tag}_d0": np.stack(d0),
                f"{tag}_fwd": np.stack(fwd), f"{tag}_fwd_dirty": np.stack(fwd_dirty), f"{tag}_adj": np.stack(adj)})


def vector_case(dt, tag, out):
    lens = [5, 1, 21, 64, 129]
    u, v, w = ([signed(dt, 31 + k, i, ln) for i, ln in enumerate(lens)] for k in range(3))
    coef = [0.37, -0.81, 1.59] if np.dtype(dt).kind != "c" else [0.37 + 0.2j, -0.81 - 0.4j, 1.59 + 1.5j]
    x = jo.barr_lincomb([np.empty_like(t) for t in u], coef, [u, v, w])
    out.update({f"{tag}_lens": np.array(lens), f"{tag}_u": np.concatenate(u), f"{tag}_v": np.concatenate(v),
                f"{tag}_w": np.concatenate(w), f"{tag}_coef": np.array(coef), f"{tag}_x": np.concatenate(x)})
    ps = [2.0, 1.0, 0.0, np.inf, -np.inf, 3.0]
    out[f"{tag}_norm_p"] = np.array(ps)
    out[f"{tag}_norms"] = np.array([jo.barr_norm(u, p) for p in ps])
    dv = jo.barr_dot(u, v)
    out[f"{tag}_dot"] = np.array([np.real(dv), np.imag(dv)])
    if np.dtype(dt).kind != "c":
        out[f"{tag}_extrema"] = np.array(jo.barr_extrema(u))


def main():
    out = {}
    tall_case(np.float32, 5, 96, "tall_f32", out)
    tall_case(np.float64, 4, 50, "tall_f64", out)
    tall_case(np.complex64, 3, 40, "tall_c32", out)
    tall_case(np.complex128, 3, 24, "tall_c64", out)
    mixed_case(np.float64, 10, "mixed_f64", out)
    mixed_case(np.complex64, 12, "mixed_c32", out)
    vector_case(np.float32, "vec_f32", out)
    vector_case(np.complex128, "vec_c64", out)
    np.savez_compressed(OUT, **out)
    print(f"wrote {OUT}: {len(out)} arrays, {os.path.getsize(OUT)} bytes")


if __name__ == "__main__":
    main()
