#!/usr/bin/env python3
"""Look for performance cliffs: block operators of ~1 GiB of Float32 coefficients in many shapes (tall / wide / square grids,
big and small blocks, mixed kinds), forward and adjoint, GB/s of UNIQUE bytes (every coefficient once, the vectors once)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import jets_jl_amd as J

J.init(0)
if os.environ.get('GENERAL_XCD'):
    J.tune(general_xcd=int(os.environ['GENERAL_XCD']))
total = (int(sys.argv[1]) if len(sys.argv) > 1 else 1024) << 20


def timed(fn, reps=3):
    fn(); fn()
    best = 1e30
    for _ in range(reps):
        e0 = J.Event().record(); fn(); e1 = J.Event().record()
        best = min(best, e0.elapsed_ms(e1))
    return best


def run(tag, nrow, ncol, n, kind_of):
    spc = J.JetSpace(np.float32, n)
    ndiag = sum(1 for i in range(nrow) for j in range(ncol) if kind_of(i, j) == "diag")
    slab = J.rand(J.JetBSpace([spc] * max(ndiag, 1)), seed=1, stream=0).arrays
    k = 0
    rows = []
    for i in range(nrow):
        row = []
        for j in range(ncol):
            kd = kind_of(i, j)
            if kd == "diag":
                row.append(J.JopDiagonal(slab[k])); k += 1
            elif kd == "identity":
                row.append(J.JopIdentity(spc))
            elif kd == "scale":
                row.append(J.JopLn(dom=spc, rng=spc, df=J.constdiag_df, df_adj=J.constdiag_df_adj, s={"a": 0.5 + i}))
            else:
                row.append(J.JopZeroBlock(spc, spc))
        rows.append(row)
    A = J.blockop(rows)
    m = J.rand(J.domain(A), seed=2, stream=0)
    d = J.rand(J.range(A), seed=3, stream=0)
    mt = J.zeros(J.domain(A))
    nz = sum(1 for i in range(nrow) for j in range(ncol) if kind_of(i, j) != "zero")
    uniq = (ndiag * n + nrow * n + ncol * n) * 4
    tf = timed(lambda: J.mul_(d, A, m))
    ta = timed(lambda: J.mul_(mt, A.H, d))
    flag = "  <-- CLIFF" if min(uniq / tf, uniq / ta) / 1e6 < 1500 and uniq > (64 << 20) else ""
    print(f"{tag:34s} {nrow:6d} x {ncol:5d} blocks of {n:8d}: fwd {tf:8.3f} ms {uniq / tf / 1e6:6.0f} GB/s | adj {ta:8.3f} ms {uniq / ta / 1e6:6.0f} GB/s{flag}", flush=True)


diag = lambda i, j: "diag"
for nrow, ncol in ((8, 8), (32, 32), (2, 512), (512, 2), (1, 1024), (1, 16384), (16384, 1), (4096, 4)):
    n = total // 4 // (nrow * ncol)
    if n >= 4:
        run("all diagonal", nrow, ncol, n - n % 4, diag)
mixed = lambda i, j: ("diag", "identity", "scale", "zero")[(i + j) % 4]
for nrow, ncol in ((1024, 1), (16384, 1), (65536, 1), (64, 64), (1, 4096)):
    n = total // 4 // (nrow * ncol) * 4          # a quarter of the blocks carry coefficients
    if n >= 4:
        run("mixed diag/identity/scale/zero", nrow, ncol, n - n % 4, mixed)
halfid = lambda i, j: "diag" if i % 2 else "identity"
for nrow in (1024, 65536):
    n = total // 4 // nrow * 2
    run("tall, half identity rows", nrow, 1, n - n % 4, halfid)
