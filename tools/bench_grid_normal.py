#!/usr/bin/env python3
"""The fused normal operator of an N x K grid of equal diagonals, K = 2 .. 4 (round 6; jh_grid_normal.hip) against the two stages the reference applies
(JetBlock_df! into zeros(range(A)), then JetBlock_df'!; knob grid_normal = 0).  REG=1: K regularisation rows (lam * I on the diagonal, zero blocks
elsewhere) under the data rows -- a grid with blocks of several kinds.

    python tools/bench_grid_normal.py [N K EDGE [dtype]]...        default: a sweep

Algorithmic bytes (s = element size, n = EDGE^3): fused N K n s + 2 K n s; the two stages 2 N K n s + 2 N n s + 2 K n s."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import jets_jl_amd as J

J.init(0)
for kv in os.environ.get("JETS_TUNE", "").split(","):          # e.g. JETS_TUNE=adj_split=0
    if "=" in kv:
        J.tune(**{kv.split("=")[0]: int(kv.split("=")[1])})


def timed(fn, reps=7):
    fn(); fn()
    best = 1e9
    for _ in range(reps):
        e0 = J.Event().record(); fn(); e1 = J.Event().record()
        best = min(best, e0.elapsed_ms(e1))
    return best


def case(N, K, edge, dt="float32"):
    spc = J.JetSpace(dt, edge, edge, edge)
    n, s = edge ** 3, np.dtype(dt).itemsize
    coeff = J.rand(J.JetBSpace([spc] * (N * K)), seed=1, stream=0)
    rows = [[J.JopDiagonal(coeff.arrays[i * K + j]) for j in range(K)] for i in range(N)]
    if os.environ.get("REG", "0") == "1":                                   # the regularised operator: K more rows, lam * I on the diagonal, zero blocks elsewhere
        lam = lambda: J.JopLn(dom=spc, rng=spc, df=J.constdiag_df, df_adj=J.constdiag_df_adj, s={"a": 0.5})
        rows += [[lam() if j == k else J.JopZeroBlock(spc, spc) for j in range(K)] for k in range(K)]
    A = J.blockop(rows)
    m = J.rand(J.domain(A), seed=2, stream=0)
    y = J.zeros(J.domain(A))
    NA = J.compose(A.H, A)
    J.tune(grid_normal=int(os.environ.get("GRID_NORMAL", "1")))
    tf = timed(lambda: J.mul_(y, NA, m))
    parts = J.tune_get("last_adj_parts")
    J.tune(grid_normal=0)
    tu = timed(lambda: J.mul_(y, NA, m))
    J.tune(grid_normal=1)
    fused_b = (N * K + 2 * K) * n * s
    print(f"{N:5d}{' + ' + str(K) + ' regularisation rows' if os.environ.get('REG', '0') == '1' else ''} x {K} of {edge}^3 {dt}: fused {tf:8.3f} ms {fused_b / tf / 1e9:6.3f} TB/s ({100 * fused_b / tf / 8e9:5.1f} % of 8 TB/s, {parts} part{'s' if parts > 1 else ''}) | "
          f"two stages {tu:8.3f} ms | {tu / tf:5.2f}x", flush=True)
    if os.environ.get("CGNR", "0") == "1":                                 # CG on the normal equations: jh_cgnr_solve through the fused pass | A then A' through the engines
        b = J.rand(J.range(A), seed=5, stream=0)
        for native in ("1", "0"):
            os.environ["JETS_CGLS_NATIVE"] = native
            J.cgnr(A, b, atol=0.0, btol=0.0, maxiter=2)
            e0 = J.Event().record(); J.cgnr(A, b, atol=0.0, btol=0.0, maxiter=12, force_maxiter=True); e1 = J.Event().record()
            e2 = J.Event().record(); J.cgnr(A, b, atol=0.0, btol=0.0, maxiter=2, force_maxiter=True); e3 = J.Event().record()
            J.synchronize()
            print(f"        CGNR {'jh_cgnr_solve (fused A^T A)' if native == '1' else 'engines (A, then A^T)      '}: {(e0.elapsed_ms(e1) - e2.elapsed_ms(e3)) / 10:8.3f} ms per iteration", flush=True)
        os.environ["JETS_CGLS_NATIVE"] = "1"
    J.close(A)


args = sys.argv[1:]
if args:
    k = 0
    while k + 2 < len(args) + 0 and k + 3 <= len(args):
        dt = args[k + 3] if k + 3 < len(args) and not args[k + 3].isdigit() else "float32"
        case(int(args[k]), int(args[k + 1]), int(args[k + 2]), dt)
        k += 4 if dt != "float32" or (k + 3 < len(args) and not args[k + 3].isdigit()) else 3
else:
    for N, K, e, dt in ((64, 4, 256, "float32"), (128, 2, 256, "float32"), (64, 3, 256, "float32"), (256, 4, 128, "float32"), (1024, 2, 128, "float32"),
                        (4096, 3, 64, "float32"), (16384, 2, 32, "float32"), (64, 4, 255, "float32"), (256, 4, 101, "float32"),
                        (64, 4, 200, "float64"), (64, 4, 200, "complex64"), (32, 4, 200, "complex128")):
        case(N, K, e, dt)
