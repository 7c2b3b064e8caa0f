#!/usr/bin/env python3
"""Cliff hunt for the fused chains (round 6): A' o W o A, (W o A)' and W o A over a grid of row counts x block lengths (Float32; ~1-4 GiB per case, blocks on
and off the 16-byte grid), TB/s over the algorithmic bytes beside the plain fused A'A / adjoint / forward of the same operator.  A shape whose chain runs far
below its plain neighbour is a cliff.

    python tools/cliff_hunt_chains.py [TOTAL_MIB]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jets_jl_amd as J
from jets_jl_amd import chains

J.init(0)
total = (int(sys.argv[1]) if len(sys.argv) > 1 else 2048) << 20


def timed(fn, reps=5):
    fn(); fn()
    best = 1e9
    for _ in range(reps):
        e0 = J.Event().record(); fn(); e1 = J.Event().record()
        best = min(best, e0.elapsed_ms(e1))
    return best


print(f"# {total >> 20} MiB of coefficients per case; TB/s over algorithmic bytes: chain | plain", flush=True)
for n in (513, 1024, 4099, 16384, 65536, 250047, 1 << 20, 1 << 22, (1 << 24) - 1, 1 << 24):
    nrow = max(2, total // (4 * n))
    if nrow > 1 << 20:
        continue
    spc = J.JetSpace(np.float32, n)
    R = J.JetBSpace([spc] * nrow)
    A = J.blockop([[J.JopDiagonal(c)] for c in J.rand(R, seed=1, stream=0).arrays])
    W = J.JopDiagonal(J.rand(R, seed=5, stream=0))
    m, y, d = J.rand(spc, seed=2, stream=0), J.zeros(spc), J.zeros(R)
    Nn = nrow * n * 4
    before = chains.STATS["chain_calls"]
    NW, WA, NA = J.compose(J.compose(A.H, W), A), J.compose(W, A), J.compose(A.H, A)
    WAH = WA.H
    t_n = timed(lambda: J.mul_(y, NW, m))
    t_a = timed(lambda: J.mul_(y, WAH, d))
    t_f = timed(lambda: J.mul_(d, WA, m))
    ran = chains.STATS["chain_calls"] - before
    p_n = timed(lambda: J.mul_(y, NA, m))
    p_a = timed(lambda: J.mul_(y, A.H, d))
    p_f = timed(lambda: J.mul_(d, A, m))
    print(f"{nrow:7d} x {n:9d}: A'WA {2 * Nn / t_n / 1e9:5.2f} | A'A {Nn / p_n / 1e9:5.2f}    (WA)' {3 * Nn / t_a / 1e9:5.2f} | A' {2 * Nn / p_a / 1e9:5.2f}    "
          f"WA {3 * Nn / t_f / 1e9:5.2f} | A {2 * Nn / p_f / 1e9:5.2f}   {'fused' if ran >= 21 else 'NOT ALL FUSED (' + str(ran) + ')'}", flush=True)
    # what a caller in the allocating style pays who builds the composite anew for every application (a new chain handle: a row table of nrow records)
    t_new = timed(lambda: J.mul_(y, J.compose(J.compose(A.H, W), A), m), reps=3)
    print(f"{'':20s} composite built per call: A'WA {t_new:8.3f} ms against {t_n:8.3f} ms with the composite kept", flush=True)
    J.close(A)
    del A, W, m, y, d, R, NW, WA, NA, WAH
