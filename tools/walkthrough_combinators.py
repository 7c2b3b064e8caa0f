#!/usr/bin/env python3
"""Combinators at the HEADLINE block size, written as a user of the reference writes them (allocating calls): composite, scalar multiple,
sum of two tall operators, a nonlinear block operator with its Jacobian -- wall time per line, two passes.  Looks for lines that take seconds.
(Results are dropped section by section: seven live range vectors next to two operators do not fit the device, here or on a host of that size.)

    python tools/walkthrough_combinators.py [nrow] [edge]        (default 512 x 256^3: two operators of 32 GiB + range vectors)
"""
import gc
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jets_jl_amd as J

nrow = int(sys.argv[1]) if len(sys.argv) > 1 else 512
edge = int(sys.argv[2]) if len(sys.argv) > 2 else 256
J.init(0)


class T:
    def __init__(self, what):
        self.what = what

    def __enter__(self):
        J.synchronize()
        self.t0 = time.perf_counter()

    def __exit__(self, *a):
        J.synchronize()
        print(f"    {1e3 * (time.perf_counter() - self.t0):10.2f} ms   {self.what}", flush=True)


blk = J.JetSpace(np.float32, edge, edge, edge)
R = J.JetBSpace([blk] * nrow)
gib = nrow * edge ** 3 * 4 / 2 ** 30
for turn in (1, 2):
    print(f"## pass {turn}: {nrow} x 1 blocks of {edge}^3 Float32 ({gib:.0f} GiB per range vector)", flush=True)
    with T("two operators A1, A2 (coefficients in one slab each)"):
        c1, c2 = J.rand(R, seed=1, stream=0), J.rand(R, seed=5, stream=0)
        A1 = J.blockop([[J.JopDiagonal(c)] for c in c1.arrays])
        A2 = J.blockop([[J.JopDiagonal(c)] for c in c2.arrays])
    m = J.rand(J.domain(A1), seed=2, stream=0)
    with T("d = A1*m"):
        d = A1 * m
    with T("N = A1'oA1 ; y = N*m ; y = N*m"):
        N = A1.H @ A1
        y = N * m
        y = N * m
    with T("C = 2.5*A1 ; dc = C*m ; mc = C'*dc"):
        Cop = 2.5 * A1
        dc = Cop * m
        mc = Cop.H * dc
    del dc, mc
    with T("S = A1 - A2 ; ds = S*m"):
        S = A1 - A2
        ds = S * m
    with T("ds = S*m again ; ms = S'*ds"):
        ds = S * m
        ms = S.H * ds
    del ds, ms
    with T("B = A1'o(A1 - A2) ; yb = B*m   (composite of a sum: one range-sized temporary)"):
        B = A1.H @ (A1 - A2)
        yb = B * m
    with T("yb = B*m again"):
        yb = B * m
    del d, y, yb
    with T("F = block operator of JopSquare ; f = F*m ; Jac = jacobian(F, m) ; df = Jac*m ; dm = Jac'*df"):
        F = J.blockop([[J.JopSquare(blk)] for _ in range(nrow)])
        f = F * m
        Jac = J.jacobian(F, m)
        df = Jac * m
        dm = Jac.H * df
    with T("the same again"):
        f = F * m
        Jac = J.jacobian(F, m)
        df = Jac * m
        dm = Jac.H * df
    with T("linearization_test(F, m) (six mu values)"):
        obs, exp = J.linearization_test(F, m, seed=7)
    with T("dropping everything"):
        for op in (A1, A2, F):
            J.close(op)
        del A1, A2, N, Cop, S, B, F, Jac, c1, c2, f, df, dm, m
        gc.collect()
print(f"# the slab cache holds {J.tune_get('slab_cached_mib') / 1024:.1f} GiB at the end", flush=True)
