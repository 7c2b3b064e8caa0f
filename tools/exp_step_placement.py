#!/usr/bin/env python3
"""Does the one-pass LSQR step (reads the coefficients, reads AND writes the range vector in place) care which allocation holds what, the way the
forward does?  Three candidate 64 GiB slabs, every ordered (coefficients, range vector) pair: forward, adjoint and step times.

    python tools/exp_step_placement.py
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jets_jl_amd as J
from jets_jl_amd._ffi import lib, check
from jets_jl_amd import jetblock

J.init(0)
nrow, edge = 1024, 256
blk = J.JetSpace(np.float32, edge, edge, edge)
R = J.JetBSpace([blk] * nrow)
xs = [J.Array(R, undef=True) for _ in range(3)]
for k, x in enumerate(xs):
    J.rand_(x, seed=1 + k, stream=0)
v, w, mt = J.rand(blk, seed=9, stream=0), J.zeros(blk), J.zeros(blk)
print("# coefficients in slab i, range vector in slab j: forward / adjoint / one-pass step (plain and chained walk), ms", flush=True)
for i in range(3):
    for j in range(3):
        if i == j:
            continue
        A = J.blockop([[J.JopDiagonal(c)] for c in xs[i].arrays])
        h = jetblock._native_op(A.jet.s["_native"], A.jet.s["ops"], A.jet.rng.eltype()).handle
        jetblock.op_tune_set(A, "fwd_walk", 7)
        u = xs[j]
        out = C.c_double(0)

        def timed(fn, reps=3):
            fn()
            J.synchronize()
            e0 = J.Event().record()
            for _ in range(reps):
                fn()
            e1 = J.Event().record()
            return e0.elapsed_ms(e1) / reps

        tf = timed(lambda: J.mul_(u, A, v))
        ta = timed(lambda: J.mul_(mt, A.H, u))
        res = []
        for mode in (0, 2):
            J.tune(step_chain=1 if mode == 2 else 0)
            res.append(timed(lambda: check(lib.jh_blockop_bidiag_step(h, u.handle, v.handle, w.handle, 1.0, -0.5, C.byref(out)))))
        J.tune(step_chain=-1)
        print(f"  {i} -> {j}: forward {tf:7.3f}  adjoint {ta:7.3f}  step plain {res[0]:7.3f}  step chained {res[1]:7.3f}", flush=True)
        J.close(A)
