#!/usr/bin/env python3
"""Third cliff hunt (round 5, last session): tall operators of MANY SMALL rows -- traces rather than volumes: 65536 x 2049, 262144 x 513 ... -- with an odd
row length beside the even neighbour: forward, adjoint, fused A'A and the one-pass step, GB/s of the algorithmic bytes.    python tools/cliff_hunt3.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jets_jl_amd as J
J.init(0)
def timed(fn, reps=3):
    fn(); fn(); fn()
    best = 1e30
    for _ in range(reps):
        e0 = J.Event().record(); fn(); e1 = J.Event().record()
        best = min(best, e0.elapsed_ms(e1))
    return best
import ctypes as C
from jets_jl_amd._ffi import lib, check
from jets_jl_amd import jetblock
for nrow, n in ((65536, 2049), (65536, 2048), (16384, 8193), (16384, 8192), (4096, 32769), (4096, 32768), (262144, 513), (262144, 512)):
    spc = J.JetSpace("float32", n)
    slab = J.rand(J.JetBSpace([spc] * nrow), seed=1, stream=0).arrays
    A = J.blockop([[J.JopDiagonal(g)] for g in slab])
    m = J.rand(spc, seed=2, stream=0); d = J.rand(J.range(A), seed=3, stream=0); mt = J.zeros(spc); w = J.zeros(spc)
    by = (2 * nrow + 1) * n * 4
    tf = timed(lambda: J.mul_(d, A, m)); ta = timed(lambda: J.mul_(mt, A.H, d))
    N = J.compose(A.H, A); tn = timed(lambda: J.mul_(mt, N, m))
    nat = jetblock._native_op(A.jet.s["_native"], A.jet.s["ops"], A.jet.rng.eltype()); out = C.c_double(0)
    ts = timed(lambda: check(lib.jh_blockop_bidiag_step(nat.handle, d.handle, m.handle, w.handle, 1.0, -0.5, C.byref(out))))
    print(f"{nrow:7d} x {n:6d}: fwd {tf:7.3f} ms {by/tf/1e6:6.0f} GB/s | adj {ta:7.3f} ms {by/ta/1e6:6.0f} | A'A {tn:7.3f} ms {(nrow+2)*n*4/tn/1e6:6.0f} | step {ts:7.3f} ms {(3*nrow+2)*n*4/ts/1e6:6.0f}", flush=True)
    J.close(A); del A, slab, d
