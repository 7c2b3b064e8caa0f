#!/usr/bin/env python3
"""LSQR iteration cost: two fused halves (jh_blockop_mul_axpby + jh_blockop_mul_adj_axpby, 5*N*n*s bytes) vs the one-pass
Golub-Kahan step (jh_blockop_bidiag_step, 3*N*n*s bytes), kernel shapes swept through the adjoint's knobs.

    python tools/bench_lsqr_step.py NBLOCKS EDGE
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jets_jl_amd as J
from jets_jl_amd._ffi import lib, check
from jets_jl_amd import jetblock as _blk

nblocks = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
edge = int(sys.argv[2]) if len(sys.argv) > 2 else 256
J.init(0)
n = edge ** 3
blk = J.JetSpace("float32", edge, edge, edge)
coeff = J.rand(J.JetBSpace([blk] * nblocks), seed=1, stream=0)
A = J.blockop([[J.JopDiagonal(c)] for c in coeff.arrays])
nat = _blk._native_op(A.jet.s["_native"], A.jet.s["ops"], A.jet.rng.eltype())
u = J.rand(J.range(A), seed=3, stream=0)
v = J.rand(J.domain(A), seed=2, stream=0)
w = J.zeros(J.domain(A))
out = C.c_double(0)
b3 = (3 * nblocks * n + 2 * n) * 4
b5 = (5 * nblocks * n + 3 * n) * 4


def two_halves():
    check(lib.jh_blockop_mul_axpby(nat.handle, u.handle, v.handle, 1.0, -0.5, C.byref(out)))
    check(lib.jh_blockop_mul_adj_axpby(nat.handle, w.handle, u.handle, 1.0, 0.0, 1.0, C.byref(out)))


def one_pass():
    check(lib.jh_blockop_bidiag_step(nat.handle, u.handle, v.handle, w.handle, 1.0, -0.5, C.byref(out)))


def timed(fn, reps=5):
    fn(); fn()
    best = 1e9
    for _ in range(reps):
        e0 = J.Event().record()
        fn()
        e1 = J.Event().record()
        best = min(best, e0.elapsed_ms(e1))
    return best


ms = timed(two_halves)
print(f"{nblocks} x {edge}^3  two fused halves          {ms:8.3f} ms  {b5 / ms / 1e6:7.1f} GB/s algorithmic (5Nn)")
shapes = [dict(adj_wg=0, adj_unroll=0, adj_depth=0)] + [dict(adj_wg=wg, adj_unroll=U, adj_depth=D) for wg in (256, 512, 1024)
                                                         for (U, D) in ((1, 4), (1, 8), (2, 2), (4, 1), (4, 2)) if not (wg == 1024 and (U, D) in ((4, 2), (1, 8)))]
for sh in shapes:
    J.tune(**sh)
    ms1 = timed(one_pass)
    print(f"{nblocks} x {edge}^3  one pass {str(sh):58s} {ms1:8.3f} ms  {b3 / ms1 / 1e6:7.1f} GB/s algorithmic (3Nn)  {ms / ms1:4.2f}x", flush=True)
