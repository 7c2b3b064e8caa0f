#!/usr/bin/env python3
"""Tall operators whose block length is ODD (rows off the 16-byte pack grid of the slab): forward + adjoint pair, fused A'A, the one-pass LSQR step and a
few LSQR iterations, under-aligned tall kernels (tall_unaligned = 1, the default since round 5) against the general kernels (tall_unaligned = 0, what such
operators ran on before), and the aligned neighbour (EDGE + 1 when EDGE is odd) for scale.  Coefficients: blocks of ONE slab (off the grid like the range
vector) or one allocation per block (SEPARATE=1: every diagonal 256-byte aligned, what a caller in the reference's style has).

    python tools/bench_unaligned.py NROW EDGE [float32|float64|complex64]
Algorithmic bytes: pair 4 N n s + 2 n s; A'A N n s + 2 n s; step 3 N n s + 2 n s."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import jets_jl_amd as J
from jets_jl_amd import jetblock as _blk
from jets_jl_amd._ffi import check, lib

nrow = int(sys.argv[1]) if len(sys.argv) > 1 else 256
edge = int(sys.argv[2]) if len(sys.argv) > 2 else 101
dt = sys.argv[3] if len(sys.argv) > 3 else "float32"
separate = os.environ.get("SEPARATE", "0") == "1"
J.init(0)
es = np.dtype(dt).itemsize


def timed(fn, reps=5):
    fn(); fn()
    best = 1e9
    for _ in range(reps):
        e0 = J.Event().record(); fn(); e1 = J.Event().record()
        best = min(best, e0.elapsed_ms(e1))
    return best


def run(e, knob):
    J.tune(tall_unaligned=knob)
    spc = J.JetSpace(dt, e, e, e)
    n = e ** 3
    if separate:
        diags = [J.rand(spc, seed=1, stream=i) for i in range(nrow)]
    else:
        diags = J.rand(J.JetBSpace([spc] * nrow), seed=1, stream=0).arrays
    A = J.blockop([[J.JopDiagonal(g)] for g in diags])
    m = J.rand(spc, seed=2, stream=0)
    d = J.rand(J.range(A), seed=3, stream=0)
    mt, w = J.zeros(spc), J.zeros(spc)
    N = J.compose(A.H, A)
    nat = _blk._native_op(A.jet.s["_native"], A.jet.s["ops"], A.jet.rng.eltype())
    out = C.c_double(0)
    t_pair = timed(lambda: (J.mul_(d, A, m), J.mul_(mt, A.H, d)))
    t_n = timed(lambda: J.mul_(mt, N, m))
    st = lib.jh_blockop_bidiag_step(nat.handle, d.handle, m.handle, w.handle, 1.0, -0.5, C.byref(out))
    t_step = timed(lambda: check(lib.jh_blockop_bidiag_step(nat.handle, d.handle, m.handle, w.handle, 1.0, -0.5, C.byref(out)))) if st == 0 else float("nan")
    b = J.rand(J.range(A), seed=5, stream=0)
    iters = 10
    J.lsqr(A, b, atol=0.0, btol=0.0, conlim=0.0, maxiter=2)
    e0 = J.Event().record(); J.lsqr(A, b, atol=0.0, btol=0.0, conlim=0.0, maxiter=iters); e1 = J.Event().record()
    J.synchronize()
    t_lsqr = e0.elapsed_ms(e1) / iters
    by_pair, by_n, by_step = (4 * nrow + 2) * n * es, (nrow + 2) * n * es, (3 * nrow + 2) * n * es
    tag = f"{nrow} x {e}^3 {dt} ({'separate diagonals' if separate else 'diagonals in one slab'}) tall_unaligned={knob}"
    print(f"{tag:86s} pair {t_pair:8.3f} ms {by_pair / t_pair / 1e6:6.0f} GB/s | A'A {t_n:7.3f} ms {by_n / t_n / 1e6:6.0f} | step {t_step:8.3f} ms "
          f"{by_step / t_step / 1e6:6.0f} | LSQR {t_lsqr:8.3f} ms/iter {by_step / t_lsqr / 1e6:6.0f}", flush=True)
    J.close(A)
    del A, d, b, diags
    J.trim() if hasattr(J, "trim") else None


for knob in (1, 0, 1):
    run(edge, knob)
if edge % 2:
    run(edge + 1, 1)
J.tune(tall_unaligned=1)
