#!/usr/bin/env python3
"""256 x 256^3 forward / adjoint / one-pass step with the range vector allocated right after the coefficients or behind a
spacer of 16..128 GiB: does the distance between the read slab and the written slab matter?"""
import ctypes as C
import gc
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jets_jl_amd as J
from jets_jl_amd._ffi import lib, check
from jets_jl_amd import jetblock as _blk

J.init(0)
edge, N = 256, int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = edge ** 3
blk = J.JetSpace(np.float32, edge, edge, edge)
coeff = J.rand(J.JetBSpace([blk] * N), seed=1, stream=0)
A = J.blockop([[J.JopDiagonal(c)] for c in coeff.arrays])
nat = _blk._native_op(A.jet.s["_native"], A.jet.s["ops"], A.jet.rng.eltype())
m, w = J.rand(blk, seed=2, stream=0), J.zeros(blk)
mt = J.zeros(blk)
nb2, nb3 = (2 * N * n + n) * 4, (3 * N * n + 2 * n) * 4
GiB = 1 << 30


def timed(fn, reps=4):
    fn(); fn()
    best = 1e9
    for _ in range(reps):
        e0 = J.Event().record(); fn(); e1 = J.Event().record()
        best = min(best, e0.elapsed_ms(e1))
    return best


out = C.c_double(0)
for spacer_gib in (0, 16, 48, 96, 128, 0):
    spacer = J.zeros(J.JetSpace(np.float32, spacer_gib * GiB // 4)) if spacer_gib else None
    d = J.rand(J.JetBSpace([blk] * N), seed=3, stream=0)
    tf = timed(lambda: J.mul_(d, A, m))
    ta = timed(lambda: J.mul_(mt, A.H, d))
    tb = timed(lambda: check(lib.jh_blockop_bidiag_step(nat.handle, d.handle, m.handle, w.handle, 1.0, -0.5, C.byref(out))))
    print(f"{N} rows, spacer {spacer_gib:4d} GiB (d at 0x{d.ptr:x}, a at 0x{coeff.ptr:x}): forward {tf:7.3f} ms {nb2 / tf / 1e6:7.1f} GB/s | "
          f"adjoint {ta:7.3f} ms {nb2 / ta / 1e6:7.1f} GB/s | one-pass step {tb:7.3f} ms {nb3 / tb / 1e6:7.1f} GB/s", flush=True)
    del d, spacer
    gc.collect()
