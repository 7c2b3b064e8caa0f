#!/usr/bin/env python3
"""Soak of the chained one-pass step's hand-off protocol at full size: STEPS steps on two copies of the same state, one through the
plain ordered walk, one through the chained row chunks; w (whole) and slices of u must stay bit-identical all the way, the sticky
error word must never be raised (an expired poll fails the call).      python tools/soak_step_chain.py [NROW] [STEPS]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import jets_jl_amd as J
from jets_jl_amd import jetblock as _blk
from jets_jl_amd._ffi import check, lib

nrow = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
edge = 256
J.init(0)
n = edge ** 3
spc = J.JetSpace("float32", edge, edge, edge)
coeff = J.rand(J.JetBSpace([spc] * nrow), seed=1, stream=0)
A = J.blockop([[J.JopDiagonal(c)] for c in coeff.arrays])
nat = _blk._tall_native(A)
u1 = J.rand(J.range(A), seed=3, stream=0)
u2 = J.rand(J.range(A), seed=3, stream=0)
v = J.rand(spc, seed=2, stream=0)
w1, w2 = J.zeros(spc), J.zeros(spc)
o1, o2 = C.c_double(0), C.c_double(0)
t0 = time.time()
chunks = 0
for k in range(steps):
    alpha, beta = (1.0, -0.5) if k % 3 else (0.75, 0.25)          # u stays bounded
    J.op_tune_set(A, "step_mode", 0)
    check(lib.jh_blockop_bidiag_step(nat.handle, u1.handle, v.handle, w1.handle, alpha, beta, C.byref(o1)))
    J.op_tune_set(A, "step_mode", 2)
    check(lib.jh_blockop_bidiag_step(nat.handle, u2.handle, v.handle, w2.handle, alpha, beta, C.byref(o2)))
    chunks = J.tune_get("last_step_chain")
    assert chunks > 0, "the chained walk did not run"
    assert abs(o1.value - o2.value) <= 1e-12 * o1.value, (k, o1.value, o2.value)
    if k % 100 == 99 or k == steps - 1:
        a, b = w1.to_numpy(), w2.to_numpy()
        assert a.tobytes() == b.tobytes(), f"step {k}: w differs"
        for off in (0, (nrow // 2) * n + 12345, nrow * n - 65536):
            assert u1._download(off, 65536).tobytes() == u2._download(off, 65536).tobytes(), f"step {k}: u differs at {off}"
        print(f"step {k + 1}: w and u slices bit-identical, ||u||^2 {o1.value:.6e}, {time.time() - t0:.0f} s", flush=True)
print(f"soak ok: {steps} chained steps of {nrow} x {edge}^3 ({chunks} chunks x {n // 4096} tiles = {chunks * (n // 4096)} hand-offs per step), no expired poll, bits of the plain walk")
