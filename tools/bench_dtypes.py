#!/usr/bin/env python3
"""The tall kernels across the four element types at equal bytes (256 rows x 64 MiB blocks = 16 GiB of coefficients):
forward, adjoint, fused A'A, one-pass LSQR step; GB/s of algorithmic bytes."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import jets_jl_amd as J
from jets_jl_amd._ffi import check, lib
from jets_jl_amd.jetblock import _tall_native

J.init(0)
nrow = int(sys.argv[1]) if len(sys.argv) > 1 else 256
block_bytes = 64 << 20


def timed(fn, reps=5):
    fn(); fn()
    best = 1e30
    for _ in range(reps):
        e0 = J.Event().record(); fn(); e1 = J.Event().record()
        best = min(best, e0.elapsed_ms(e1))
    return best


print(f"# {nrow} x 1 tall diagonal operator, 64 MiB per block, best of 5")
for dt in (np.float32, np.float64, np.complex64, np.complex128):
    s = np.dtype(dt).itemsize
    n = block_bytes // s
    spc = J.JetSpace(dt, n)
    coeff = J.rand(J.JetBSpace([spc] * nrow), seed=1, stream=0)
    A = J.blockop([[J.JopDiagonal(c)] for c in coeff.arrays])
    m = J.rand(J.domain(A), seed=2, stream=0)
    d = J.rand(J.range(A), seed=3, stream=0)
    mt, w = J.zeros(J.domain(A)), J.zeros(J.domain(A))
    N = A.H @ A
    out = C.c_double(0)
    nat = _tall_native(A)
    tf = timed(lambda: J.mul_(d, A, m))
    ta = timed(lambda: J.mul_(mt, A.H, d))
    tn = timed(lambda: J.mul_(mt, N, m))
    ts = timed(lambda: check(lib.jh_blockop_bidiag_step(nat.handle, d.handle, m.handle, w.handle, 1.0, -0.5, C.byref(out))))
    b2, b1, b3 = (2 * nrow * n + n) * s, (nrow * n + 2 * n) * s, (3 * nrow * n + 2 * n) * s
    print(f"{np.dtype(dt).name:10s}: fwd {tf:7.3f} ms {b2 / tf / 1e6:6.0f} GB/s | adj {ta:7.3f} ms {b2 / ta / 1e6:6.0f} | A'A {tn:7.3f} ms {b1 / tn / 1e6:6.0f} | "
          f"step {ts:7.3f} ms {b3 / ts / 1e6:6.0f}", flush=True)
    del A, N, coeff, m, d, mt, w, nat
