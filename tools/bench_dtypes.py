#!/usr/bin/env python3
"""The tall kernels per element type: forward, adjoint, fused A'A and the one-pass LSQR step at NROW rows of 64 MiB blocks
(Float32 256^3, Float64 / ComplexF32 256x256x128, ComplexF64 256x128x128), algorithmic GB/s.   python tools/bench_dtypes.py [NROW]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jets_jl_amd as J
from jets_jl_amd._ffi import lib, check
from jets_jl_amd import jetblock as _blk

nrow = int(sys.argv[1]) if len(sys.argv) > 1 else 128
J.init(0)


def timed(fn, reps=7, warm=3):
    for _ in range(warm):
        fn()
    best = 1e9
    for _ in range(reps):
        e0 = J.Event().record()
        fn()
        e1 = J.Event().record()
        best = min(best, e0.elapsed_ms(e1))
    return best


out = C.c_double(0)
for dt, shape, s in (("float32", (256, 256, 256), 4), ("float64", (256, 256, 128), 8), ("complex64", (256, 256, 128), 8),
                     ("complex128", (256, 128, 128), 16)):
    spc = J.JetSpace(dt, *shape)
    n = shape[0] * shape[1] * shape[2]
    coeff = J.rand(J.JetBSpace([spc] * nrow), seed=1, stream=0)
    A = J.blockop([[J.JopDiagonal(c)] for c in coeff.arrays])
    m = J.rand(spc, seed=2, stream=0)
    d = J.rand(J.JetBSpace([spc] * nrow), seed=3, stream=0)
    mt, w = J.zeros(spc), J.zeros(spc)
    N = A.H @ A
    nat = _blk._tall_native(A)
    blk = n * s
    for _ in range(18):                                                  # let the forward walk settle
        J.mul_(d, A, m)
        J.synchronize()
    t_f = timed(lambda: J.mul_(d, A, m))
    t_a = timed(lambda: J.mul_(mt, A.H, d))
    t_n = timed(lambda: J.mul_(w, N, m))
    res = {}
    for mode in (0, 2):
        J.op_tune_set(A, "step_mode", mode)
        res[mode] = timed(lambda: check(lib.jh_blockop_bidiag_step(nat.handle, d.handle, m.handle, w.handle, 1.0, -0.5, C.byref(out))))
    b1 = (2 * nrow + 1) * blk
    print(f"{dt:10s} {nrow} x {shape}: fwd {t_f:7.3f} ms {b1 / t_f / 1e6:7.1f} GB/s | adj {t_a:7.3f} ms {b1 / t_a / 1e6:7.1f} | "
          f"A'A {t_n:7.3f} ms {(nrow + 2) * blk / t_n / 1e6:7.1f} | step plain {res[0]:7.3f} ms {(3 * nrow + 2) * blk / res[0] / 1e6:7.1f} | "
          f"chained {res[2]:7.3f} ms {(3 * nrow + 2) * blk / res[2] / 1e6:7.1f}  (fwd_walk {J.op_tune_get(A, 'fwd_walk')})", flush=True)
    J.close(A)
    del coeff, d, m, mt, w, A, N
