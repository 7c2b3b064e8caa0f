#!/usr/bin/env python3
"""Tall operator of large diagonal blocks whose coefficient arrays are SEPARATE allocations (the row table is read in the kernels)
against the same operator over one slab (strided addressing): forward, adjoint, fused A'A, one-pass step.
    python tools/bench_table_vs_slab.py NROW [EDGE]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jets_jl_amd as J
from jets_jl_amd import jetblock as _blk
from jets_jl_amd._ffi import check, lib

nrow = int(sys.argv[1]) if len(sys.argv) > 1 else 256
edge = int(sys.argv[2]) if len(sys.argv) > 2 else 256
J.init(0)
n = edge ** 3
spc = J.JetSpace("float32", edge, edge, edge)
out = C.c_double(0)


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


b = n * 4
for table in (False, True, False, True):
    if table:
        diags = [J.rand(spc, seed=1, stream=i) for i in range(nrow)]
    else:
        diags = J.rand(J.JetBSpace([spc] * nrow), seed=1, stream=0).arrays
    A = J.blockop([[J.JopDiagonal(g)] for g in diags])
    m = J.rand(spc, seed=2, stream=0)
    d = J.rand(J.range(A), seed=3, stream=0)
    mt, w = J.zeros(spc), J.zeros(spc)
    N = A.H @ A
    nat = _blk._tall_native(A)
    for _ in range(18):
        J.mul_(d, A, m)
        J.synchronize()
    tf = timed(lambda: J.mul_(d, A, m))
    ta = timed(lambda: J.mul_(mt, A.H, d))
    tn = timed(lambda: J.mul_(w, N, m))
    J.op_tune_set(A, "step_mode", 0)
    ts = timed(lambda: check(lib.jh_blockop_bidiag_step(nat.handle, d.handle, m.handle, w.handle, 1.0, -0.5, C.byref(out))))
    J.op_tune_set(A, "step_mode", 2)
    tc = timed(lambda: check(lib.jh_blockop_bidiag_step(nat.handle, d.handle, m.handle, w.handle, 1.0, -0.5, C.byref(out))))
    print(f"{nrow} x {edge}^3 {'separate arrays' if table else 'one slab':16s}: fwd {tf:7.3f} ms {(2 * nrow + 1) * b / tf / 1e6:7.1f} | adj {ta:7.3f} ms {(2 * nrow + 1) * b / ta / 1e6:7.1f} | "
          f"A'A {tn:7.3f} ms {(nrow + 2) * b / tn / 1e6:7.1f} | step {ts:7.3f} ms {(3 * nrow + 2) * b / ts / 1e6:7.1f} | chained {tc:7.3f} ms {(3 * nrow + 2) * b / tc / 1e6:7.1f} GB/s", flush=True)
    J.close(A)
    del A, N, diags, m, d, mt, w
