#!/usr/bin/env python3
"""The one-pass step the way the pipelined multi-GPU exchange issues it -- four element ranges back to back, ||u||^2 deferred,
one read-back -- for each walk (0 plain, 1 XCD-contiguous tiles, 2 chained) against the whole-vector call.
    python tools/ab_step_ranged.py [NROW EDGE]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jets_jl_amd as J
from jets_jl_amd._ffi import lib, check
from jets_jl_amd import jetblock as _blk
J.init(0)
shapes = [(128, 256), (256, 256), (512, 256)] if len(sys.argv) < 3 else [(int(sys.argv[1]), int(sys.argv[2]))]
for nblocks, edge in shapes:
    n = edge ** 3
    blk = J.JetSpace("float32", edge, edge, edge)
    coeff = J.rand(J.JetBSpace([blk] * nblocks), seed=1, stream=0)
    A = J.blockop([[J.JopDiagonal(c)] for c in coeff.arrays])
    nat = _blk._tall_native(A)
    u = J.rand(J.range(A), seed=3, stream=0); v = J.rand(J.domain(A), seed=2, stream=0); w = J.zeros(J.domain(A))
    out = C.c_double(0)
    q = n // 4
    def whole():
        check(lib.jh_blockop_bidiag_step(nat.handle, u.handle, v.handle, w.handle, 1.0, -0.5, C.byref(out)))
    def ranged():
        check(lib.jh_normsq_reset())
        for r in range(4):
            check(lib.jh_blockop_bidiag_step_range(nat.handle, u.handle, v.handle, w.handle, 1.0, -0.5, r * q, q, None))
        check(lib.jh_normsq_read(C.byref(out)))
    def timed(fn, reps=7):
        fn(); fn()
        best = 1e9
        for _ in range(reps):
            e0 = J.Event().record(); fn(); e1 = J.Event().record()
            best = min(best, e0.elapsed_ms(e1))
        return best
    b3 = (3 * nblocks * n + 2 * n) * 4
    row = []
    for mode in (0, 1, 2):
        nat.tune_set("step_mode", mode)
        row.append((timed(whole), timed(ranged)))
    print(f"{nblocks:5d} x {edge}^3 one-pass step, whole / in 4 ranges:  " + "  |  ".join(
        f"mode {m}: {a:7.3f} / {b:7.3f} ms ({b3/a/1e6:6.0f} / {b3/b/1e6:6.0f} GB/s)" for m, (a, b) in enumerate(row)), flush=True)
    del u, v, w, coeff; J.close(A)
