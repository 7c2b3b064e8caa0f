#!/usr/bin/env python3
"""The chained one-pass step (k_tall_diag_bidiag_chain: one batch of 8 rows of one tile per workgroup, tiles fastest) at 1024 / 512 / 256 lanes per workgroup
(knobs step_chain = 1 + adj_wg) against the plain walk.   python tools/exp_chain_wg.py [NROW EDGE]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jets_jl_amd as J
from jets_jl_amd import jetblock as _blk
from jets_jl_amd._ffi import check, lib

J.init(0)
shapes = [(128, 256), (256, 256), (1024, 256), (1024, 128)]
if len(sys.argv) > 2:
    shapes = [(int(sys.argv[1]), int(sys.argv[2]))]
for nblocks, edge in shapes:
    n = edge ** 3
    blk = J.JetSpace("float32", edge, edge, edge)
    coeff = J.rand(J.JetBSpace([blk] * nblocks), seed=1, stream=0)
    A = J.blockop([[J.JopDiagonal(c)] for c in coeff.arrays])
    nat = _blk._tall_native(A)
    u = J.rand(J.range(A), seed=3, stream=0)
    v = J.rand(J.domain(A), seed=2, stream=0)
    w = J.zeros(J.domain(A))
    out = C.c_double(0)

    def one_pass():
        check(lib.jh_blockop_bidiag_step(nat.handle, u.handle, v.handle, w.handle, 1.0, -0.5, C.byref(out)))

    def timed(reps=7):
        one_pass()
        one_pass()
        best = 1e9
        for _ in range(reps):
            e0 = J.Event().record()
            one_pass()
            e1 = J.Event().record()
            best = min(best, e0.elapsed_ms(e1))
        return best

    b3 = (3 * nblocks * n + 2 * n) * 4
    nat.tune_set("step_mode", 0)
    t = timed()
    print(f"{nblocks:5d} x {edge}^3 one-pass step, plain walk:          {t:8.3f} ms {b3 / t / 1e6:7.1f} GB/s", flush=True)
    nat.tune_set("step_mode", -1)
    for wg in (1024, 512, 256, 1024):
        J.tune(step_chain=1, adj_wg=wg)
        t = timed()
        print(f"{nblocks:5d} x {edge}^3 one-pass step, chained, {wg:4d} lanes: {t:8.3f} ms {b3 / t / 1e6:7.1f} GB/s  (chunks {J.tune_get('last_step_chain')})", flush=True)
    J.tune(step_chain=-1, adj_wg=0)
    del u, v, w, coeff
    J.close(A)
