#!/usr/bin/env python3
"""One-pass LSQR step on a tall operator with rows of several kinds: which (workgroup, vectors per lane, rows in flight) shape of the
MIXED k_tall_diag_bidiag is fastest at a given block size.    python tools/exp_mixed_step_shapes.py NROW EDGE"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jets_jl_amd as J
from jets_jl_amd._ffi import lib, check
from jets_jl_amd import jetblock as _blk

nrow = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
edge = int(sys.argv[2]) if len(sys.argv) > 2 else 128
J.init(0)
n = edge ** 3
spc = J.JetSpace("float32", edge, edge, edge)
coeff = J.rand(J.JetBSpace([spc] * nrow), seed=1, stream=0)
rows = [[J.JopDiagonal(coeff.arrays[i])] for i in range(nrow - 2)]
rows += [[J.JopZeroBlock(spc, spc)], [J.JopLn(dom=spc, rng=spc, df=J.constdiag_df, df_adj=J.constdiag_df_adj, s={"a": 0.5})]]
A = J.blockop(rows)
Ad = J.blockop([[J.JopDiagonal(coeff.arrays[i])] for i in range(nrow)])
m = J.rand(spc, seed=2, stream=0)
d = J.rand(J.JetBSpace([spc] * nrow), seed=3, stream=0)
w = J.zeros(spc)
out = C.c_double(0)
J.tune(step_chain=0)


def timed(fn, reps=7):
    fn(); fn()
    best = 1e9
    for _ in range(reps):
        e0 = J.Event().record()
        fn()
        e1 = J.Event().record()
        best = min(best, e0.elapsed_ms(e1))
    return best


blk = n * 4
for name, op, nd in (("all diagonal", Ad, nrow), ("data + zero + lambda*I", A, nrow - 2)):
    nat = _blk._tall_native(op)
    J.op_tune_set(op, "step_mode", 0)
    b_step = (3 * nd + 2 * (nrow - nd)) * blk + 2 * blk
    for shape in ((0, 0, 0), (256, 2, 2), (256, 4, 1), (256, 1, 4), (512, 1, 4)):
        J.tune(adj_wg=shape[0], adj_unroll=shape[1], adj_depth=shape[2])
        t = timed(lambda: check(lib.jh_blockop_bidiag_step(nat.handle, d.handle, m.handle, w.handle, 1.0, -0.5, C.byref(out))))
        print(f"{nrow} x {edge}^3 {name:24s} shape {shape}: {t:7.3f} ms {b_step / t / 1e6:7.1f} GB/s  parts {J.tune_get('last_adj_parts')}", flush=True)
    J.tune(adj_wg=0, adj_unroll=0, adj_depth=0)
