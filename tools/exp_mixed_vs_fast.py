#!/usr/bin/env python3
"""EXPERIMENT: the all-diagonal fast path against the MIXED instantiations on the SAME aligned operator (one child adjointed: real elements, same arithmetic),
forward, adjoint and the alternating pair, alternating in one process.     python tools/exp_mixed_vs_fast.py NROW EDGE"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jets_jl_amd as J
J.init(0)
nrow, e = int(sys.argv[1]), int(sys.argv[2])
def timed(fn, reps=5):
    fn(); fn()
    best = 1e9
    for _ in range(reps):
        e0 = J.Event().record(); fn(); e1 = J.Event().record()
        best = min(best, e0.elapsed_ms(e1))
    return best
spc = J.JetSpace("float32", e, e, e); n = e ** 3
diags = J.rand(J.JetBSpace([spc] * nrow), seed=1, stream=0).arrays
A = J.blockop([[J.JopDiagonal(g)] for g in diags])
B = J.blockop([[J.JopDiagonal(g)] if i else [J.JopDiagonal(g).H] for i, g in enumerate(diags)])
m = J.rand(spc, seed=2, stream=0); d = J.rand(J.range(A), seed=3, stream=0); mt = J.zeros(spc)
by = (2 * nrow + 1) * n * 4
for _ in range(24):                      # the fast path's lazy walk search
    J.mul_(d, A, m)
for rnd in range(3):
    for name, op in (("fast path", A), ("MIXED    ", B)):
        tf = timed(lambda: J.mul_(d, op, m)); ta = timed(lambda: J.mul_(mt, op.H, d)); tp = timed(lambda: (J.mul_(d, op, m), J.mul_(mt, op.H, d)))
        print(f"{nrow} x {e}^3 {name}: pair {tp:7.3f} ms | forward {tf:7.3f} ms {by / tf / 1e6:6.0f} GB/s | adjoint {ta:7.3f} ms {by / ta / 1e6:6.0f} GB/s", flush=True)
