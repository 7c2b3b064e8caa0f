#!/usr/bin/env python3
"""EXPERIMENT: rows off the 16-byte grid with temporal instead of nontemporal loads (knob ua_nt: 1 streamed, 0 temporal): forward, adjoint and the alternating
pair, two rounds in one process (one child adjointed, so that an ALIGNED size also runs the MIXED instantiations).    python tools/exp_ua_nt.py NROW EDGE"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
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
A = J.blockop([[J.JopDiagonal(g)] if i else [J.JopDiagonal(g).H] for i, g in enumerate(diags)])   # (one adjointed child: the MIXED kernels also at an aligned size)
m = J.rand(spc, seed=2, stream=0); d = J.rand(J.range(A), seed=3, stream=0); mt = J.zeros(spc)
by = (2 * nrow + 1) * n * 4
for rnd in range(2):
    for k in (1, 0):
        J.tune(ua_nt=k)
        tf = timed(lambda: J.mul_(d, A, m)); ta = timed(lambda: J.mul_(mt, A.H, d))
        tp = timed(lambda: (J.mul_(d, A, m), J.mul_(mt, A.H, d)))
        print(f"{nrow} x {e}^3 ua_nt={k}: pair {tp:7.3f} ms | forward {tf:7.3f} ms {by / tf / 1e6:6.0f} GB/s | adjoint {ta:7.3f} ms {by / ta / 1e6:6.0f} GB/s", flush=True)
