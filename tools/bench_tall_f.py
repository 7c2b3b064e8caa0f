#!/usr/bin/env python3
"""F(m) of a tall NONLINEAR block operator (JetBlock_f!, src/Jets.jl:988-1008) of SQUARE children (the reference's JopBar) and of a mix with linear rows:
the tall tiling (knob tall_f = 1, round 5) against the general one-line kernels (tall_f = 0).  Bytes: every row written once, the model read once.
    python tools/bench_tall_f.py NROW EDGE"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jets_jl_amd as J

J.init(0)
nrow = int(sys.argv[1]) if len(sys.argv) > 1 else 256
edge = int(sys.argv[2]) if len(sys.argv) > 2 else 256
spc = J.JetSpace("float32", edge, edge, edge)
n = edge ** 3


def timed(fn, reps=5):
    fn(); fn()
    best = 1e9
    for _ in range(reps):
        e0 = J.Event().record(); fn(); e1 = J.Event().record()
        best = min(best, e0.elapsed_ms(e1))
    return best


diags = J.rand(J.JetBSpace([spc] * nrow), seed=1, stream=0).arrays
cases = {"all SQUARE children": [[J.JopSquare(spc)] for _ in range(nrow)],
         "SQUARE / diagonal rows alternating": [[J.JopSquare(spc)] if i % 2 else [J.JopDiagonal(diags[i])] for i in range(nrow)]}
m = J.rand(spc, seed=2, stream=0)
for name, rows in cases.items():
    F = J.blockop(rows)
    d = J.zeros(J.range(F))
    ndiag = sum(1 for i in range(nrow) if "alternating" in name and i % 2 == 0)
    by = (nrow + ndiag + 1) * n * 4
    for rnd in range(2):
        for knob in (1, 0):
            J.tune(tall_f=knob)
            t = timed(lambda: J.mul_(d, F, m))
            print(f"{nrow} x {edge}^3 {name:36s} tall_f={knob}: F(m) {t:8.3f} ms {by / t / 1e6:6.0f} GB/s", flush=True)
    J.close(F)
J.tune(tall_f=1)
