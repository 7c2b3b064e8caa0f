#!/usr/bin/env python3
"""Block-vector primitives on slabs whose length is not a multiple of 16 bytes, and on views that start off a 16-byte boundary (block 1 of a vector of odd
blocks): GB/s against the aligned neighbour.     python tools/bench_vec_unaligned.py [EDGE] [NBLOCKS]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import jets_jl_amd as J

J.init(0)
edge = int(sys.argv[1]) if len(sys.argv) > 1 else 255
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 15


def timed(fn, reps=5):
    fn(); fn()
    best = 1e9
    for _ in range(reps):
        e0 = J.Event().record(); fn(); e1 = J.Event().record()
        best = min(best, e0.elapsed_ms(e1))
    return best


for e in (edge, edge + 1):
    spc = J.JetSpace("float32", e, e, e)
    R = J.JetBSpace([spc] * nb)
    d, f, g = J.rand(R, seed=1, stream=0), J.rand(R, seed=2, stream=0), J.zeros(R)
    by = R.length() * 4
    rows = [("f .= a*d .+ b*e  [lincomb]", 3 * by, lambda: J.lincomb_(g, [0.5, 2.0], [d, f])),
            ("g .= d .* f      [hadamard]", 3 * by, lambda: J.hadamard_(g, d, f)),
            ("broadcast exp(-d*d)*f [JIT]", 3 * by, lambda: J.broadcast_(g, "exp(-x0*x0)*x1", [d, f], ())),
            ("norm(d)", by, lambda: J.norm(d)), ("dot(d, f)", 2 * by, lambda: J.dot(d, f)), ("fill!(g, 3)", by, lambda: J.fill_(g, 3.0)),
            ("copyto!(g, d)", 2 * by, lambda: J.copyto_(g, d))]
    for name, b, fn in rows:
        t = timed(fn)
        print(f"{nb} x {e}^3 whole vector   {name:34s} {t:8.3f} ms {b / t / 1e6:6.0f} GB/s", flush=True)
    d1, f1, g1 = d.arrays[1], f.arrays[1], g.arrays[1]
    by1 = e ** 3 * 4
    rows = [("block 1: g1 .= a*d1 .+ b*f1", 3 * by1, lambda: J.lincomb_(g1, [0.5, 2.0], [d1, f1])),
            ("block 1: broadcast [JIT]", 3 * by1, lambda: J.broadcast_(g1, "exp(-x0*x0)*x1", [d1, f1], ())),
            ("block 1: norm", by1, lambda: J.norm(d1)), ("block 1: setblock!(g, 1, d1)", 2 * by1, lambda: J.setblock_(g, 1, d1)),
            ("block 1: fill!", by1, lambda: J.fill_(g1, 3.0))]
    for name, b, fn in rows:
        t = timed(fn)
        print(f"{nb} x {e}^3 one block      {name:34s} {t:8.3f} ms {b / t / 1e6:6.0f} GB/s", flush=True)
