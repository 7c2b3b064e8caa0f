#!/usr/bin/env python3
"""Banded forward walks (k row groups in flight, fwd_order = k) across re-allocations of the range vector: is there a band
width that is both faster than the sequential sweep and immune to the placement lottery of the all-rows walk?"""
import gc
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jets_jl_amd as J

J.init(0)
J.tune(autotune=0)
edge, N = 256, int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n = edge ** 3
blk = J.JetSpace(np.float32, edge, edge, edge)
coeff = J.rand(J.JetBSpace([blk] * N), seed=1, stream=0)
A = J.blockop([[J.JopDiagonal(c)] for c in coeff.arrays])
m = J.rand(blk, seed=2, stream=0)


def timed(fn, reps=3):
    fn(); fn()
    best = 1e9
    for _ in range(reps):
        e0 = J.Event().record(); fn(); e1 = J.Event().record()
        best = min(best, e0.elapsed_ms(e1))
    return best


cfgs = [("seq 1024x8x16", dict(fwd_wg=1024, fwd_unroll=8, fwd_group=16, fwd_order=0)),
        ("all 512x1x2", dict(fwd_wg=512, fwd_unroll=1, fwd_group=2, fwd_order=1))]
for (wg, u, g) in ((1024, 8, 16), (256, 4, 16), (512, 1, 2), (512, 4, 8)):
    for k in (2, 4, 8, 16):
        cfgs.append((f"b{k} {wg}x{u}x{g}", dict(fwd_wg=wg, fwd_unroll=u, fwd_group=g, fwd_order=k)))
print("alloc# " + " | ".join(f"{name:>14s}" for name, _ in cfgs))
for it in range(6):
    d = J.zeros(J.JetBSpace([blk] * N))
    row = []
    for name, cfg in cfgs:
        J.tune(**cfg)
        row.append(timed(lambda: J.mul_(d, A, m)))
    print(f"{it:5d}  " + " | ".join(f"{t:14.3f}" for t in row), flush=True)
    del d
    gc.collect()
