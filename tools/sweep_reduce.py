#!/usr/bin/env python3
"""Reduction grid size sweep (norm / dot on a 16 GiB Float32 slab)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jets_jl_amd as J
J.init(0)
L = 1 << 32
x = J.rand(J.JetSpace(np.float32, L), seed=1, stream=0)
y = J.rand(J.JetSpace(np.float32, L), seed=2, stream=0)
def timeit(fn, reps=5):
    fn(); fn()
    e0 = J.Event().record()
    for _ in range(reps): fn()
    e1 = J.Event().record()
    return e0.elapsed_ms(e1) / reps
for rnd in range(2):
    for g in (1024, 2048, 4096, 8192, 16384, 65536, 262144):
        J.tune(red_wgs=g)
        tn, td = timeit(lambda: J.norm(x)), timeit(lambda: J.dot(x, y))
        print(f"red_wgs={g:7d} norm {tn:7.3f} ms {L*4/tn/1e6:8.1f} GB/s | dot {td:7.3f} ms {2*L*4/td/1e6:8.1f} GB/s")
