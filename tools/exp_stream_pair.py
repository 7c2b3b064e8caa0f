#!/usr/bin/env python3
"""Jets.stream_pair at the headline size: the probe's two figures, then the pair rate of the operator built on the kept order (after the forward's
per-operator measurement).  Run several times in a row: consecutive processes land differently.

    python tools/exp_stream_pair.py [nrow] [edge]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jets_jl_amd as J

nrow = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
edge = int(sys.argv[2]) if len(sys.argv) > 2 else 256
J.init(0)
blk = J.JetSpace(np.float32, edge, edge, edge)
R = J.JetBSpace([blk] * nrow)
t0 = time.perf_counter()
coeff, d, info = J.stream_pair(R, candidates=int(os.environ.get('CANDIDATES', '2')))
J.synchronize()
t_probe = time.perf_counter() - t0
J.rand_(coeff, seed=1, stream=0)
J.rand_(d, seed=3, stream=0)
m, mt = J.rand(blk, seed=2, stream=0), J.zeros(blk)
A = J.blockop([[J.JopDiagonal(c)] for c in coeff.arrays])
k = 0
J.mul_(d, A, m)
while J.op_tune_get(A, "fwd_walk") == -1 and 0 < J.op_tune_get(A, "fwd_trials") and k < 32:
    J.mul_(d, A, m)
    J.synchronize()
    k += 1
for _ in range(3):
    J.mul_(d, A, m)
    J.mul_(mt, A.H, d)
J.synchronize()
reps, tf, ta = 10, 0.0, 0.0
e = [J.Event() for _ in range(3)]
for _ in range(reps):
    e[0].record(); J.mul_(d, A, m); e[1].record(); J.mul_(mt, A.H, d); e[2].record()
    J.synchronize()
    tf += e[0].elapsed_ms(e[1]); ta += e[1].elapsed_ms(e[2])
tf, ta = tf / reps, ta / reps
print(f"{nrow} x {edge}^3: probe {info} in {t_probe:.2f} s (allocation included) -> forward {tf:7.3f} ms  adjoint {ta:7.3f} ms  pair {tf + ta:7.3f} ms = "
      f"{1e3 / (tf + ta):6.3f} pairs/s (walk {J.op_tune_get(A, 'fwd_walk')})", flush=True)
