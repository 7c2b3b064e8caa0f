#!/usr/bin/env python3
"""How long does the host take to see that a kernel has finished?  A loop of (fused A'A, then a reduction whose scalar the host reads)
against the kernels' own time, for a short and a long kernel; run once as is and once with HSA_ENABLE_INTERRUPT=0 (polling waits).

    python tools/exp_sync_latency.py ; HSA_ENABLE_INTERRUPT=0 python tools/exp_sync_latency.py
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jets_jl_amd as J

J.init(0)
print(f"# HSA_ENABLE_INTERRUPT={os.environ.get('HSA_ENABLE_INTERRUPT', '(unset)')}", flush=True)
for nrow, edge in ((64, 128), (256, 256), (1024, 256)):
    blk = J.JetSpace(np.float32, edge, edge, edge)
    coeff = J.rand(J.JetBSpace([blk] * nrow), seed=1, stream=0)
    A = J.blockop([[J.JopDiagonal(c)] for c in coeff.arrays])
    m = J.rand(J.domain(A), seed=4, stream=0)
    y = J.zeros(J.domain(A))
    N = A.H @ A
    for _ in range(3):
        J.mul_(y, N, m)
    J.synchronize()
    reps = 20
    e0 = J.Event().record()
    for _ in range(reps):
        J.mul_(y, N, m)
    e1 = J.Event().record()
    t_k = e0.elapsed_ms(e1) / reps
    J.dot(y, m)
    e0 = J.Event().record()
    for _ in range(reps):
        J.dot(y, m)
    e1 = J.Event().record()
    t_d = e0.elapsed_ms(e1) / reps
    J.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        J.mul_(y, N, m)
        J.dot(y, m)
    t_loop = 1e3 * (time.perf_counter() - t0) / reps
    print(f"{nrow:5d} x {edge}^3: A'A {t_k:8.3f} ms, dot (with its read-back, back to back) {t_d:6.3f} ms, loop of both {t_loop:8.3f} ms per turn "
          f"=> {1e3 * (t_loop - t_k - t_d):7.1f} us lost per turn", flush=True)
    J.close(A)
    del A, coeff, N, y, m
    import gc
    gc.collect()
