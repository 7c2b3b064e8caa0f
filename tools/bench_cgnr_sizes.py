#!/usr/bin/env python3
"""CG on the normal equations through the fused A'A at several operator sizes: ms per iteration against the time of the fused A'A
alone (what an iteration must spend) -- the rest is the domain-side vector work and the host's scalar round trips.

    python tools/bench_cgnr_sizes.py [CG_DEV [LSQR_GRAPH]] > profiles/bench_cgnr_sizes_r04.txt     (knobs cg_dev / lsqr_graph; 2 = the device-resident loops at every size)
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jets_jl_amd as J

J.init(0)
if len(sys.argv) > 1:
    J.tune(cg_dev=int(sys.argv[1]))                               # 2: the device-resident CG loops at every size
if len(sys.argv) > 2:
    J.tune(lsqr_graph=int(sys.argv[2]))                           # 2: LSQR's too
print("# rows x block (Float32)   fused A'A alone   CGNR per iteration   LSQR per iteration   CGLS per iteration", flush=True)
for nrow, edge in ((64, 64), (256, 64), (64, 128), (256, 128), (1024, 128), (256, 256), (1024, 256)):
    blk = J.JetSpace(np.float32, edge, edge, edge)
    coeff = J.rand(J.JetBSpace([blk] * nrow), seed=1, stream=0)
    A = J.blockop([[J.JopDiagonal(c)] for c in coeff.arrays])
    x_true = J.rand(J.domain(A), seed=4, stream=0)
    b = J.mul(A, x_true)
    y = J.zeros(J.domain(A))
    N = A.H @ A
    for _ in range(3):
        J.mul_(y, N, x_true)
    J.synchronize()
    e0 = J.Event().record()
    reps = 20
    for _ in range(reps):
        J.mul_(y, N, x_true)
    e1 = J.Event().record()
    t_n = e0.elapsed_ms(e1) / reps
    out, marg = [], []
    small = 3.0 * nrow * edge ** 3 * 4 < (1 << 30)             # the sizes whose loops keep their recurrences on the device (graph-replayed)
    for solve in (lambda k: J.cgnr(A, b, maxiter=k, atol=0.0, btol=0.0, force_maxiter=True),
                  lambda k: J.lsqr(A, b, maxiter=k, atol=0.0, btol=0.0, conlim=0.0, force_maxiter=True, overwrite_b=True),
                  lambda k: J.cgls(A, b, maxiter=k, atol=0.0, btol=0.0, force_maxiter=True, overwrite_b=True)):
        took = {}
        hi = 112 if small else 36
        for iters in (12, 12, hi):   # 12 iterations twice (the first run carries the lazy per-operator measurements), then 112
            J.mul_(b, A, x_true)                               # LSQR / CGLS use b's storage (no range-sized allocation inside the timed solve)
            J.synchronize()
            t0 = time.perf_counter()
            r = solve(iters)
            J.synchronize()
            took[iters] = (time.perf_counter() - t0, r.itn)
        out.append(1e3 * took[12][0] / max(took[12][1], 1))
        # what ONE MORE iteration costs (set-up -- work vectors, ||b||, A'b, graph capture -- cancels): (t(112) - t(12)) / 100
        marg.append(1e3 * (took[hi][0] - took[12][0]) / max(took[hi][1] - took[12][1], 1) if took[hi][1] > took[12][1] else float("nan"))
    print(f"{nrow:5d} x {edge}^3   {t_n:9.3f} ms   {out[0]:9.3f} ms ({out[0] / t_n:4.2f}x)   {out[1]:9.3f} ms   {out[2]:9.3f} ms"
          + f"   | per further iteration: CGNR {1e3 * marg[0]:8.1f} us  LSQR {1e3 * marg[1]:8.1f} us  CGLS {1e3 * marg[2]:8.1f} us", flush=True)
    J.close(A)
    del A, coeff, N, b, y, x_true
    import gc
    gc.collect()
