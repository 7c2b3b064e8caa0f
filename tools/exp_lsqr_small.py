#!/usr/bin/env python3
"""LSQR iteration cost on small and mid-size operators: the graph-replayed loop with device-resident recurrences (jh_lsqr_solve below
1 GiB per pass), the host loop in C++ behind the ABI (tune lsqr_graph=0) and the Python driver over the same kernels (JETS_LSQR_NATIVE=0).  The kernel-only figure is 3*N*n*4 bytes at 6 TB/s."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jets_jl_amd as J
J.init(0)
for nrow, edge in ((64, 128), (16, 128), (64, 64), (256, 128)):
    blk = J.JetSpace(np.float32, edge, edge, edge)
    coeff = J.rand(J.JetBSpace([blk] * nrow), seed=1, stream=0)
    A = J.blockop([[J.JopDiagonal(c)] for c in coeff.arrays])
    x_true = J.rand(J.domain(A), seed=4, stream=0)
    b = J.mul(A, x_true)
    out = []
    for native, graph in (("1", 1), ("1", 0), ("0", 0)):               # graph-replayed device loop / host loop in C++ / Python driver
        os.environ["JETS_LSQR_NATIVE"] = native
        J.tune(lsqr_graph=graph)
        J.lsqr(A, b, maxiter=5, atol=0, btol=0, force_maxiter=True)
        J.synchronize()
        t0 = time.perf_counter()
        res = J.lsqr(A, b, maxiter=100, atol=0, btol=0, force_maxiter=True)
        J.synchronize()
        out.append((1e3 * (time.perf_counter() - t0) / res.itn, J.tune_get("last_lsqr_graph")))
    J.tune(lsqr_graph=1)
    n = edge ** 3
    print(f"{nrow:4d} x {edge}^3: graph-replayed {out[0][0]:.3f} ms/iteration ({out[0][1]} replays), host loop in C++ {out[1][0]:.3f}, Python driver {out[2][0]:.3f}; "
          f"kernel only (3Nn at 6 TB/s) {3 * nrow * n * 4 / 6e12 * 1e3:.3f} ms", flush=True)
