#!/usr/bin/env python3
"""LSQR (the Python driver over mul!) on tall operators of dense children: ms per iteration against the two matrix passes an
iteration needs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jets_jl_amd as J
J.init(0)
for nrow, k in ((64, 2048), (1024, 512), (4096, 256)):
    mats = [J.rand(J.JetSpace(np.float32, k, k), seed=1, stream=i) for i in range(nrow)]
    A = J.blockop([[J.JopDense(M)] for M in mats])
    x_true = J.rand(J.domain(A), seed=4, stream=0)
    b = A * x_true
    J.lsqr(A, b, maxiter=3, atol=0, btol=0, force_maxiter=True)
    J.synchronize(); t0 = time.perf_counter()
    res = J.lsqr(A, b, maxiter=30, atol=0, btol=0, force_maxiter=True)
    J.synchronize(); t = (time.perf_counter() - t0) / res.itn
    nbytes = nrow * k * k * 4
    print(f"{nrow} x 1 of {k}^2 dense: {1e3 * t:.3f} ms/iteration; two matrix passes at 6 TB/s = {2 * nbytes / 6e12 * 1e3:.3f} ms", flush=True)
    del A, mats
