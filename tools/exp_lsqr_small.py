import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import jets_jl_amd as J
J.init(0)
for nrow, edge in ((64, 128), (16, 128), (64, 64), (256, 128)):
    blk = J.JetSpace(np.float32, edge, edge, edge)
    coeff = J.rand(J.JetBSpace([blk] * nrow), seed=1, stream=0)
    A = J.blockop([[J.JopDiagonal(c)] for c in coeff.arrays])
    x_true = J.rand(J.domain(A), seed=4, stream=0)
    b = J.mul(A, x_true)
    J.lsqr(A, b, maxiter=5, atol=0, btol=0, force_maxiter=True)
    J.synchronize()
    t0 = time.perf_counter()
    res = J.lsqr(A, b, maxiter=100, atol=0, btol=0, force_maxiter=True)
    J.synchronize()
    t = time.perf_counter() - t0
    n = edge ** 3
    print(f"{nrow} x {edge}^3: {1e3 * t / res.itn:.3f} ms/iteration ({res.itn} its); 3Nn bytes at 6 TB/s = {3 * nrow * n * 4 / 6e12 * 1e3:.3f} ms")
