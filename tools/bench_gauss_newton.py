#!/usr/bin/env python3
"""Application-level check of the path: Gauss-Newton on a tall NONLINEAR block operator F(m) = [f_i.(m)]_i (the seismic-shot
layout: many block rows, one block column), every product on the device:
    r = d_obs - F(m)            JetBlock_f!  (one JIT-fused pass per child)
    J = jacobian!(F, m)         block point!: each child refreshes its diagonal in place (one fused pass per child)
    dm = lsqr(J, r)             the native LSQR: ONE pass over J and u per iteration (J is all-DIAG for the device)
    m += dm
Prints time per outer iteration and per LSQR iteration, and the error of m.

    python tools/bench_gauss_newton.py [NROWS EDGE]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jets_jl_amd as J

nrow = int(sys.argv[1]) if len(sys.argv) > 1 else 256
edge = int(sys.argv[2]) if len(sys.argv) > 2 else 256
J.init(0)
R = J.JetSpace(np.float32, edge, edge, edge)
n = edge ** 3
# children alternate between three elementwise nonlinearities with different parameters
kinds = [("s0*x0 + s1*x0*x0", "s0 + 2*s1*x0"), ("s0*tanh(s1*x0)", "s0*s1*(1 - tanh(s1*x0)*tanh(s1*x0))"), ("s0*x0*x0*x0 + s1*x0", "3*s0*x0*x0 + s1")]
F = J.blockop([[J.JopElementwise(R, *kinds[i % 3], [0.5 + 0.01 * i, 1.0 + 0.003 * i])] for i in range(nrow)])
x_true = J.rand(R, seed=4, stream=0)
d_obs = F * x_true
m = J.zeros(R)
J.fill_(m, 0.4)
r = J.zeros(J.range(F))
gb = nrow * n * 4 / 1e9
print(f"# Gauss-Newton, {nrow} x 1 nonlinear block operator of {edge}^3 Float32 blocks ({gb:.1f} GB per range vector)")
for it in range(6):
    J.synchronize(); t0 = time.perf_counter()
    J.mul_(r, F, m)                                   # F(m)
    J.lincomb_(r, [1.0, -1.0], [d_obs, r])            # r = d_obs - F(m)
    rn = float(J.norm(r))
    J.synchronize(); t1 = time.perf_counter()
    Jm = J.jacobian_(F, m)                            # point!
    J.synchronize(); t2 = time.perf_counter()
    res = J.lsqr(Jm, r, maxiter=12, atol=1e-6, btol=1e-6, overwrite_b=True)
    J.synchronize(); t3 = time.perf_counter()
    J.lincomb_(m, [1.0, 1.0], [m, res.x])
    err = J.zeros(R)
    J.lincomb_(err, [1.0, -1.0], [m, x_true])
    rel = float(J.norm(err)) / float(J.norm(x_true))
    print(f"outer {it}: ||r|| = {rn:10.4e}   residual {1e3 * (t1 - t0):7.2f} ms   point! {1e3 * (t2 - t1):7.2f} ms   "
          f"LSQR {res.itn:2d} its {1e3 * (t3 - t2):8.2f} ms ({1e3 * (t3 - t2) / max(res.itn, 1):6.2f} ms/it, {3 * gb / max((t3 - t2) / max(res.itn, 1), 1e-9) / 1e3:5.2f} TB/s of 3Nn)   "
          f"||m - x_true|| / ||x_true|| = {rel:.2e}", flush=True)
