#!/usr/bin/env python3
"""LSQR with force_maxiter far past convergence (throughput runs do that): the iterates must stay finite.  python tools/exp_lsqr_past_convergence.py [ITERS]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import jets_jl_amd as J

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
J.init(0)
spc = J.JetSpace("float32", 64, 64, 64)
coeff = J.rand(J.JetBSpace([spc] * 16), seed=1, stream=0)
A = J.blockop([[J.JopDiagonal(c)] for c in coeff.arrays])
x_true = J.rand(spc, seed=4, stream=0)
b = A * x_true
for native in ("1", "0"):
    os.environ["JETS_LSQR_NATIVE"] = native
    res = J.lsqr(A, b, atol=0.0, btol=0.0, conlim=0.0, maxiter=iters, force_maxiter=True)
    x = res.x.to_numpy()
    err = float(np.linalg.norm((x - x_true.to_numpy()).ravel()) / np.linalg.norm(x_true.to_numpy().ravel()))
    first_bad = next((k for k, h in enumerate(res.history) if not (np.isfinite(h[1]) and np.isfinite(h[2]))), None)
    print(f"native={native}: itn {res.itn} istop {res.istop} finite x {bool(np.isfinite(x).all())} rel err {err:.3e} first non-finite history entry {first_bad} last r1norm {res.history[-1][1]:.3e}")
