"""fp64 CPU CGLS (TEST INFRASTRUCTURE ONLY) -- the checker for jets.jl_amd/cgls.py and jh_cgls_solve.

SURVEY.md section 8 f-1 names "LSQR/CGLS"; like LSQR, CGLS has no counterpart inside Jets.jl (src/Jets.jl:1143-1152 points its
users at IterativeSolvers.jl, un-vendored), so parity cannot be anchored on reference code.  This is the TEXTBOOK recurrence
(Hestenes & Stiefel 1952; Bjorck, "Numerical Methods for Least Squares Problems", 1996, algorithm 7.4.1) in numpy float64 on
matvec / rmatvec callables, with the explicit q = A p vector -- deliberately NOT the two-pass form the product uses
(||A p||^2 as <p, A'A p>), so the two only agree if that rewrite is right.
"""
from __future__ import annotations

import math

import numpy as np


def cgls_fp64(matvec, rmatvec, b, n, x0=None, damp=0.0, atol=1e-6, btol=1e-6, maxiter=100):
    b = np.asarray(b, dtype=np.float64 if not np.iscomplexobj(b) else np.complex128)
    x = np.zeros(n, dtype=b.dtype) if x0 is None else np.array(x0, dtype=b.dtype)
    r = b.copy() if x0 is None else b - matvec(x)
    bnorm = float(np.linalg.norm(b))
    s = rmatvec(r) - damp ** 2 * x
    p = s.copy()
    gamma = float(np.vdot(s, s).real)
    gamma0 = gamma
    history = []
    itn = istop = 0
    if gamma > 0:
        while itn < maxiter:
            itn += 1
            q = matvec(p)
            delta = float(np.vdot(q, q).real) + damp ** 2 * float(np.vdot(p, p).real)
            if not (delta > 0 and math.isfinite(delta)):
                istop, itn = 6, itn - 1
                break
            alpha = gamma / delta
            x = x + alpha * p
            r = r - alpha * q
            s = rmatvec(r) - damp ** 2 * x
            gamma_new = float(np.vdot(s, s).real)
            p = s + (gamma_new / gamma) * p
            gamma = gamma_new
            rnorm, arnorm = float(np.linalg.norm(r)), math.sqrt(gamma)
            history.append((itn, rnorm, arnorm))
            if itn >= maxiter:
                istop = 7
            if arnorm <= atol * math.sqrt(gamma0):
                istop = 2
            if rnorm <= btol * bnorm:
                istop = 1
            if istop:
                break
    return x, dict(istop=istop, itn=itn, r1norm=float(np.linalg.norm(r)), arnorm=math.sqrt(gamma), xnorm=float(np.linalg.norm(x)), history=history)
