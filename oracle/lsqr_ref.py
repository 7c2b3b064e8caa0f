"""fp64 CPU LSQR (TEST INFRASTRUCTURE ONLY) -- the checker for jets.jl_amd/lsqr.py.

The reference's solver caller is IterativeSolvers.jl `lsqr` (src/Jets.jl:1143-1152), which is neither
vendored nor version-pinned, so LSQR parity cannot be anchored on reference code.  This is the textbook
algorithm of Paige & Saunders (ACM TOMS 8(1), 1982, section 4 and the stopping rules of section 6), with
explicit normalisation of both Lanczos vectors, in numpy float64, on matvec / rmatvec callables.
"""
from __future__ import annotations

import math

import numpy as np


def lsqr_fp64(matvec, rmatvec, b, n, x0=None, damp=0.0, atol=1e-6, btol=1e-6, conlim=1e8, maxiter=100):
    b = np.asarray(b, dtype=np.float64 if not np.iscomplexobj(b) else np.complex128)
    x = np.zeros(n, dtype=b.dtype) if x0 is None else np.array(x0, dtype=b.dtype)
    u = b.copy() if x0 is None else b - matvec(x)
    bnorm = float(np.linalg.norm(b))
    beta = float(np.linalg.norm(u))
    history = []
    itn = istop = 0
    anorm = acond = ddnorm = res2 = xnorm = xxnorm = z = 0.0
    cs2, sn2 = -1.0, 0.0
    if beta > 0:
        u = u / beta
        v = rmatvec(u)
        alpha = float(np.linalg.norm(v))
    else:
        v = x.copy()
        alpha = 0.0
    if alpha > 0:
        v = v / alpha
    w = v.copy()
    rhobar, phibar = alpha, beta
    rnorm = r1norm = r2norm = beta
    arnorm = alpha * beta
    if arnorm == 0:
        return x, dict(istop=0, itn=0, r1norm=r1norm, history=history)
    eps = np.finfo(np.float64).eps
    ctol = 1.0 / conlim if conlim > 0 else 0.0
    while itn < maxiter:
        itn += 1
        u = matvec(v) - alpha * u
        beta = float(np.linalg.norm(u))
        if beta > 0:
            u = u / beta
            anorm = math.sqrt(anorm ** 2 + alpha ** 2 + beta ** 2 + damp ** 2)
            v = rmatvec(u) - beta * v
            alpha = float(np.linalg.norm(v))
            if alpha > 0:
                v = v / alpha
        rhobar1 = math.sqrt(rhobar ** 2 + damp ** 2)
        cs1, sn1 = rhobar / rhobar1, damp / rhobar1
        psi = sn1 * phibar
        phibar = cs1 * phibar
        rho = math.sqrt(rhobar1 ** 2 + beta ** 2)
        cs, sn = rhobar1 / rho, beta / rho
        theta = sn * alpha
        rhobar = -cs * alpha
        phi = cs * phibar
        phibar = sn * phibar
        tau = sn * phi
        t1, t2 = phi / rho, -theta / rho
        dk = w / rho
        x = x + t1 * w
        w = v + t2 * w
        ddnorm += float(np.linalg.norm(dk)) ** 2
        delta = sn2 * rho
        gambar = -cs2 * rho
        rhs = phi - delta * z
        zbar = rhs / gambar
        xnorm = math.sqrt(xxnorm + zbar ** 2)
        gamma = math.sqrt(gambar ** 2 + theta ** 2)
        cs2, sn2 = gambar / gamma, theta / gamma
        z = rhs / gamma
        xxnorm += z ** 2
        acond = anorm * math.sqrt(ddnorm)
        res1 = phibar ** 2
        res2 += psi ** 2
        rnorm = math.sqrt(res1 + res2)
        arnorm = alpha * abs(tau)
        r1sq = rnorm ** 2 - damp ** 2 * xxnorm
        r1norm = math.sqrt(abs(r1sq)) * (1 if r1sq >= 0 else -1)
        history.append((itn, r1norm, arnorm))
        test1 = rnorm / bnorm if bnorm > 0 else 0.0
        test2 = arnorm / (anorm * rnorm + eps) if rnorm > 0 else 0.0
        test3 = 1.0 / (acond + eps)
        t1_ = test1 / (1 + anorm * xnorm / bnorm) if bnorm > 0 else 0.0
        rtol = btol + atol * anorm * xnorm / bnorm if bnorm > 0 else 0.0
        if itn >= maxiter:
            istop = 7
        if 1 + test3 <= 1:
            istop = 6
        if 1 + test2 <= 1:
            istop = 5
        if 1 + t1_ <= 1:
            istop = 4
        if test3 <= ctol:
            istop = 3
        if test2 <= atol:
            istop = 2
        if test1 <= rtol:
            istop = 1
        if istop:
            break
    return x, dict(istop=istop, itn=itn, r1norm=r1norm, anorm=anorm, acond=acond, arnorm=arnorm, xnorm=xnorm, history=history)
