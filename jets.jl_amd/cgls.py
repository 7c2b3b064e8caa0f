"""CGLS over the Jets operator API ("LSQR/CGLS solver driver").

Conjugate gradients on the normal equations (Hestenes & Stiefel 1952; Bjorck 1996, algorithm 7.4.1): min ||A x - b||^2
(+ damp^2 ||x||^2).  Like LSQR it has no counterpart inside Jets.jl (src/Jets.jl:1143-1152 points its users at
IterativeSolvers.jl, un-vendored): the published recurrence, checked against the fp64 CPU CGLS of oracle/cgls_ref.py.

For a device-native tall block operator the whole loop runs behind the C ABI (jh_cgls_solve / _partitioned / _team): TWO
passes per iteration and no range-sized temporary -- ||A p||^2 = <p, A'A p> through the fused normal operator (N n s bytes),
then r <- r - alpha A p, ||r||^2 and A'r in one pass of the Golub-Kahan step kernel (3 N n s).  Any other operator runs the
textbook loop (q = A p kept in a range vector) over the same engines as LSQR: the two fused halves where they exist
(jh_blockop_mul_axpby / jh_blockop_mul_adj_axpby), plain mul! otherwise; row-partitioned and team operators included.
"""
from __future__ import annotations

import builtins
import ctypes as C
import math
import os

from ._ffi import lib, check, JetsHipError
from .arrays import reshape
from . import jets as _j
from .lsqr import LsqrResult, _Engine, _ShardEngine, _TeamEngine, _unwrap_vec

__all__ = ["cgls", "cgls_core", "cgnr", "cgnr_core"]


def cgls(A, b, x0=None, damp: float = 0.0, atol: float = 1e-6, btol: float = 1e-6, maxiter: int = 100, overwrite_b: bool = False,
         force_maxiter: bool = False) -> LsqrResult:
    """x, info = cgls(A, b).  `b` lives in range(A), the result in domain(A); `A` may be a rowpart.RowPartitionedOp (b = this
    rank's rows, every rank gets the same x) or a rowpart.TeamOp (b, x: TeamVec).  istop: 1 ||r|| <= btol ||b||, 2 ||A'r - damp^2 x||
    <= atol times its starting value, 6 breakdown, 7 maxiter.  The record is LSQR's (r1norm = ||r||, arnorm = ||A'r - damp^2 x||,
    history = (itn, ||r||, ||A'r - damp^2 x||) per iteration; anorm = acond = 0).  `overwrite_b=True`: b's storage becomes r."""
    from .rowpart import RowPartitionedOp, TeamOp

    if isinstance(A, TeamOp):
        eng = _TeamEngine(A)
        native = _native_team(eng, b, x0, damp, atol, btol, maxiter, overwrite_b, force_maxiter)
        return native if native is not None else cgls_core(eng, b, x0, damp, atol, btol, maxiter, overwrite_b, force_maxiter)
    if isinstance(A, RowPartitionedOp):
        eng = _ShardEngine(A)
        dom, rng = _j.domain(A.local_op), _j.range_(A.local_op)
    else:
        A = _unwrap_vec(A)
        eng = _Engine(A)
        dom, rng = _j.domain(A), _j.range_(A)
    b = reshape(b, rng)
    x0 = None if x0 is None else reshape(x0, dom)
    native = _native(eng, b, x0, damp, atol, btol, maxiter, overwrite_b, force_maxiter)
    return native if native is not None else cgls_core(eng, b, x0, damp, atol, btol, maxiter, overwrite_b, force_maxiter)


def _result(x, res, hist):
    history = [(k + 1, hist[2 * k], hist[2 * k + 1]) for k in builtins.range(res.itn)]
    return LsqrResult(x, res.istop, res.itn, res.r1norm, res.r2norm, res.anorm, res.acond, res.arnorm, res.xnorm, history)


def _native(eng, b, x0, damp, atol, btol, maxiter, overwrite_b, force_maxiter):
    """jh_cgls_solve (one GPU) / jh_cgls_solve_partitioned (the ABI's own RCCL communicator).  None when it does not apply."""
    from ._ffi import LsqrResultC
    from .rowpart import AbiComm

    if os.environ.get("JETS_CGLS_NATIVE", "1") == "0" or eng.native is None:
        return None
    shard = getattr(eng, "shard", None)
    if shard is not None and not (isinstance(shard.comm, AbiComm) or shard.comm.world == 1):
        return None
    x = eng.zeros_dom() if x0 is None else eng.copy(eng.zeros_dom(), x0)
    u = b if overwrite_b else (eng.copy_of_rng(b) if hasattr(eng, "copy_of_rng") else eng.copy(eng.zeros_rng(), b))
    res = LsqrResultC()
    hist = (C.c_double * builtins.max(2 * int(maxiter), 1))()
    try:
        solve = lib.jh_cgls_solve_partitioned if shard is not None else lib.jh_cgls_solve
        check(solve(eng.native.handle, u.handle, x.handle, 0 if x0 is None else 1, float(damp), float(atol), float(btol), int(maxiter),
                    1 if force_maxiter else 0, C.byref(res), hist))
    except JetsHipError as e:
        if e.status != 4:                                       # JH_ERR_UNSUPPORTED comes before anything is touched: the generic loop
            raise
        return None
    return _result(x, res, hist)


def _native_team(eng, b, x0, damp, atol, btol, maxiter, overwrite_b, force_maxiter):
    from ._ffi import LsqrResultC

    T = eng.T
    if os.environ.get("JETS_CGLS_NATIVE", "1") == "0" or any(n is None for n in T._natives):
        return None
    M = eng.team.world
    x = eng.zeros_dom() if x0 is None else eng.copy(eng.zeros_dom(), x0)
    u = b if overwrite_b else eng.copy(eng.zeros_rng(), b)
    arr = lambda hs: (C.c_void_p * M)(*[h.value if hasattr(h, "value") else h for h in hs])
    res = LsqrResultC()
    hist = (C.c_double * builtins.max(2 * int(maxiter), 1))()
    try:
        check(lib.jh_cgls_solve_team(M, arr([n.handle for n in T._natives]), arr([u[k].handle for k in builtins.range(M)]),
                                     arr([x[k].handle for k in builtins.range(M)]), 0 if x0 is None else 1, float(damp), float(atol), float(btol),
                                     int(maxiter), 1 if force_maxiter else 0, C.byref(res), hist))
    except JetsHipError as e:
        if e.status != 4:
            raise
        return None
    return _result(x, res, hist)


def cgls_core(eng, b, x0, damp, atol, btol, maxiter, overwrite_b=False, force_maxiter=False) -> LsqrResult:
    """The textbook loop on an engine (lsqr._Engine's interface: zeros_dom / zeros_rng / copy / lincomb / norm_dom / norm_rng and the
    two half-iterations fwd(u, v, alpha, beta) -> ||alpha A v + beta u||, adj(v, u, alpha, beta) -> ||alpha A'u + beta v||)."""
    x = eng.zeros_dom() if x0 is None else eng.copy(eng.zeros_dom(), x0)
    r = b if overwrite_b else eng.copy(eng.zeros_rng(), b)
    bnorm = eng.norm_rng(b)
    rnorm = eng.fwd(r, x, -1.0, 1.0) if x0 is not None else bnorm          # r = b - A x0
    s, p, q = eng.zeros_dom(), eng.zeros_dom(), eng.zeros_rng()
    eng.adj(s, r, 1.0, 0.0)                                                 # s = A'r
    if damp:
        eng.lincomb(s, [1.0, -damp * damp], [s, x])
    gamma = eng.norm_dom(s) ** 2
    gamma0 = gamma
    eng.copy(p, s)
    history, itn, istop = [], 0, 0
    if gamma > 0:
        while itn < maxiter:
            itn += 1
            qn = eng.fwd(q, p, 1.0, 0.0)                                     # q = A p, ||q||
            delta = qn * qn + (damp * eng.norm_dom(p)) ** 2
            if not (delta > 0 and math.isfinite(delta)):
                istop, itn = 6, itn - 1
                break
            alpha = gamma / delta
            eng.lincomb(x, [1.0, alpha], [x, p])
            eng.lincomb(r, [1.0, -alpha], [r, q])
            rnorm = eng.norm_rng(r)
            eng.adj(s, r, 1.0, 0.0)                                         # s = A'r - damp^2 x
            if damp:
                eng.lincomb(s, [1.0, -damp * damp], [s, x])
            gamma_new = eng.norm_dom(s) ** 2
            eng.lincomb(p, [1.0, gamma_new / gamma], [s, p])
            gamma = gamma_new
            arnorm = math.sqrt(gamma)
            history.append((itn, rnorm, arnorm))
            if itn >= maxiter:
                istop = 7
            if arnorm <= atol * math.sqrt(gamma0):
                istop = 2
            if rnorm <= btol * bnorm:
                istop = 1
            if istop and not (force_maxiter and itn < maxiter and gamma > 0):
                break
    xnorm = eng.norm_dom(x)
    return LsqrResult(x, istop, itn, rnorm, math.sqrt(rnorm ** 2 + (damp * xnorm) ** 2), 0.0, 0.0, math.sqrt(gamma), xnorm, history)


# ------------------------------------------------------------------ CG on the normal equations through the fused A'A
def cgnr(A, b, x0=None, damp: float = 0.0, atol: float = 1e-6, btol: float = 1e-6, maxiter: int = 100, force_maxiter: bool = False) -> LsqrResult:
    """Conjugate gradients on (A'A + damp^2 I) x = A'b with the normal operator applied as ONE fused pass (JetComposite (A', A),
    src/Jets.jl:530-534 -> jh_blockop_normal_mul): after one adjoint pass for A'b an iteration reads the coefficients once
    (N n s bytes, a third of the LSQR iteration) and works on domain-sized vectors; b is read once and never written; ||r|| follows
    the exact CG recurrence.  Same iterates as cgls / lsqr in exact arithmetic; the attainable accuracy goes with cond(A)^2 (the
    residual A'r is updated by recurrence in the domain), so: well-conditioned operators, throughput.  A device-native tall block
    operator runs behind the C ABI (jh_cgnr_solve / _partitioned / _team); anything else applies A then A' through the engines.
    The record: r2norm = sqrt(||r||^2 + damp^2 ||x||^2) from the recurrence, r1norm = ||r|| derived from it, arnorm = ||A'r - damp^2 x||."""
    from .rowpart import RowPartitionedOp, TeamOp

    if isinstance(A, TeamOp):
        eng = _TeamEngine(A)
        native = _native_cgnr_team(eng, b, x0, damp, atol, btol, maxiter, force_maxiter)
        return native if native is not None else cgnr_core(eng, b, x0, damp, atol, btol, maxiter, force_maxiter)
    if isinstance(A, RowPartitionedOp):
        eng = _ShardEngine(A)
        dom, rng = _j.domain(A.local_op), _j.range_(A.local_op)
    else:
        A = _unwrap_vec(A)
        eng = _Engine(A)
        dom, rng = _j.domain(A), _j.range_(A)
    b = reshape(b, rng)
    x0 = None if x0 is None else reshape(x0, dom)
    native = _native_cgnr(eng, b, x0, damp, atol, btol, maxiter, force_maxiter)
    return native if native is not None else cgnr_core(eng, b, x0, damp, atol, btol, maxiter, force_maxiter)


def _native_cgnr(eng, b, x0, damp, atol, btol, maxiter, force_maxiter):
    from ._ffi import LsqrResultC
    from .rowpart import AbiComm

    shard = getattr(eng, "shard", None)
    nat = eng.native
    if nat is None and shard is None and hasattr(eng, "A"):
        from . import jetblock as _blk

        nat = _blk._grid_native(eng.A)                 # round 6: N x (2 .. 4) grids of diagonals have a fused A'A too (jh_grid_normal.hip)
    if os.environ.get("JETS_CGLS_NATIVE", "1") == "0" or nat is None:
        return None
    if shard is not None and not (isinstance(shard.comm, AbiComm) or shard.comm.world == 1):
        return None
    x = eng.zeros_dom() if x0 is None else eng.copy(eng.zeros_dom(), x0)
    res = LsqrResultC()
    hist = (C.c_double * builtins.max(2 * int(maxiter), 1))()
    try:
        solve = lib.jh_cgnr_solve_partitioned if shard is not None else lib.jh_cgnr_solve
        check(solve(nat.handle, b.handle, x.handle, 0 if x0 is None else 1, float(damp), float(atol), float(btol), int(maxiter),
                    1 if force_maxiter else 0, C.byref(res), hist))
    except JetsHipError as e:
        if e.status != 4:
            raise
        return None
    return _result(x, res, hist)


def _native_cgnr_team(eng, b, x0, damp, atol, btol, maxiter, force_maxiter):
    from ._ffi import LsqrResultC

    T = eng.T
    if os.environ.get("JETS_CGLS_NATIVE", "1") == "0" or any(n is None for n in T._natives):
        return None
    M = eng.team.world
    x = eng.zeros_dom() if x0 is None else eng.copy(eng.zeros_dom(), x0)
    arr = lambda hs: (C.c_void_p * M)(*[h.value if hasattr(h, "value") else h for h in hs])
    res = LsqrResultC()
    hist = (C.c_double * builtins.max(2 * int(maxiter), 1))()
    try:
        check(lib.jh_cgnr_solve_team(M, arr([n.handle for n in T._natives]), arr([b[k].handle for k in builtins.range(M)]),
                                     arr([x[k].handle for k in builtins.range(M)]), 0 if x0 is None else 1, float(damp), float(atol), float(btol),
                                     int(maxiter), 1 if force_maxiter else 0, C.byref(res), hist))
    except JetsHipError as e:
        if e.status != 4:
            raise
        return None
    return _result(x, res, hist)


def cgnr_core(eng, b, x0, damp, atol, btol, maxiter, force_maxiter=False) -> LsqrResult:
    """The same recurrences on an engine: the normal operator as A then A' (two passes and a range-sized temporary -- operators without
    the fused kernel)."""
    x = eng.zeros_dom() if x0 is None else eng.copy(eng.zeros_dom(), x0)
    bnorm = eng.norm_rng(b)
    s, p, y, q = eng.zeros_dom(), eng.zeros_dom(), eng.zeros_dom(), eng.zeros_rng()
    eng.adj(s, b, 1.0, 0.0)                                                 # s = A'b
    phi = bnorm * bnorm                                                     # ||r||^2 + damp^2 ||x||^2, by recurrence
    if x0 is not None:
        qn = eng.fwd(q, x, 1.0, 0.0)                                         # A x0
        eng.adj(y, q, 1.0, 0.0)                                             # A'A x0
        from .arrays import dot

        xb = dot(x[0], s[0]) if hasattr(x, "members") else dot(x, s)
        phi = phi - 2.0 * float(getattr(xb, "real", xb)) + qn * qn + (damp * eng.norm_dom(x)) ** 2
        eng.lincomb(s, [1.0, -1.0], [s, y])
        if damp:
            eng.lincomb(s, [1.0, -damp * damp], [s, x])
    gamma = eng.norm_dom(s) ** 2
    gamma0 = gamma
    eng.copy(p, s)
    history, itn, istop = [], 0, 0
    if gamma > 0:
        while itn < maxiter:
            itn += 1
            qn = eng.fwd(q, p, 1.0, 0.0)                                     # q = A p ; <p, A'A p> = ||q||^2
            eng.adj(y, q, 1.0, 0.0)                                         # y = A'A p
            delta = qn * qn + (damp * eng.norm_dom(p)) ** 2
            if damp:
                eng.lincomb(y, [1.0, damp * damp], [y, p])
            if not (delta > 0 and math.isfinite(delta)):
                istop, itn = 6, itn - 1
                break
            alpha = gamma / delta
            eng.lincomb(x, [1.0, alpha], [x, p])
            eng.lincomb(s, [1.0, -alpha], [s, y])
            phi = builtins.max(phi - alpha * gamma, 0.0)
            gamma_new = eng.norm_dom(s) ** 2
            eng.lincomb(p, [1.0, gamma_new / gamma], [s, p])
            gamma = gamma_new
            rnorm, arnorm = math.sqrt(phi), math.sqrt(gamma)
            history.append((itn, rnorm, arnorm))
            if itn >= maxiter:
                istop = 7
            if arnorm <= atol * math.sqrt(gamma0):
                istop = 2
            if rnorm <= btol * bnorm:
                istop = 1
            if istop and not (force_maxiter and itn < maxiter and gamma > 0):
                break
    xnorm = eng.norm_dom(x)
    r1sq = phi - (damp * xnorm) ** 2
    return LsqrResult(x, istop, itn, math.sqrt(builtins.max(r1sq, 0.0)), math.sqrt(phi), 0.0, 0.0, math.sqrt(gamma), xnorm, history)
