"""LSQR over the Jets operator API (BASELINE.json configs[4]).

The reference does not ship a solver: its documented caller is IterativeSolvers.jl's `lsqr(vec(A), vec(d))`
(src/Jets.jl:1143-1152, docs/src/index.md:235-246), an un-vendored package with no pinned version, so this
is a from-scratch implementation of the published algorithm (Paige & Saunders, "LSQR: an algorithm for
sparse linear equations and sparse least squares", ACM TOMS 8(1), 1982), checked against the fp64 CPU
restatement in oracle/lsqr_ref.py (LSQR parity is otherwise unpinned).

Per iteration it needs   u <- A v - alpha u, ||u||   and   v <- A'u - beta v, ||v||.
For a device-native tall block operator BOTH happen in ONE pass over the operator and the range vector
(jh_blockop_bidiag_step): u <- A v - (alpha/beta) u, ||u||^2 and w = A'u (un-normalised) together, then
v <- w/||u|| - ||u|| v on domain-sized vectors (A' is linear, so the normalisation follows the pass).  The big
range-side vector u is never normalised in memory -- its scale 1/beta is carried into the next update.
Algorithmic HBM bytes per iteration for an N x 1 operator of n-element blocks of s bytes: 3*N*n*s (a, u in, u out)
+ ~12*n*s for the domain-side vectors.  JETS_LSQR_FUSED_STEP=0 selects the older schedule of two fused halves
(jh_blockop_mul_axpby 3*N*n*s + jh_blockop_mul_adj_axpby 2*N*n*s).  Any other operator runs the same recurrences
through mul!, one fused broadcast and one norm per half.
"""
from __future__ import annotations

import builtins
import ctypes as C
import math
import os

from ._ffi import lib, check, JetsHipError
from .arrays import zeros, lincomb_, norm, copyto_, reshape, fill_


def _plain(coefs):
    """The solvers' recurrences run in fp64 and their coefficients are applied in the vectors' element type (what the C++ loops of
    jh_lsqr.hip do): plain Python floats, so that a numpy float64 from a norm is not read as Julia's Float64 (promoted arithmetic)."""
    return [float(c) for c in coefs]
from . import jets as _j
from . import jetblock as _blk

__all__ = ["lsqr", "lsqr_core", "LsqrResult"]


class LsqrResult:
    """x plus the convergence record (names follow the LSQR paper / scipy's lsqr)."""

    def __init__(self, x, istop, itn, r1norm, r2norm, anorm, acond, arnorm, xnorm, history):
        self.x, self.istop, self.itn = x, istop, itn
        self.r1norm, self.r2norm, self.anorm, self.acond, self.arnorm, self.xnorm = r1norm, r2norm, anorm, acond, arnorm, xnorm
        self.history = history  # list of (itn, r1norm, arnorm) per iteration

    def __iter__(self):  # x, info = lsqr(...)
        yield self.x
        yield self


def _unwrap_vec(A):
    """vec(A) (src/Jets.jl:1129-1136) just reshapes; run the recurrences on the inner operator's own spaces."""
    j = _j.jet(A)
    if j.f is _j.JetVec_f:
        return j.s["op"]
    return A


class _Engine:
    """Vector algebra + the two half-iterations on one GPU; the halves are fused kernels when the operator is a
    device-native tall block operator.  `lsqr_core` only talks to this interface, so the row-partitioned
    engine below (and the CPU test double of tests/test_rowpart_gloo.py) can replace it."""

    def __init__(self, A):
        self.A = A
        self.L = _j.JopLn(A) if not isinstance(A, _j.JopAdjoint) else A
        self.native = None
        if isinstance(A, _j.JopLn):
            self.native = _blk._tall_native(A)       # None for an un-pointed Jacobian: the generic path then raises like mul! does
        self._tmp_r = None
        self._tmp_d = None
        self.fused_step = os.environ.get("JETS_LSQR_FUSED_STEP", "1") != "0"   # one pass per iteration (jh_blockop_bidiag_step)

    # --- vector algebra
    def zeros_dom(self):
        return zeros(_j.domain(self.A))

    def zeros_rng(self):
        return zeros(_j.range_(self.A))

    def copy_of_rng(self, src):
        """A private copy of a range vector (the solver's u when b must survive): storage without the zero fill, then the copy."""
        from .arrays import Array, ROLE_OUTPUT

        return copyto_(Array(_j.range_(self.A), undef=True, role=ROLE_OUTPUT), src)     # (rewritten in every iteration)

    def copy(self, dst, src):
        return copyto_(dst, src)

    def lincomb(self, dst, coefs, xs):
        return lincomb_(dst, _plain(coefs), xs)

    def norm_dom(self, x) -> float:
        return float(norm(x))

    def norm_rng(self, x) -> float:
        return float(norm(x))

    # --- half-iterations (return the LOCAL sum of squares; `finish` makes it a global norm)
    def _fwd_local(self, u, v, alpha, beta) -> float:
        if self.native is not None:
            out = C.c_double(0)
            try:
                check(lib.jh_blockop_mul_axpby(self.native.handle, u.handle, v.handle, float(alpha), float(beta), C.byref(out)))
                return out.value
            except JetsHipError as e:
                if e.status != 4:  # JH_ERR_UNSUPPORTED: mixed kinds / ragged blocks -> generic path
                    raise
                self.native = None
        if self._tmp_r is None:
            self._tmp_r = zeros(_j.range_(self.A))
            self._fwd_overwrites = _blk.overwrites_its_whole_range(self.L)
        elif not self._fwd_overwrites:
            # a block operator with several columns ADDS to d as found (src/Jets.jl:1024: `_d .+= mul!(dtmp, ...)`), one with zero blocks leaves
            # their rows (1022): the reference's own `A*m` hands mul! fresh zeros (src/Jets.jl:395), and so must a reused temporary
            fill_(self._tmp_r, 0)
        _j.mul_(self._tmp_r, self.L, v)
        lincomb_(u, _plain([alpha, beta]), [self._tmp_r, u])
        return float(norm(u)) ** 2

    def fwd(self, u, v, alpha, beta) -> float:
        """u <- alpha*(A v) + beta*u ; returns ||u||."""
        return math.sqrt(self._fwd_local(u, v, alpha, beta))

    def _step_local(self, u, v, alpha, beta):
        """One pass: u <- alpha*(A v) + beta*u and w <- A'u (new u, un-normalised).  Returns (LOCAL ||u||^2, w) or None
        when the operator has no fused step (then lsqr_core runs the two halves separately)."""
        if self.native is None or not self.fused_step:
            return None
        if self._tmp_d is None:
            self._tmp_d = zeros(_j.domain(self.A))
        out = C.c_double(0)
        try:
            check(lib.jh_blockop_bidiag_step(self.native.handle, u.handle, v.handle, self._tmp_d.handle, float(alpha), float(beta), C.byref(out)))
        except JetsHipError as e:
            if e.status != 4:
                raise
            self.fused_step = False
            return None
        return out.value, self._tmp_d

    def step(self, u, v, alpha, beta):
        """u <- alpha*(A v) + beta*u ; w <- A'u.  Returns (||u||, w) or None."""
        r = self._step_local(u, v, alpha, beta)
        return None if r is None else (math.sqrt(r[0]), r[1])

    def adj(self, v, u, alpha, beta) -> float:
        """v <- alpha*(A' u) + beta*v ; returns ||v||."""
        if self.native is not None:
            out = C.c_double(0)
            try:
                check(lib.jh_blockop_mul_adj_axpby(self.native.handle, v.handle, u.handle, float(alpha), float(beta), 1.0, C.byref(out)))
                return math.sqrt(out.value)
            except JetsHipError as e:
                if e.status != 4:
                    raise
                self.native = None
        if self._tmp_d is None:
            self._tmp_d = zeros(_j.domain(self.A))
        elif not (_blk.isblockop(self.L) and self.L.jet.s["ops"].shape[0] > 1) and not _blk.overwrites_its_whole_range(_j.adjoint(self.L)):
            fill_(self._tmp_d, 0)              # (a one-row block operator's adjoint skips zero blocks, 1047; with several rows it starts from `_m .= 0`, 1042)
        _j.mul_(self._tmp_d, _j.adjoint(self.L), u)
        lincomb_(v, _plain([alpha, beta]), [self._tmp_d, v])
        return float(norm(v))


class _ShardEngine(_Engine):
    """Row-partitioned tall operator (rowpart.RowPartitionedOp): range-side vectors are this rank's rows,
    domain-side vectors are replicated.  Forward half: local fused kernel + one scalar all-reduce for ||u||^2.
    Adjoint half: local ordered sum, ONE all-reduce of the domain vector, then the small axpby + norm."""

    def __init__(self, shard):
        super().__init__(shard.local_op)
        self.shard = shard
        self.force_collective = os.environ.get("BENCH_FORCE_DIST", "0") == "1"   # run the exchange even with one rank (validation)

    def norm_rng(self, x) -> float:
        return self.shard.norm_range(x, 2)

    def fwd(self, u, v, alpha, beta) -> float:
        return math.sqrt(self.shard.comm.all_reduce_scalars([self._fwd_local(u, v, alpha, beta)], "sum")[0])

    def step(self, u, v, alpha, beta):
        if self.native is not None and self.fused_step:  # pipelined: all-reduce of a finished chunk of A'u under the next chunk's kernel
            if self._tmp_d is None:
                self._tmp_d = zeros(_j.domain(self.A))
            nrm2 = self.shard.bidiag_step_(u, v, self._tmp_d, alpha, beta, force_collective=self.force_collective)
            if nrm2 is not None:
                return math.sqrt(nrm2), self._tmp_d
        r = self._step_local(u, v, alpha, beta)          # this rank's rows: local ||u||^2 and local A'u
        if r is None:
            return None
        self.shard.comm.all_reduce_sum_(r[1])            # ONE vector all-reduce per iteration, as before
        return math.sqrt(self.shard.comm.all_reduce_scalars([r[0]], "sum")[0]), r[1]

    def adj(self, v, u, alpha, beta) -> float:
        if self._tmp_d is None:
            self._tmp_d = zeros(_j.domain(self.A))
        self.shard.mul_adj_(self._tmp_d, u)              # local A'u + all-reduce
        lincomb_(v, _plain([alpha, beta]), [self._tmp_d, v])
        return float(norm(v))


class _TeamEngine:
    """ONE process, several contexts (rowpart.Team / TeamOp): vectors are rowpart.TeamVec -- range side: every member's
    rows; domain side: the members' replicas, kept identical by the same deterministic updates on each.  Scalars need no
    collective: the host adds the members' partial sums."""

    def __init__(self, T):
        self.T = T
        self.team = T.team
        self._tmp_d = None
        self._engines = [_Engine(A) for A in T.local_ops]     # per member: the fused local halves
        self.fused_step = os.environ.get("JETS_LSQR_FUSED_STEP", "1") != "0"
        self.native = None                                    # jh_lsqr_solve* are per-rank loops: lsqr_core drives a team

    def zeros_dom(self):
        return self.team.zeros(self.T.domain())

    def zeros_rng(self):
        return self.team.zeros(self.T.ranges())

    def copy(self, dst, src):
        for k, _ in self.team.each():
            copyto_(dst[k], src[k])
        return dst

    def lincomb(self, dst, coefs, xs):
        for k, _ in self.team.each():
            lincomb_(dst[k], _plain(coefs), [x[k] for x in xs])
        return dst

    def norm_dom(self, x) -> float:
        return float(norm(x[0]))                              # replicas are identical

    def norm_rng(self, x) -> float:
        return math.sqrt(builtins.sum(float(norm(x[k])) ** 2 for k, _ in self.team.each()))

    def fwd(self, u, v, alpha, beta) -> float:
        return math.sqrt(builtins.sum(self._engines[k]._fwd_local(u[k], v[k], alpha, beta) for k, _ in self.team.each()))

    def step(self, u, v, alpha, beta):
        if not self.fused_step:
            return None
        if self._tmp_d is None:
            self._tmp_d = self.zeros_dom()
        nrm2 = self.T.bidiag_step_(u, v, self._tmp_d, alpha, beta)
        return None if nrm2 is None else (math.sqrt(nrm2), self._tmp_d)

    def adj(self, v, u, alpha, beta) -> float:
        if self._tmp_d is None:
            self._tmp_d = self.zeros_dom()
        self.T.mul_adj_(self._tmp_d, u)
        self.lincomb(v, [alpha, beta], [self._tmp_d, v])
        return self.norm_dom(v)


def lsqr(A, b, x0=None, damp: float = 0.0, atol: float = 1e-6, btol: float = 1e-6, conlim: float = 1e8, maxiter: int = 100,
         overwrite_b: bool = False, force_maxiter: bool = False) -> LsqrResult:
    """min ||A x - b||_2 (+ damp^2 ||x||^2).  `b` lives in range(A) (a BlockArray for a block operator), the
    result in domain(A).  `A` may also be a rowpart.RowPartitionedOp (then `b` is this rank's rows of the
    right-hand side and every rank returns the same x).  `overwrite_b=True` lets the solver use b's storage for
    the Lanczos vector u (at the headline size b is 64 GiB).  `force_maxiter=True` keeps iterating past every
    stopping rule (throughput measurements only)."""
    from .rowpart import RowPartitionedOp, TeamOp

    if isinstance(A, TeamOp):                            # one process, several contexts: b and the result are TeamVecs
        eng = _TeamEngine(A)
        native = _native_team_solve(eng, b, x0, damp, atol, btol, conlim, maxiter, overwrite_b, force_maxiter)
        if native is not None:
            return native
        return lsqr_core(eng, b, x0, damp, atol, btol, conlim, maxiter, overwrite_b, force_maxiter)
    if isinstance(A, RowPartitionedOp):
        eng = _ShardEngine(A)
        dom, rng = _j.domain(A.local_op), _j.range_(A.local_op)
    else:
        A = _unwrap_vec(A)
        eng = _Engine(A)
        dom, rng = _j.domain(A), _j.range_(A)
    b = reshape(b, rng)
    x0 = None if x0 is None else reshape(x0, dom)
    native = _native_solve(eng, b, x0, damp, atol, btol, conlim, maxiter, overwrite_b, force_maxiter)
    if native is not None:
        return native
    return lsqr_core(eng, b, x0, damp, atol, btol, conlim, maxiter, overwrite_b, force_maxiter)


def _native_solve(eng, b, x0, damp, atol, btol, conlim, maxiter, overwrite_b, force_maxiter):
    """The whole loop behind the C ABI (jh_lsqr_solve: the same recurrences in C++ over the one-pass step) for a
    device-native tall diagonal operator on one GPU, or row-partitioned over the ABI's own RCCL communicator (AbiComm).
    None when it does not apply (generic operators, torch.distributed exchange, JETS_LSQR_NATIVE=0): lsqr_core then runs."""
    from ._ffi import LsqrResultC
    from .rowpart import AbiComm

    if os.environ.get("JETS_LSQR_NATIVE", "1") == "0" or eng.native is None or not eng.fused_step:
        return None
    shard = getattr(eng, "shard", None)
    if shard is not None and not (isinstance(shard.comm, AbiComm) or shard.comm.world == 1):
        return None
    x = eng.zeros_dom() if x0 is None else eng.copy(eng.zeros_dom(), x0)
    u = b if overwrite_b else (eng.copy_of_rng(b) if hasattr(eng, "copy_of_rng") else eng.copy(eng.zeros_rng(), b))
    res = LsqrResultC()
    hist = (C.c_double * builtins.max(2 * int(maxiter), 1))()
    try:
        # a rank-local operator is solved locally even while an AbiComm is alive; only a RowPartitionedOp is a collective solve
        solve = lib.jh_lsqr_solve_partitioned if shard is not None else lib.jh_lsqr_solve
        check(solve(eng.native.handle, u.handle, x.handle, 0 if x0 is None else 1, float(damp), float(atol), float(btol),
                    float(conlim), int(maxiter), 1 if force_maxiter else 0, C.byref(res), hist))
    except JetsHipError as e:
        if e.status != 4:                                       # JH_ERR_UNSUPPORTED is raised before anything is touched: generic path
            raise
        return None
    history = [(k + 1, hist[2 * k], hist[2 * k + 1]) for k in builtins.range(res.itn)]
    return LsqrResult(x, res.istop, res.itn, res.r1norm, res.r2norm, res.anorm, res.acond, res.arnorm, res.xnorm, history)


def _native_team_solve(eng, b, x0, damp, atol, btol, conlim, maxiter, overwrite_b, force_maxiter):
    """jh_lsqr_solve_team: the whole loop over a single-process team behind ONE call (what the Julia binding uses).  None when a
    member has no native tall operator or JETS_LSQR_NATIVE=0: lsqr_core then drives the team from here."""
    from ._ffi import LsqrResultC
    from .rowpart import TeamVec

    T = eng.T
    if os.environ.get("JETS_LSQR_NATIVE", "1") == "0" or not eng.fused_step or any(n is None for n in T._natives):
        return None
    M = eng.team.world
    x = eng.zeros_dom() if x0 is None else eng.copy(eng.zeros_dom(), x0)
    u = b if overwrite_b else eng.copy(eng.zeros_rng(), b)
    arr = lambda hs: (C.c_void_p * M)(*[h.value if hasattr(h, "value") else h for h in hs])
    res = LsqrResultC()
    hist = (C.c_double * builtins.max(2 * int(maxiter), 1))()
    try:
        check(lib.jh_lsqr_solve_team(M, arr([n.handle for n in T._natives]), arr([u[k].handle for k in builtins.range(M)]),
                                     arr([x[k].handle for k in builtins.range(M)]), 0 if x0 is None else 1, float(damp), float(atol), float(btol),
                                     float(conlim), int(maxiter), 1 if force_maxiter else 0, C.byref(res), hist))
    except JetsHipError as e:
        if e.status != 4:
            raise
        return None
    history = [(k + 1, hist[2 * k], hist[2 * k + 1]) for k in builtins.range(res.itn)]
    return LsqrResult(x, res.istop, res.itn, res.r1norm, res.r2norm, res.anorm, res.acond, res.arnorm, res.xnorm, history)


def lsqr_core(eng, b, x0, damp, atol, btol, conlim, maxiter, overwrite_b=False, force_maxiter=False) -> LsqrResult:
    """Paige & Saunders' recurrences on an engine (see _Engine for the interface)."""
    x = eng.zeros_dom() if x0 is None else eng.copy(eng.zeros_dom(), x0)
    u = b if overwrite_b else eng.copy(eng.zeros_rng(), b)
    bnorm = eng.norm_rng(b)

    # u_hat holds beta*u (un-normalised); `beta` is its norm.  u = b - A x0
    if x0 is not None:
        beta = eng.fwd(u, x, -1.0, 1.0)
    else:
        beta = bnorm
    v = eng.zeros_dom()
    w = eng.zeros_dom()
    history = []
    itn, istop = 0, 0
    anorm = acond = ddnorm = res2 = xnorm = xxnorm = z = 0.0
    cs2, sn2 = -1.0, 0.0
    if beta > 0:
        alpha = eng.adj(v, u, 1.0 / beta, 0.0)          # v = A'u with u = u_hat / beta
    else:
        eng.copy(v, x)
        alpha = 0.0
    if alpha > 0:
        eng.lincomb(v, [1.0 / alpha], [v])
    eng.copy(w, v)
    rhobar, phibar = alpha, beta
    rnorm = r1norm = r2norm = beta
    arnorm = alpha * beta
    if arnorm == 0:
        return LsqrResult(x, 0, 0, r1norm, r2norm, anorm, acond, arnorm, xnorm, history)
    eps = 2.220446049250313e-16
    ctol = 1.0 / conlim if conlim > 0 else 0.0

    step = getattr(eng, "step", None)
    while itn < maxiter:
        itn += 1
        # --- bidiagonalisation:  beta*u = A v - alpha*u ;  alpha*v = A'u - beta*v
        beta_prev = beta
        fused = step(u, v, 1.0, -alpha / beta_prev) if step is not None else None
        if fused is not None:                            # one pass over A and u: u_hat and A'u_hat together
            beta, atu = fused
            if beta > 0:
                anorm = math.sqrt(anorm ** 2 + alpha ** 2 + beta ** 2 + damp ** 2)
                eng.lincomb(v, [1.0 / beta, -beta], [atu, v])      # v <- A'(u_hat) / beta - beta v   (A' is linear)
                alpha = eng.norm_dom(v)
                if alpha > 0:
                    eng.lincomb(v, [1.0 / alpha], [v])
        else:
            beta = eng.fwd(u, v, 1.0, -alpha / beta_prev)    # u_hat <- A v - alpha * (u_hat / beta_prev)
            if beta > 0:
                anorm = math.sqrt(anorm ** 2 + alpha ** 2 + beta ** 2 + damp ** 2)
                alpha = eng.adj(v, u, 1.0 / beta, -beta)     # v <- A'(u_hat / beta) - beta v
                if alpha > 0:
                    eng.lincomb(v, [1.0 / alpha], [v])
        # --- eliminate the damping parameter
        rhobar1 = math.sqrt(rhobar ** 2 + damp ** 2)
        rho = math.sqrt(rhobar1 ** 2 + beta ** 2)
        if not (rhobar1 > 0 and math.isfinite(rho)):     # only reachable with force_maxiter, far past convergence: the recurrences have
            istop = istop or 6                           # underflowed; x stays at its last finite update
            itn -= 1
            break
        cs1, sn1 = rhobar / rhobar1, damp / rhobar1
        psi = sn1 * phibar
        phibar = cs1 * phibar
        # --- plane rotation to eliminate the subdiagonal of the bidiagonal matrix
        cs, sn = rhobar1 / rho, beta / rho
        theta = sn * alpha
        rhobar = -cs * alpha
        phi = cs * phibar
        phibar = sn * phibar
        tau = sn * phi
        # --- update x and w (domain-sized vectors)
        t1, t2 = phi / rho, -theta / rho
        ddnorm += (eng.norm_dom(w) / rho) ** 2           # ||w / rho||^2 without materialising w / rho
        eng.lincomb(x, [1.0, t1], [x, w])
        eng.lincomb(w, [1.0, t2], [v, w])
        # --- norms for the stopping rules
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
        r2norm = rnorm
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
        if istop and not (force_maxiter and itn < maxiter and alpha > 0 and beta > 0):
            break
    return LsqrResult(x, istop, itn, r1norm, r2norm, anorm, acond, arnorm, xnorm, history)
