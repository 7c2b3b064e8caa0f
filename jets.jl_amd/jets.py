"""Jet / Jop core, composition, sums, scalar multiples, vectorised operators and test utilities.

Host-side mirror of /root/reference/src/Jets.jl:131-403 (Jet, JopNl/JopLn/JopAdjoint, accessors,
mul!), 518-623 (JetComposite), 625-731 (JetSum), 1126-1164 (JetVec, scalar * operator) and
1187-1226 (dot_product_test), for operators whose vectors live in HBM.  Only dispatch lives here:
every array operation is a HIP kernel behind the C ABI.

Python spelling of the Julia API (INTEGRATION.md has the full table):
    mul!(d, A, m)  -> mul_(d, A, m)        A*m     -> A * m            A'      -> A.H  /  adjoint(A)
    A2 o A1        -> A2 @ A1 / compose    A1 + A2 -> A1 + A2          a*A     -> a * A
    f!, df!, df'!  -> f=, df=, df_adj=     upstate! -> upstate=        m_o kw  -> mo=
Closures have the reference's signatures: f(d, m, **state), df(d, m, mo=..., **state),
df_adj(m, d, mo=..., **state); each mutates and RETURNS its first argument (src/Jets.jl:190-192).
Operator kinds are recognised by the identity of the `f` function, the way the reference dispatches
on typeof(f!) (src/Jets.jl:543, 949, 1085, 1097).
"""
from __future__ import annotations

import builtins
import copy as _copy
from typing import Any, Callable, Sequence

import numpy as np

from . import arrays as _arr
from . import device as _dev
from .arrays import DeviceArray, BlockArray, zeros, ones, lincomb_, hadamard_, copyto_, fill_, reshape, dot
from .spaces import JetAbstractSpace

__all__ = [
    "Jet", "Jop", "JopNl", "JopLn", "JopAdjoint", "jet_missing", "f_", "df_", "df_adj_", "domain", "range_", "eltype",
    "state", "state_", "perfstat", "point", "point_", "close", "jet", "shape", "size", "jacobian_", "jacobian", "adjoint",
    "mul_", "mul", "JetComposite", "JetComposite_f", "JetComposite_df", "JetComposite_df_adj", "compose", "JetSum",
    "JetSum_f", "JetSum_df", "JetSum_df_adj", "JetVec", "JopVec", "JetVec_f", "JetVec_df", "JetVec_df_adj", "vec_op",
    "scale_op", "constdiag_df", "constdiag_df_adj", "dot_product_test", "linearization_test", "linearity_test", "convert_op", "copy_op",
    "PLUS", "MINUS",
]


def jet_missing(*_a, **_k):  # src/Jets.jl:131
    raise NotImplementedError("not implemented")


def _default_upstate(m, s):  # src/Jets.jl:176
    return None


# bumped whenever any jet's linearisation point changes: block jets cache "all my children are pointed where I am" against it
POINT_GEN = [0]
# bumped by state!: argument tables a block jet built from its children's states are stale afterwards
STATE_GEN = [0]


class Jet:
    """Jet(; dom, rng, f!, df!, df'!, upstate!, s)  (src/Jets.jl:133-188)."""

    def __init__(self, *, dom: JetAbstractSpace, rng: JetAbstractSpace, f: Callable = jet_missing, df: Callable = jet_missing,
                 df_adj: Callable = jet_missing, upstate: Callable = _default_upstate, s: dict | None = None, mo: Any = None):
        if f is jet_missing and df is jet_missing:  # :178-180
            raise ValueError("must set at-least one of f! and df!")
        if f is jet_missing:  # :181-183
            f = df
        if df_adj is jet_missing:  # :184-186
            df_adj = df
        self.dom, self.rng = dom, rng
        self.f, self.df, self.df_adj, self.upstate = f, df, df_adj, upstate
        self.mo = mo  # the reference holds a zero-size array here until point! is called (:187)
        self.s = dict(s or {})


def f_(d, jet_: Jet, m, **kw):  # :190
    return jet_.f(d, m, **kw)


def df_(d, jet_: Jet, m, **kw):  # :191
    return jet_.df(d, m, **kw)


def df_adj_(m, jet_: Jet, d, **kw):  # :192
    return jet_.df_adj(m, d, **kw)


def _as_op(x):
    """A Jop as it is; a MATRIX operand of a composition or a sum (src/Jets.jl:572-576, 691-692, 707-708: `A3 o A2` with a plain
    Julia matrix A3) wrapped as the dense operator d .= A*m / m .= A'*d: a 2-D device array, a 2-D numpy array (uploaded),
    or a scaled device matrix like `3.0*A3`.  Anything else: None."""
    if isinstance(x, Jop):
        return x
    from .arrays import LinExpr, from_numpy

    if isinstance(x, LinExpr) and all(isinstance(t, DeviceArray) and len(t.shape) == 2 for _, t in x.terms):
        x = x.materialize()
    if isinstance(x, np.ndarray) and x.ndim == 2:
        x = from_numpy(np.asfortranarray(x))
    if isinstance(x, DeviceArray) and len(x.shape) == 2:
        from .jetblock import JopDense

        return JopDense(x)
    return None


class Jop:
    """abstract Jop (src/Jets.jl:194)."""

    __array_ufunc__ = None  # numpy_matrix @ A, numpy_matrix + A reach __rmatmul__ / __radd__

    # A * m  (:399) ;  a * A (:1161-1164)
    def __mul__(self, m):
        if isinstance(m, (DeviceArray, BlockArray)):
            return mul(self, m)
        return NotImplemented

    def __rmul__(self, a):
        if isinstance(a, (int, float, complex, np.number)):
            return scale_op(a, self)
        return NotImplemented

    def __matmul__(self, other):  # A2 o A1   (A1 may be a matrix, :575)
        other = _as_op(other)
        return compose(self, other) if other is not None else NotImplemented

    def __rmatmul__(self, other):  # matrix o A1   (:576)
        other = _as_op(other)
        return compose(other, self) if other is not None else NotImplemented

    def __add__(self, other):  # (:689-692)
        other = _as_op(other)
        return _sum(self, other, PLUS) if other is not None else NotImplemented

    def __radd__(self, other):
        other = _as_op(other)
        return _sum(other, self, PLUS) if other is not None else NotImplemented

    def __sub__(self, other):  # (:705-708)
        other = _as_op(other)
        return _sum(self, other, MINUS) if other is not None else NotImplemented

    def __rsub__(self, other):
        other = _as_op(other)
        return _sum(other, self, MINUS) if other is not None else NotImplemented

    @property
    def H(self):
        return adjoint(self)

    T = H


class JopNl(Jop):  # :196-207
    def __init__(self, jet_: Jet | None = None, **kw):
        self.jet = jet_ if jet_ is not None else Jet(**kw)

    def __repr__(self):  # :403
        return f"Jet nonlinear operator, {domain(self).size()} -> {range_(self).size()}"


class JopLn(Jop):  # :209-224
    def __new__(cls, jet_=None, mo=None, **kw):
        if isinstance(jet_, JopLn):  # JopLn(A::JopLn) = A  (:223)
            return jet_
        if isinstance(jet_, JopAdjoint):  # JopLn(A::JopAdjoint) = A  (:235)
            return jet_
        return super().__new__(cls)

    def __init__(self, jet_=None, mo=None, **kw):
        if isinstance(jet_, (JopLn, JopAdjoint)):
            return
        if isinstance(jet_, JopNl):  # JopLn(F::JopNl) = JopLn(jet(F))  (:224)
            jet_ = jet_.jet
        if jet_ is None:
            jet_ = Jet(**kw)  # :221
        if mo is not None:  # JopLn(jet, mo) = JopLn(point!(jet, mo))  (:212)
            point_(jet_, mo)
        self.jet = jet_

    def __repr__(self):  # :401
        return f"Jet linear operator, {domain(self).size()} -> {range_(self).size()}"


class JopAdjoint(Jop):  # :226-228
    def __init__(self, op: Jop):
        self.op = op

    def __repr__(self):  # :402
        return f"Jet adjoint operator, {domain(self).size()} -> {range_(self).size()}"


def copy_op(A, copymo: bool = True):
    """copy(jet/op) (src/Jets.jl:230-233): same closures, deep-copied state, optionally copied mo."""
    if isinstance(A, Jet):
        mo = A.mo
        if copymo and mo is not None:
            mo = copyto_(_arr.similar(mo), mo)
        j = Jet(dom=A.dom, rng=A.rng, f=A.f, df=A.df, df_adj=A.df_adj, upstate=A.upstate, mo=mo)
        j.s = {k: (copy_op(v, copymo) if isinstance(v, (Jet, Jop)) else _deepcopy_state(v, copymo)) for k, v in A.s.items()}
        return j
    if isinstance(A, JopLn):
        return JopLn(copy_op(A.jet, copymo))
    if isinstance(A, JopAdjoint):
        return JopAdjoint(copy_op(A.op, copymo))
    if isinstance(A, JopNl):
        return JopNl(copy_op(A.jet, copymo))
    raise TypeError(type(A))


def _deepcopy_state(v, copymo):
    if isinstance(v, (DeviceArray, BlockArray)):
        return copyto_(_arr.similar(v), v)
    if isinstance(v, (list, tuple)):
        return type(v)(_deepcopy_state(x, copymo) if not isinstance(x, (Jet, Jop)) else copy_op(x, copymo) for x in v)
    if isinstance(v, np.ndarray) and v.dtype == object:
        out = np.empty_like(v)
        for idx in np.ndindex(v.shape):
            x = v[idx]
            out[idx] = copy_op(x, copymo) if isinstance(x, (Jet, Jop)) else _deepcopy_state(x, copymo)
        return out
    return _copy.deepcopy(v)


# ------------------------------------------------------------------------------ accessors ----------
def jet(A) -> Jet:  # :308-309
    if isinstance(A, JopAdjoint):
        return jet(A.op)
    return A.jet


def domain(A):  # :242, 319, 322
    if isinstance(A, Jet):
        return A.dom
    if isinstance(A, JopAdjoint):
        return range_(A.op)
    return A.jet.dom


def range_(A):  # :249, 320, 323
    if isinstance(A, Jet):
        return A.rng
    if isinstance(A, JopAdjoint):
        return domain(A.op)
    return A.jet.rng


def eltype(A):  # :256, 310
    if isinstance(A, (JetAbstractSpace, DeviceArray, BlockArray)):
        return A.eltype()
    j = A if isinstance(A, Jet) else jet(A)
    return np.promote_types(j.dom.eltype(), j.rng.eltype())


def state(A, key=None):  # :264-265, 313-314, 607-623
    j = A if isinstance(A, Jet) else jet(A)
    if key is None:
        return j.s
    if j.f is JetComposite_f and key not in j.s:  # :607-623
        found = [op for op in j.s["ops"] if key in state(op)]
        if not found:
            raise KeyError(f"key {key} does not exist in the state of the composite operator")
        if len(found) > 1:
            raise KeyError(f"ambiguous: key {key} exists in more than one operator in the composition")
        return state(found[0], key)
    return j.s[key]


def state_(A, s: dict):  # :272, 315  merge
    j = A if isinstance(A, Jet) else jet(A)
    j.s = {**j.s, **s}
    STATE_GEN[0] += 1                                       # cached argument tables built from a state (jetblock) are stale now
    return A


_perfstat_registry: dict = {}


def perfstat(A):  # :281, 316, 597-605, 723-731
    j = A if isinstance(A, Jet) else jet(A)
    if j.f is JetComposite_f or j.f is JetSum_f:
        s = None
        for op in j.s["ops"]:
            s = perfstat(op)
            if s is not None:
                break
        return s
    fn = _perfstat_registry.get(j.f)
    return fn(j) if fn else None


def register_perfstat(f: Callable, fn: Callable) -> None:
    """Operator authors override perfstat for their jet kind (test/runtests.jl:9)."""
    _perfstat_registry[f] = fn


def point(A):  # :288, 311-312
    j = A if isinstance(A, Jet) else jet(A)
    return j.mo


_close_registry: dict = {}


def register_close(f: Callable, fn: Callable) -> None:
    """Operator authors release resources on close (test/runtests.jl:18)."""
    _close_registry[f] = fn


def close(A):  # :290, 317, 591-595, 717-721, 1120-1124
    j = A if isinstance(A, Jet) else jet(A)
    from . import jetblock as _blk  # late import: blockop imports this module

    if j.f is JetComposite_f or j.f is JetSum_f:
        for op in j.s["ops"]:
            close(op)
        if isinstance(j.s.get("_ws"), _Workspace):
            j.s["_ws"].close()
        if hasattr(j.s.get("_chains"), "close"):
            j.s["_chains"].close()
        return None
    if j.f is _blk.JetBlock_f:
        _blk.close_block(j)
        return None
    if j.f is JetVec_f:
        return close(j.s["op"])
    fn = _close_registry.get(j.f)
    if fn:
        return fn(j)
    return False


def point_(A, mo):  # :297-301, 578-589, 710-715, 1059-1066
    j = A if isinstance(A, Jet) else jet(A)
    from . import jetblock as _blk

    if j.f is JetComposite_f:  # :578-589
        j.mo = mo
        POINT_GEN[0] += 1
        ops = j.s["ops"]
        _m = copyto_(_arr.similar(mo), mo)
        for i in builtins.range(len(ops) - 1, -1, -1):
            point_(jet(ops[i]), _m)
            if i > 0:
                _m = mul_(zeros(range_(ops[i])), ops[i], _m)
        return A
    if j.f is JetSum_f:  # :710-715
        for op in j.s["ops"]:
            point_(jet(op), mo)
        return A
    if j.f is _blk.JetBlock_f:  # :1059-1066
        _blk.point_block(j, mo)
        return A
    j.mo = mo
    POINT_GEN[0] += 1
    j.upstate(mo, j.s)
    return A


def shape(A, i=None):  # :328-345
    if i is None:
        return (range_(A).size(), domain(A).size())
    return range_(A).size() if i == 1 else domain(A).size()


def size(A, i=None):  # :354-355  (i = 1: range, i = 2: domain, as in the reference)
    if i is None:
        return (range_(A).length(), domain(A).length())
    return range_(A).length() if i == 1 else domain(A).length()


def jacobian_(F, mo):  # :364-366
    if isinstance(F, Jet):
        return JopLn(F, mo)
    if isinstance(F, JopNl):
        return jacobian_(F.jet, mo)
    return F


def jacobian(F, mo):  # :374
    return jacobian_(copy_op(F, False), copyto_(_arr.similar(mo), mo))


def adjoint(A):  # :382-383
    if isinstance(A, JopAdjoint):
        return A.op
    if isinstance(A, JopLn):
        return JopAdjoint(A)
    raise TypeError("adjoint is defined for linear operators (JopLn / JopAdjoint)")


# ------------------------------------------------------------------------------ mul! ---------------
def _enter_context_of(x):
    """Several contexts in this process: temporaries that the combinators allocate while applying an operator must land in
    the operand's context, so it becomes the current one (the C ABI switches per call anyway; this is for the factories)."""
    if _dev.several_contexts() and hasattr(x, "handle"):
        _dev.context_use(_dev.context_of(x))


def mul_(d, A: Jop, m):
    """mul!(d, A, m) (src/Jets.jl:390-392)."""
    _enter_context_of(m)
    if isinstance(A, JopNl):
        return f_(d, A.jet, m, **A.jet.s)
    if isinstance(A, JopLn):
        return df_(d, A.jet, m, mo=A.jet.mo, **A.jet.s)
    if isinstance(A, JopAdjoint):
        if not isinstance(A.op, JopLn):
            raise TypeError("mul! with an adjoint needs a linear operator")
        j = A.op.jet
        return df_adj_(d, j, m, mo=j.mo, **j.s)
    raise TypeError(type(A))


def mul(A: Jop, m):
    """A*m = mul!(zeros(range(A)), A, m) (src/Jets.jl:399).  The zeros are there because a df! may accumulate into its output (the block
    loop with several columns does, 1024; a zero block leaves its row as found, 1022); a one-column block operator of device-native
    children without zero blocks overwrites every row (1026), so its output is allocated without the fill (12 ms per 64 GiB)."""
    _enter_context_of(m)
    from .jetblock import overwrites_its_whole_range

    R = range_(A)                                      # (an operator's output: of the cached slabs of its size the one that is fastest to write)
    return mul_(_arr.Array(R, undef=overwrites_its_whole_range(A), role=_arr.ROLE_OUTPUT), A, m)


# ------------------------------------------------------------------------------ composition --------
class _Workspace:
    """Temporaries of a combinator.  The reference allocates `zeros(range(op))` for every stage on every call
    (src/Jets.jl:525, 531, 537, 632, 641, 650).  Here: a temporary of 16 MiB or more is taken for ONE call and given back when the
    call returns -- the library's slab cache (include/jetship.h, jh_trim) makes that a pointer hand-over, and no combinator object
    sits on 64 GiB between calls; smaller ones are kept and re-zeroed (an allocation per call would cost more than the kernels).
    The zero fill is skipped where the stage overwrites its whole output (jetblock.overwrites_its_whole_range): block operators
    skip zero blocks and accumulate into their output otherwise (src/Jets.jl:1022-1024)."""

    KEEP_BELOW = 16 << 20

    def __init__(self):
        self._pool = {}
        self._call = []

    def __deepcopy__(self, memo):  # copy(jet) gets its own, empty pool
        return _Workspace()

    def zeros(self, slot, R, for_op=None, overwritten=False):
        from .jetblock import overwrites_its_whole_range

        undef = overwritten or (for_op is not None and overwrites_its_whole_range(for_op))
        if R.length() * np.dtype(R.eltype()).itemsize >= self.KEEP_BELOW:
            x = _arr.Array(R, undef=undef, role=_arr.ROLE_OUTPUT)
            self._call.append(x)
            return x
        hit = self._pool.get(slot)
        if hit is None or hit[0] != R:
            hit = (R, zeros(R))
            self._pool[slot] = hit
            return hit[1]
        return hit[1] if undef else fill_(hit[1], 0)

    def release(self):
        """End of a call: the big temporaries go back (to the slab cache)."""
        for x in self._call:
            x.close()
        self._call = []

    def close(self):
        self.release()
        for _, x in self._pool.values():
            x.close()
        self._pool = {}


def _zeroed_output(out, op):
    """The output of a combinator's last stage: zeroed like the reference's zeros() unless the stage overwrites all of it."""
    from .jetblock import overwrites_its_whole_range

    return out if overwrites_its_whole_range(op) else fill_(out, 0)


def JetComposite(ops: Sequence[Jop]) -> Jet:  # :522
    from .chains import ChainCache

    ops = tuple(ops)
    return Jet(f=JetComposite_f, df=JetComposite_df, df_adj=JetComposite_df_adj, dom=domain(ops[-1]), rng=range_(ops[0]),
               s={"ops": ops, "_ws": _Workspace(), "_chains": ChainCache()})


def _chain(out, x, stages, ws):
    """x -> stages[0] -> ... -> stages[-1] -> out.  Every stage but the last writes a zero-filled temporary
    (src/Jets.jl:525/531/537); the last writes `out` itself after zeroing it, which is the reference's
    `d .= chain(m)` (526/532/538) without the extra copy."""
    ws = ws if ws is not None else _Workspace()
    try:
        for k, (op, R) in enumerate(stages[:-1]):
            x = mul_(ws.zeros(k, R, op), op, x)
        op, _ = stages[-1]
        return mul_(_zeroed_output(out, op), op, x)
    finally:
        ws.release()


def JetComposite_f(d, m, *, ops, _ws=None, _chains=None, **kw):  # :524-528  right-to-left chain
    from . import chains as _chn

    stages = [(op, range_(op)) for op in reversed(ops)]
    if len(ops) >= 2 and _chn.run(d, m, stages, _ws, _chains, "f") is not None:      # runs of elementwise stages (F o A o F o A): one fused pass (chains.py)
        return d
    return _chain(d, m, stages, _ws)


def JetComposite_df(d, m, *, ops, _ws=None, _chains=None, **kw):  # :530-534
    from . import jetblock as _blk
    from . import chains as _chn

    fused = _blk.try_fused_chain(d, m, ops)
    if fused is not None:
        return fused
    stages = [(JopLn(op), range_(JopLn(op))) for op in reversed(ops)]
    if len(ops) >= 2 and _chn.run(d, m, stages, _ws, _chains, "df") is not None:      # chains of any depth: every fusable run in one pass (chains.py)
        return d
    return _chain(d, m, stages, _ws)


def JetComposite_df_adj(m, d, *, ops, _ws=None, _chains=None, **kw):  # :536-540
    from . import jetblock as _blk
    from . import chains as _chn

    fused = _blk.try_fused_chain(m, d, tuple(adjoint(JopLn(op)) for op in reversed(ops)))
    if fused is not None:
        return fused
    stages = [(adjoint(JopLn(op)), domain(JopLn(op))) for op in ops]
    if len(ops) >= 2 and _chn.run(m, d, stages, _ws, _chains, "df_adj") is not None:
        return m
    return _chain(m, d, stages, _ws)


def jops_comp(op: Jop) -> tuple:  # :542-550
    if isinstance(op, (JopLn, JopNl)) and op.jet.f is JetComposite_f:
        return tuple(op.jet.s["ops"])
    if isinstance(op, JopAdjoint) and isinstance(op.op, JopLn) and op.op.jet.f is JetComposite_f:
        ops = op.op.jet.s["ops"]
        return tuple(JopAdjoint(o) if not isinstance(o, JopAdjoint) else o.op for o in reversed(ops))
    return (op,)


def compose(A2: Jop, A1: Jop) -> Jop:  # :569-570
    ops = jops_comp(A2) + jops_comp(A1)
    if isinstance(A2, (JopLn, JopAdjoint)) and isinstance(A1, (JopLn, JopAdjoint)):
        return JopLn(JetComposite(ops))
    return JopNl(JetComposite(ops))


# ------------------------------------------------------------------------------ sums ---------------
PLUS, MINUS = "+", "-"


def JetSum(ops: Sequence[Jop], sgns: Sequence[str]) -> Jet:  # :628
    from .chains import ChainCache

    ops = tuple(ops)
    return Jet(f=JetSum_f, df=JetSum_df, df_adj=JetSum_df_adj, dom=domain(ops[0]), rng=range_(ops[0]),
               s={"ops": ops, "sgns": tuple(sgns), "_ws": _Workspace(), "_chains": ChainCache()})


def _accumulate(sgn: str, acc, term):
    """broadcast!(sgn, acc, acc, term) (src/Jets.jl:634)."""
    return lincomb_(acc, [1.0, 1.0 if sgn == PLUS else -1.0], [acc, term])


def JetSum_f(d, m, *, ops, sgns, _ws=None, **kw):  # :630-637
    fill_(d, 0)
    ws = _ws or _Workspace()
    try:
        _d = ws.zeros("rng", range_(ops[0]))   # one temporary, zeroed once per call like the reference (:632)
        for op, sg in zip(ops, sgns):
            _accumulate(sg, d, mul_(_d, op, m))
    finally:
        ws.release()
    return d


def JetSum_df(d, m, *, ops, sgns, _ws=None, _chains=None, **kw):  # :639-646
    from . import jetblock as _blk
    from . import chains as _chn

    if _blk.try_fused_sum(d, m, ops, sgns, False) is not None:
        return d
    if _chn.try_sum(d, m, ops, sgns, False, _ws, _chains) is not None:                 # terms that are chains: each adds itself in its last stage
        return d
    fill_(d, 0)
    ws = _ws or _Workspace()
    try:
        _d = ws.zeros("rng", range_(ops[0]))
        for op, sg in zip(ops, sgns):
            _accumulate(sg, d, mul_(_d, JopLn(op), m))
    finally:
        ws.release()
    return d


def JetSum_df_adj(m, d, *, ops, sgns, _ws=None, _chains=None, **kw):  # :648-655
    from . import jetblock as _blk
    from . import chains as _chn

    if _blk.try_fused_sum(m, d, ops, sgns, True) is not None:
        return m
    if _chn.try_sum(m, d, ops, sgns, True, _ws, _chains) is not None:
        return m
    fill_(m, 0)
    ws = _ws or _Workspace()
    try:
        _m = ws.zeros("dom", domain(ops[0]))
        for op, sg in zip(ops, sgns):
            _accumulate(sg, m, mul_(_m, adjoint(JopLn(op)), d))
    finally:
        ws.release()
    return m


def jops_sum(op: Jop) -> tuple:  # :657-665
    if isinstance(op, (JopLn, JopNl)) and op.jet.f is JetSum_f:
        return tuple(op.jet.s["ops"])
    if isinstance(op, JopAdjoint) and isinstance(op.op, JopLn) and op.op.jet.f is JetSum_f:
        return tuple(JopAdjoint(o) for o in op.op.jet.s["ops"])
    return (op,)


def flipsgn(sgn: str, sgnnew: str) -> str:  # :667-671
    if sgnnew == PLUS:
        return sgn
    return PLUS if sgn == MINUS else MINUS


def sgns(op: Jop, r: str) -> tuple:  # :673-676
    inner = op.op if isinstance(op, JopAdjoint) else op
    if isinstance(inner, (JopLn, JopNl)) and inner.jet.f is JetSum_f:
        return tuple(flipsgn(s, r) for s in inner.jet.s["sgns"])
    return (r,)


def _sum(A2, A1, sign: str):  # :689-690, 705-706
    if not isinstance(A1, Jop) or not isinstance(A2, Jop):
        return NotImplemented
    ops = jops_sum(A2) + jops_sum(A1)
    sg = sgns(A2, PLUS) + sgns(A1, sign)
    if isinstance(A2, (JopLn, JopAdjoint)) and isinstance(A1, (JopLn, JopAdjoint)):
        return JopLn(JetSum(ops, sg))
    return JopNl(JetSum(ops, sg))


# ------------------------------------------------------------------------------ scalar * operator --
def constdiag_df(d, m, *, a, **kw):  # :1159   d .= a * m
    return lincomb_(d, [a], [m])


def constdiag_df_adj(m, d, *, a, **kw):  # :1160   m .= conj(a) * d
    return lincomb_(m, [a.conjugate() if isinstance(a, (complex, np.complexfloating)) else a], [d])   # (conj keeps a's TYPE: np.conj would widen a Python complex)


def scale_op(a, A: Jop) -> Jop:  # :1161-1164
    """a*A.  Documented fix: the reference builds the scalar operator on domain(A) for BOTH spaces (:1162), which only
    works for square A (its test uses a 10x10 matrix, test/runtests.jl:789-795); here it lives on range(A), which is
    identical for square operators and makes a*A valid for tall block operators too."""
    _a = JopLn(dom=range_(A), rng=range_(A), df=constdiag_df, df_adj=constdiag_df_adj, s={"a": a})
    return compose(_a, A)


# ------------------------------------------------------------------------------ vectorised op ------
def JetVec(op: Jop):  # :1129-1130
    if range_(op).ndims() == 1 and domain(op).ndims() == 1:
        return op
    return Jet(f=JetVec_f, df=JetVec_df, df_adj=JetVec_df_adj, dom=domain(op).vec(), rng=range_(op).vec(), s={"op": op})


def JopVec(op: Jop):  # :1131-1132
    j = JetVec(op)
    if isinstance(j, Jop):
        return j
    return JopLn(j) if isinstance(op, (JopLn, JopAdjoint)) else JopNl(j)


vec_op = JopVec  # vec(A)  (:1154)


def JetVec_f(d, m, *, op, **kw):  # :1134
    mul_(reshape(d, range_(op)), op, reshape(m, domain(op)))
    return d


def JetVec_df(d, m, *, op, **kw):  # :1135
    mul_(reshape(d, range_(op)), JopLn(op), reshape(m, domain(op)))
    return d


def JetVec_df_adj(m, d, *, op, **kw):  # :1136
    mul_(reshape(m, domain(op)), adjoint(JopLn(op)), reshape(d, range_(op)))
    return m


# ------------------------------------------------------------------------------ utilities ----------
def dot_product_test(op: JopLn, m, d, mmask=None, dmask=None):
    """lhs, rhs = dot_product_test(A, m, d; mmask, dmask)  (src/Jets.jl:1211-1226)."""
    def masked(mask, x):  # mask .* x ; a REAL mask on a complex vector is a mixed-eltype broadcast (real (x) complex, Julia's rule)
        if mask is None:
            # the default mask is ones(space) (:1212-1213) and 1 .* x has x's bits: neither the ones nor the product is materialised --
            # at the headline size each would be another 64 GiB next to A, d and A*m
            return x
        if np.dtype(mask.dtype) != np.dtype(x.dtype):
            from .broadcast import broadcast_

            return broadcast_(_arr.similar(x), "x0*x1", [mask, x])
        return hadamard_(_arr.similar(x), mask, x)

    mm = masked(mmask, m)  # mmask .* m
    dd = masked(dmask, d)  # dmask .* d
    ms = mul(adjoint(op), dd)  # :1216
    lhs = dot(mm, ms)  # :1218
    ds = mul(op, mm)  # :1215
    rhs = dot(ds, dd)  # :1219
    for tmp, src in ((ds, None), (dd, d), (mm, m), (ms, None)):          # range-sized temporaries go back (to the slab cache) now, not at the next GC
        if tmp is not src and hasattr(tmp, "close"):
            tmp.close()
    if np.iscomplexobj(lhs) and np.iscomplexobj(rhs):  # :1221-1225
        return lhs, rhs
    return np.real(lhs), np.real(rhs)


def linearization_test(F: JopNl, mo, mu=(1.0, 0.5, 0.25, 0.125, 0.0625, 0.03125), dm=None, mmask=None, dmask=None, seed=None):
    """mu_obs, mu_exp = linearization_test(F, mo; mu, dm, mmask, dmask, seed)  (src/Jets.jl:1228-1269): the Jacobian of F
    satisfies F(mo + mu dm) = F(mo) + mu J dm + O(mu^2), so halving mu divides the misfit by 4.  All vector work on the
    device (F!, point!, J, fused broadcasts, norms)."""
    from .arrays import rand, lincomb_, norm

    from .broadcast import broadcast_

    # the default masks are ones(space) (:1238-1239) and 1 .* x has x's bits: they are not materialised (a range-sized one is 64 GiB at the
    # headline size); likewise d_lin and the per-mu F*(...) live in ONE range vector -- the loop below holds Fo, Jo*dm and that one
    if dm is None:  # :1242-1243   dm = mmask .* (-1 .+ 2 .* rand(domain(F)))
        r = rand(domain(F)) if seed is None else rand(domain(F), seed=int(seed), stream=0)
        lincomb_(r, [2.0], [r])
        r2 = ones(domain(F))
        lincomb_(r, [1.0, -1.0], [r, r2])
        dm = r if mmask is None else hadamard_(r, mmask, r)
    elif mmask is not None:
        dm = hadamard_(dm, mmask, dm)  # :1245   dm .*= mmask
    Fo = mul(F, mo)  # :1248
    Jo = jacobian_(F, mo)  # :1249
    Jodm = mul(Jo, dm)  # :1250
    mus = sorted((float(x) for x in mu), reverse=True)  # :1252
    phi = []
    d_non, m_mu = _arr.similar(Fo), _arr.similar(mo)
    for x in mus:
        lincomb_(m_mu, [1.0, x], [mo, dm])
        mul_(fill_(d_non, 0), F, m_mu)  # :1259   F*(mo .+ mu .* dm) = mul!(zeros(range(F)), F, .) (re-points nothing: F's jet keeps mo until the next point!)
        broadcast_(d_non, "x0 - (x1 + s0*x2)", [d_non, Fo, Jodm], [x])  # :1258, 1260   d_non .- (Fo .+ mu .* Jo dm), every operation rounded as written
        if dmask is not None:
            hadamard_(d_non, dmask, d_non)
        phi.append(float(norm(d_non)))  # :1260
    for tmp in (Fo, Jodm, d_non, m_mu):
        tmp.close()
    mu_obs = [phi[i - 1] / phi[i] for i in builtins.range(1, len(mus))]  # :1262
    mu_exp = [(mus[i - 1] / mus[i]) ** 2 for i in builtins.range(1, len(mus))]  # :1263
    return np.array(mu_obs), np.array(mu_exp)


def linearity_test(A, m1=None, m2=None):
    """lhs, rhs = linearity_test(A)  (src/Jets.jl:1271-1283): A(m1 + m2) against A m1 + A m2."""
    from .arrays import rand, lincomb_

    def draw():  # :1279   -1 .* 2 .* rand(domain(A))
        r = rand(domain(A))
        return lincomb_(r, [-2.0], [r])

    m1 = draw() if m1 is None else m1
    m2 = draw() if m2 is None else m2
    s12 = lincomb_(_arr.similar(m1), [1.0, 1.0], [m1, m2])
    lhs = mul(A, s12)  # :1281
    a1, a2 = mul(A, m1), mul(A, m2)
    rhs = lincomb_(a1, [1.0, 1.0], [a1, a2])  # :1282
    return lhs, rhs


def convert_op(A) -> np.ndarray:
    """convert(Array, A::Jop)  (src/Jets.jl:1172-1183): the operator's matrix, one column per unit vector of the domain
    (size(A,2) products on the device: a debugging aid for small operators, like the reference's)."""
    from .arrays import fill_

    nr, nc = size(A)
    m, d = zeros(domain(A)), zeros(range_(A))
    B = np.zeros((nr, nc), dtype=eltype(A))
    for icol in builtins.range(nc):
        fill_(m, 0)  # :1177
        fill_(d, 0)  # :1178
        if hasattr(m, "indices"):
            m[icol] = 1  # :1179  (BlockArray linear indexing, src/Jets.jl:825-827)
        else:
            m._upload(np.ones(1, dtype=m.dtype), icol)
        B[:, icol] = mul_(d, A, m).to_numpy().ravel(order="F")  # :1180
    return B
