"""Block operators: JetBlock / JopBlock / @blockop / JopZeroBlock and the block loops.

Mirrors /root/reference/src/Jets.jl:926-1124.  `JetBlock_df` / `JetBlock_df_adj` issue ONE call into
libjetship.so (jh_blockop_mul / jh_blockop_mul_adj) when every block is a device-native kind
(zero / identity / scaled identity / diagonal, recognised by the identity of its df function the
way the reference recognises JopZeroBlock by typeof(df!), src/Jets.jl:949); any other block
falls back to the reference's per-block loop over child mul! calls on slab views.
"""
from __future__ import annotations

import builtins
import ctypes as C
import itertools
from typing import Sequence

import numpy as np

from ._ffi import lib, check, BlockDesc, KINDS, JetsHipError, scalar_flags, SCALAR_WIDE
from . import arrays as _arr
from . import device as _dev
from .arrays import DeviceArray, zeros, lincomb_, hadamard_, copyto_, fill_, getblock, _i64arr
from .spaces import JetAbstractSpace, JetSpace, JetBSpace, dtype_code
from . import jets as _j
from .jets import Jet, Jop, JopLn, JopNl, JopAdjoint, mul_, domain, range_, jet, adjoint, point_, close

__all__ = [
    "JetBlock", "JopBlock", "blockop", "JopZeroBlock", "JopZeroBlock_df", "iszero", "JetBlock_f", "JetBlock_df",
    "JetBlock_df_adj", "nblocks_op", "getblock_op", "isblockop", "JopDiagonal", "JopIdentity", "diagonal_df",
    "diagonal_df_adj", "identity_df", "NativeBlockOp", "JopDense", "dense_df", "dense_df_adj", "JopSquare", "square_f",
    "square_df", "square_df_adj", "op_tune_get", "op_tune_set", "JopElementwise", "elementwise_f", "elementwise_df", "elementwise_df_adj", "elementwise_upstate",
]


# ------------------------------------------------------------------------------ native kinds -------
def JopZeroBlock_df(d, m, **kw):  # src/Jets.jl:942   d .= 0
    return fill_(d, 0)


def JopZeroBlock(dom: JetAbstractSpace, rng: JetAbstractSpace) -> JopLn:  # :941
    return JopLn(df=JopZeroBlock_df, dom=dom, rng=rng)


def iszero(A) -> bool:  # :949-951  a type test on df!, not a value test
    j = A if isinstance(A, Jet) else jet(A)
    return j.df is JopZeroBlock_df


def diagonal_df(d, m, *, diagonal, **kw):  # test/runtests.jl:3   d .= diagonal .* m
    return hadamard_(d, diagonal, m)


def diagonal_df_adj(m, d, *, diagonal, **kw):  # test/runtests.jl:4   m .= conj.(diagonal) .* d
    return hadamard_(m, diagonal, d, conj_x=True)


def JopDiagonal(diag: DeviceArray) -> JopLn:
    """Diagonal operator over a device array (the reference's test fixture JopFoo, test/runtests.jl:3-8;
    JetPack's JopDiagonal in docs/src/index.md:205)."""
    spc = _arr.space(diag) if isinstance(diag, _arr.BlockArray) else JetSpace(diag.dtype, *diag.shape)   # (weights over a block range keep its blocks)
    return JopLn(df=diagonal_df, df_adj=diagonal_df_adj, dom=spc, rng=spc, s={"diagonal": diag})


def identity_df(d, m, **kw):  # d .= m
    return copyto_(d, m)


def JopIdentity(spc: JetAbstractSpace) -> JopLn:
    return JopLn(df=identity_df, dom=spc, rng=spc)


def dense_df(d, m, *, A, **kw):  # test/runtests.jl:27   d .= A * m
    check(lib.jh_gemv(C.c_void_p(A.ptr), A.shape[0], A.shape[1], dtype_code(A.dtype), d.handle, m.handle, 0))
    return d


def dense_df_adj(m, d, *, A, **kw):  # test/runtests.jl:28   m .= A' * d
    check(lib.jh_gemv(C.c_void_p(A.ptr), A.shape[0], A.shape[1], dtype_code(A.dtype), m.handle, d.handle, 1))
    return m


def JopDense(A: DeviceArray) -> JopLn:
    """Dense matrix operator over a 2-D device array (the reference's test fixture JopBaz, test/runtests.jl:27-33)."""
    if A.ndim != 2:
        raise ValueError("JopDense needs a 2-D device array")
    dom, rng = JetSpace(A.dtype, A.shape[1]), JetSpace(A.dtype, A.shape[0])
    return JopLn(df=dense_df, df_adj=dense_df_adj, dom=dom, rng=rng, s={"A": A})


def square_f(d, m, **kw):  # test/runtests.jl:19   d .= m.^2
    return hadamard_(d, m, m)


def _need_point(mo, n):
    # the reference's jet holds an EMPTY mo until point! (src/Jets.jl:187): the broadcast then throws DimensionMismatch
    if mo is None or mo.length() != n:
        raise ValueError("DimensionMismatch: the Jacobian of a nonlinear operator needs a linearization point (point! / jacobian!)")
    return mo


def square_df(d, m, *, mo, **kw):  # test/runtests.jl:20   dd .= 2 .* mo .* dm
    return hadamard_(d, _need_point(mo, m.length()), m, twice_x=True)


def square_df_adj(m, d, *, mo, **kw):  # conj.(2 .* mo) .* dd  (== df! for real eltypes, src/Jets.jl:184-186)
    return hadamard_(m, _need_point(mo, d.length()), d, conj_x=True, twice_x=True)


def JopSquare(spc: JetAbstractSpace) -> JopNl:
    """The reference's nonlinear fixture JopBar (test/runtests.jl:19-24) as a device-native block kind."""
    return JopNl(f=square_f, df=square_df, df_adj=square_df_adj, dom=spc, rng=spc)


def elementwise_f(d, m, *, f_expr, params, **kw):  # f!(d, m): d .= f.(m), any elementwise expression (one fused JIT kernel)
    from .broadcast import broadcast_

    return broadcast_(d, f_expr, [m], params)


def elementwise_upstate(mo, s):  # upstate!(mo, s) (src/Jets.jl:297-301): refresh the Jacobian's diagonal IN PLACE
    from .broadcast import broadcast_

    broadcast_(s["diagonal"], s["jac_expr"], [mo], s["params"])
    s["pointed"][0] = True
    _j.POINT_GEN[0] += 1


def elementwise_df(d, m, *, diagonal, pointed, **kw):  # dd .= f'.(mo) .* dm
    if not pointed[0]:
        raise ValueError("DimensionMismatch: the Jacobian of a nonlinear operator needs a linearization point (point! / jacobian!)")
    return hadamard_(d, diagonal, m)


def elementwise_df_adj(m, d, *, diagonal, pointed, **kw):  # dm .= conj.(f'.(mo)) .* dd
    if not pointed[0]:
        raise ValueError("DimensionMismatch: the Jacobian of a nonlinear operator needs a linearization point (point! / jacobian!)")
    return hadamard_(m, diagonal, d, conj_x=True)


def JopElementwise(spc: JetAbstractSpace, f: str, jac: str, params=()) -> JopNl:
    """A nonlinear operator d .= f.(m) for ANY elementwise f, written the way the reference's users write one
    (`JopNl(f!, df!, df'!, upstate!)`, src/Jets.jl:196-207): `f` and `jac` are C expressions over x0 (the element of m /
    of the linearization point) and s0.. (`params`), e.g. JopElementwise(R, "exp(x0)", "exp(x0)") or
    JopElementwise(R, "s0*x0*x0*x0", "3*s0*x0*x0", [a]).  point! evaluates `jac` once into a diagonal kept in the state
    (one fused pass), so the Jacobian is a DIAGONAL operator: inside a block operator it is device-native and a tall
    operator of such children linearises onto the tall fast path (fused A'A, one-pass LSQR step)."""
    return JopNl(f=elementwise_f, df=elementwise_df, df_adj=elementwise_df_adj, upstate=elementwise_upstate, dom=spc, rng=spc,
                 s={"f_expr": f, "jac_expr": jac, "params": tuple(params), "diagonal": zeros(spc), "pointed": [False]})


def _native_desc(op: Jop):
    """(kind, adjoint_flag, coeff_array_or_None, scale) if `op` is device-native, else None."""
    adj = 0
    if isinstance(op, JopNl):  # a nonlinear child: f! in JetBlock_f!, its Jacobian about jet.mo in JetBlock_df!/df'!
        j = op.jet
        if j.f is square_f and j.df is square_df and j.df_adj is square_df_adj:
            return ("square", 0, None, 0.0)
        if j.f is elementwise_f and j.df is elementwise_df and j.df_adj is elementwise_df_adj:
            return ("diag", 0, j.s["diagonal"], 0.0)          # its Jacobian; f! itself runs through the host loop (host_f)
        return None
    if isinstance(op, JopAdjoint):
        adj, op = 1, op.op
    if not isinstance(op, JopLn):
        return None
    j = op.jet
    if j.df is JopZeroBlock_df:
        return ("zero", adj, None, 0.0)
    if j.df is identity_df:
        return ("identity", adj, None, 0.0)
    if j.df is diagonal_df and j.df_adj is diagonal_df_adj:
        return ("diag", adj, j.s["diagonal"], 0.0)
    if j.df is _j.constdiag_df and j.df_adj is _j.constdiag_df_adj:
        return ("scale", adj, None, j.s["a"])             # (as given: its TYPE says how Julia multiplies, _ffi.scalar_flags)
    if j.df is dense_df and j.df_adj is dense_df_adj:
        return ("dense", adj, j.s["A"], 0.0)
    return None


class NativeBlockOp:
    """Owns a jh_blockop handle for a matrix of device-native blocks."""

    _serials = itertools.count(1)

    def __init__(self, ops: np.ndarray, descs, dtype):
        nrow, ncol = ops.shape
        self.serial = next(NativeBlockOp._serials)              # never reused (an address is): what caches of things built on this handle key on
        self.has_zero = any(dsc[0] == "zero" for row in descs for dsc in row)
        arr = (BlockDesc * (nrow * ncol))()
        self._keep = []
        for jcol in builtins.range(ncol):
            for irow in builtins.range(nrow):
                kind, adj, coeff, scale = descs[irow][jcol]
                op = ops[irow, jcol]
                base = op.op if isinstance(op, JopAdjoint) else op
                b = arr[irow + jcol * nrow]  # column-major like a Julia Matrix
                b.kind, b.adjoint = KINDS[kind], adj
                b.coeff = coeff.ptr if coeff is not None else None
                sc = complex(scale)
                b.scale_re, b.scale_im, b.scale_flags = sc.real, sc.imag, scalar_flags(scale)
                b.nr, b.nc = range_(base).length(), domain(base).length()
                if coeff is not None:
                    self._keep.append(coeff)
        self.nonlinear = any(dsc[0] == "square" for row in descs for dsc in row)
        # children whose f! is not a device-native kind (JopElementwise: a JIT broadcast per child): JetBlock_f! loops on the host
        self.host_f = any(isinstance(op, JopNl) and op.jet.f is elementwise_f for op in ops.flat)
        self._point = None  # the device array the SQUARE blocks are linearised about (kept alive here)
        row_len = [range_(ops[i, 0]).length() for i in builtins.range(nrow)]
        col_len = [domain(ops[0, jc]).length() for jc in builtins.range(ncol)]
        self._h = C.c_void_p()
        if self._keep and _dev.several_contexts():          # the operator lives where its coefficients live
            _dev.context_use(_dev.context_of(self._keep[0]))
        check(lib.jh_blockop_create(nrow, ncol, arr, _i64arr(row_len), _i64arr(col_len), dtype_code(dtype), C.byref(self._h)))

    @property
    def handle(self):
        if self._h is None:
            raise ValueError("native block operator already closed")
        return self._h

    def mul(self, d, m):
        check(lib.jh_blockop_mul(self.handle, d.handle, m.handle))
        return d

    def mul_adj(self, m, d):
        check(lib.jh_blockop_mul_adj(self.handle, m.handle, d.handle))
        return m

    def f(self, d, m):
        check(lib.jh_blockop_f(self.handle, d.handle, m.handle))
        return d

    def point(self, mo):
        if self._point is None or self._point.ptr != mo.ptr:
            check(lib.jh_blockop_point(self.handle, mo.handle))
            self._point = mo
        return self

    def normal_mul(self, y, m):
        check(lib.jh_blockop_normal_mul(self.handle, y.handle, m.handle))
        return y

    def tune_get(self, name: str) -> int:
        v = C.c_int64(0)
        check(lib.jh_blockop_tune_get(self.handle, name.encode(), C.byref(v)))
        return v.value

    def tune_set(self, name: str, value: int):
        check(lib.jh_blockop_tune_set(self.handle, name.encode(), int(value)))
        return self

    def close(self):
        if self._h is not None:
            h, self._h = self._h, None
            lib.jh_blockop_destroy(h)
        self._keep = []
        self._point = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _NativeCell:
    """Mutable slot in a block jet's state holding its lazily built NativeBlockOp (closures only see
    the state as keyword arguments, src/Jets.jl:391, so the cache travels with the state)."""

    def __init__(self):
        self.value = "unset"

    def __deepcopy__(self, memo):  # copy(jet) deep-copies the state (src/Jets.jl:230): fresh cache
        return _NativeCell()

    def close(self):
        if isinstance(self.value, NativeBlockOp):
            self.value.close()
        self.value = "unset"
        self.__dict__.pop("fpacks", None)
        self.__dict__.pop("ppack", None)


def _desc_signature(descs):
    """What the device handle captured of the children's states: kinds, adjoint flags, coefficient ADDRESSES, scalars."""
    return tuple((dsc[0], dsc[1], None if dsc[2] is None else (dsc[2].ptr, dsc[2].length()), complex(dsc[3]))
                 for row in descs for dsc in row)


def _native_op(cell, ops, dtype):
    """The cached NativeBlockOp of a block jet, or None when some block is not device-native.

    The handle holds raw coefficient pointers taken from the children's states.  The reference's `state!` merges a new
    state that takes effect on the next `mul!` (src/Jets.jl:272, 391), so after any `state_()` (jets.STATE_GEN moved)
    the children are described again and the handle is rebuilt when a kind, a coefficient address or a scalar changed."""
    if cell is None:
        cell = _NativeCell()
    gen = _j.STATE_GEN[0]
    if cell.value != "unset" and cell.__dict__.get("gen") == gen:
        return cell.value
    descs = [[_native_desc(ops[i, jc]) for jc in builtins.range(ops.shape[1])] for i in builtins.range(ops.shape[0])]
    native = not any(dsc is None for row in descs for dsc in row)
    sig = _desc_signature(descs) if native else None
    if cell.value != "unset" and cell.__dict__.get("sig") == sig:
        cell.gen = gen                                         # an unrelated state changed: the handle is still right
        return cell.value
    if cell.value != "unset":
        cell.close()                                           # stale pointers: drop the handle and every table built on it
    cell.value = NativeBlockOp(ops, descs, dtype) if native else None
    cell.gen, cell.sig = gen, sig
    return cell.value


# ------------------------------------------------------------------------------ construction -------
def _as_matrix(ops) -> np.ndarray:
    if isinstance(ops, np.ndarray) and ops.dtype == object:
        m = ops
    else:
        rows = list(ops)
        if rows and isinstance(rows[0], Jop):  # vector -> N x 1   (src/Jets.jl:933)
            m = np.empty((len(rows), 1), dtype=object)
            for i, op in enumerate(rows):
                m[i, 0] = op
            return m
        m = np.empty((len(rows), len(rows[0])), dtype=object)
        for i, row in enumerate(rows):
            if len(row) != m.shape[1]:
                raise ValueError("ragged block matrix")
            for jc, op in enumerate(row):
                m[i, jc] = op
    if m.ndim == 1:
        m = m.reshape(-1, 1)
    return m


def JetBlock(ops, dadom: bool = False, **kwargs) -> Jet:  # :926-930
    ops = _as_matrix(ops)
    nrow, ncol = ops.shape
    dom = domain(ops[0, 0]) if (ncol == 1 and not dadom) else JetBSpace([domain(ops[0, jc]) for jc in builtins.range(ncol)])
    rng = JetBSpace([range_(ops[i, 0]) for i in builtins.range(nrow)])
    return Jet(f=JetBlock_f, df=JetBlock_df, df_adj=JetBlock_df_adj, dom=dom, rng=rng,
               s={"ops": ops, "dom": dom, "rng": rng, "_native": _NativeCell(), **kwargs})


def JopBlock(ops, **kwargs) -> Jop:  # :931-933
    ops = _as_matrix(ops)
    if all(isinstance(op, (JopLn, JopAdjoint)) for op in ops.flat):
        return JopLn(JetBlock(ops, **kwargs))
    return JopNl(JetBlock(ops, **kwargs))


def blockop(ops, **kwargs) -> Jop:
    """@blockop ops [kw...]  (src/Jets.jl:953-986)."""
    return JopBlock(ops, **kwargs)


# ------------------------------------------------------------------------------ the block loops ----
def _pointed_native(nat, ops, mo):
    """`nat` linearised about `mo`, or None when some nonlinear child sits at a point of its own (someone called point!
    on the child, not on the block jet): then the per-child loop below is the faithful path."""
    if not nat.host_f and not nat.nonlinear:
        return nat
    # the checks below walk every child: remember their outcome until some jet's point changes (jets.POINT_GEN)
    key = (_j.POINT_GEN[0], None if mo is None else (mo.ptr, mo.length()))
    hit = getattr(nat, "_pt_cache", None)
    if hit is not None and hit[0] == key:
        return nat if hit[1] else None
    res = _pointed_native_check(nat, ops, mo)
    nat._pt_cache = (key, res is not None)
    return res


def _pointed_native_check(nat, ops, mo):
    if nat.host_f:                                             # JopElementwise children carry their own (refreshed in place) diagonal
        for op in ops.flat:
            if isinstance(op, JopNl) and op.jet.f is elementwise_f and not op.jet.s["pointed"][0]:
                return None
    if not nat.nonlinear:
        return nat
    if mo is None or mo.length() != sum(domain(ops[0, jc]).length() for jc in builtins.range(ops.shape[1])):
        return None
    for jc in builtins.range(ops.shape[1]):
        want = getblock(mo, jc).ptr
        for i in builtins.range(ops.shape[0]):
            op = ops[i, jc]
            if isinstance(op, JopNl) and (op.jet.mo is None or op.jet.mo.ptr != want):
                return None
    return nat.point(mo)


def JetBlock_f(d, m, *, ops, dom, rng, _native=None, **kw):  # :988-1008
    nat = _native_op(_native, ops, rng.eltype())
    if nat is not None and not nat.host_f:
        return nat.f(d, m)  # one fused launch, same loop order and rounding
    nrow, ncol = ops.shape
    memo = _native.__dict__ if _native is not None else {}
    all_elementwise = memo.get("all_elementwise")             # (decided once per operator: a loop over 16 384 children costs a millisecond per call)
    if all_elementwise is None:
        all_elementwise = memo["all_elementwise"] = ncol == 1 and all(isinstance(op, JopNl) and op.jet.f is elementwise_f for op in ops.flat)
    if all_elementwise:
        from .broadcast import pack_many, run_packed   # every child's f! (1003) in one trip through the ABI (one launch when alike)

        packs = _native.__dict__.setdefault("fpacks", {}) if _native is not None else {}
        key = (d.ptr, m.ptr, d.length(), _j.STATE_GEN[0])     # state!(child, ...) may change a child's parameters
        pack = packs.get(key)
        if pack is None:                                     # the argument tables of this (d, m) pair: built once, replayed afterwards
            if len(packs) >= 4:
                packs.clear()
            pack = packs[key] = pack_many((getblock(d, i), ops[i, 0].jet.s["f_expr"], [m], ops[i, 0].jet.s["params"])
                                          for i in builtins.range(nrow))
        run_packed(pack)
        return d
    dtmp = zeros(range_(ops[0, 0])) if ncol > 1 else None
    for i in builtins.range(nrow):
        _d = getblock(d, i)
        if ncol > 1 and dtmp.shape != tuple(range_(ops[i, 0]).size()):
            dtmp = zeros(range_(ops[i, 0]))
        for jc in builtins.range(ncol):
            _m = getblock(m, jc)
            if ncol > 1:
                lincomb_(_d, [1.0, 1.0], [_d, mul_(dtmp, ops[i, jc], _m)])  # _d .+= mul!(dtmp, op, _m)   (:1001)
            else:
                mul_(_d, ops[i, jc], m)  # (:1003)
    return d


def JetBlock_df(d, m, *, ops, dom, rng, _native=None, **kw):  # :1010-1032
    nat = _native_op(_native, ops, rng.eltype())
    if nat is not None:
        nat = _pointed_native(nat, ops, kw.get("mo"))
    if nat is not None:
        return nat.mul(d, m)  # one fused launch, same loop order and rounding
    nrow, ncol = ops.shape
    dtmp = zeros(range_(ops[0, 0])) if ncol > 1 else None  # :1012-1014
    for i in builtins.range(nrow):  # :1015
        _d = getblock(d, i)
        if ncol > 1 and dtmp.shape != tuple(range_(ops[i, 0]).size()):  # :1018
            dtmp = zeros(range_(ops[i, 0]))
        for jc in builtins.range(ncol):  # :1020
            _m = getblock(m, jc)
            if not iszero(ops[i, jc]):  # :1022
                if ncol > 1:
                    lincomb_(_d, [1.0, 1.0], [_d, mul_(dtmp, JopLn(ops[i, jc]), _m)])  # :1024
                else:
                    mul_(_d, JopLn(ops[i, jc]), _m)  # :1026
    return d


def JetBlock_df_adj(m, d, *, ops, dom, rng, _native=None, **kw):  # :1034-1057
    nat = _native_op(_native, ops, rng.eltype())
    if nat is not None:
        nat = _pointed_native(nat, ops, kw.get("mo"))
    if nat is not None:
        return nat.mul_adj(m, d)
    nrow, ncol = ops.shape
    mtmp = zeros(domain(ops[0, 0])) if nrow > 1 else None  # :1036-1038
    for jc in builtins.range(ncol):  # :1039
        _m = getblock(m, jc)
        if nrow > 1:
            fill_(_m, 0)  # :1042
            if mtmp.shape != tuple(domain(ops[0, jc]).size()):  # :1043
                mtmp = zeros(domain(ops[0, jc]))
        for i in builtins.range(nrow):  # :1045
            _d = getblock(d, i)
            if not iszero(ops[i, jc]):  # :1047
                if nrow > 1:
                    lincomb_(_m, [1.0, 1.0], [_m, mul_(mtmp, adjoint(JopLn(ops[i, jc])), _d)])  # :1049
                else:
                    mul_(_m, adjoint(JopLn(ops[i, jc])), _d)  # :1051
    return m


def point_block(j: Jet, mo):  # :1059-1066
    ops = j.s["ops"]
    cell = j.s.get("_native")
    # Re-pointing at the SAME vector (a Gauss-Newton loop updates its model in place): every child already holds its block of
    # `mo`; only the children's upstate!s have to run again -- replay their packed batch instead of walking the children.
    if cell is not None and j.mo is mo:
        hit = cell.__dict__.get("ppack")
        if hit is not None and hit[0] == (mo.ptr, mo.length(), _j.POINT_GEN[0], _j.STATE_GEN[0]):
            from .broadcast import run_packed

            run_packed(hit[1])
            return j
    j.mo = mo
    _j.POINT_GEN[0] += 1
    jobs, all_batched = [], True
    for jc in builtins.range(ops.shape[1]):
        mo_j = getblock(mo, jc)
        for i in builtins.range(ops.shape[0]):
            cj = jet(ops[i, jc])
            if cj.upstate is elementwise_upstate and cj.f is elementwise_f:   # point!(child) = set mo + upstate!; the upstate!s are batched
                cj.mo = mo_j
                jobs.append((cj.s["diagonal"], cj.s["jac_expr"], [mo_j], cj.s["params"]))
                cj.s["pointed"][0] = True
            else:
                point_(cj, mo_j)
                # a child with an upstate! of its own, or a composite / sum / block child (whose point! recomputes the points of
                # its members), must see every point!: no replay for such operators
                all_batched = all_batched and cj.upstate is _j._default_upstate and cj.f not in (_j.JetComposite_f, _j.JetSum_f, JetBlock_f)
    _j.POINT_GEN[0] += 1
    if jobs:
        from .broadcast import pack_many, run_packed

        pack = pack_many(jobs)
        run_packed(pack)
        if cell is not None and all_batched:
            cell.__dict__["ppack"] = ((mo.ptr, mo.length(), _j.POINT_GEN[0], _j.STATE_GEN[0]), pack)
    return j


def close_block(j: Jet):  # :1120-1124
    ops = j.s["ops"]
    for op in ops.flat:
        close(op)
    cell = j.s.get("_native")
    if cell is not None:
        cell.close()
    return None


def nblocks_op(A, i=None):  # :1074-1077
    r, dm = _arr.nblocks(range_(A)), _arr.nblocks(domain(A))
    if i is None:
        return (r, dm)
    return r if i == 1 else dm


def overwrites_its_whole_range(A) -> bool:
    """True when mul!(d, A, m) is known to write every element of d whatever d held: a sum or a composite (they set their output themselves),
    or a one-column LINEAR block operator (not adjointed) whose children are all device-native and none a zero block -- each row is one
    overwrite (src/Jets.jl:1026)."""
    inner = A.op if isinstance(A, JopAdjoint) else A
    if isinstance(inner, (JopLn, JopNl)) and inner.jet.f in (_j.JetSum_f, _j.JetComposite_f):
        return True                    # a sum starts with `d .= 0` (src/Jets.jl:631, 640, 649), a composite ends with `d .= chain` (526, 532, 538)
    if not isinstance(A, JopLn) or not isblockop(A):
        return False
    ops = A.jet.s["ops"]
    if ops.shape[1] != 1:
        return False
    # (the answer is a property of the children's KINDS, fixed when the operator is built: remembered in its state -- `A * m` asks on every call, and
    # walking 65 536 children cost 17 ms per call where the forward takes 0.09; tools/prof_host.py)
    memo = A.jet.s.get("_overwrites")
    if memo is not None:
        return memo
    res = True
    for i in builtins.range(ops.shape[0]):
        desc = _native_desc(ops[i, 0])
        if desc is None or desc[0] not in ("identity", "diag", "scale", "dense") or isinstance(ops[i, 0], JopNl):
            res = False
            break
    A.jet.s["_overwrites"] = res
    return res


def isblockop(A) -> bool:  # :1097-1098
    return isinstance(A, Jop) and jet(A).f is JetBlock_f


def getblock_op(A, i: int, jc: int, kind=None):  # :1085-1090, 1100-1110
    if kind is JopNl:  # getblock(JopNl, A, i, j)
        r = getblock_op(jet(A), i, jc)
        if not isinstance(r, JopNl):
            raise TypeError("block is not a JopNl")
        return r
    if kind is JopLn:
        return JopLn(getblock_op(jet(A), i, jc))
    if isinstance(A, Jet):
        if A.f is JetBlock_f:
            return A.s["ops"][i, jc]  # :1085
        if A.f is _j.JetComposite_f:  # :1100-1110
            ops = [getblock_op(op, i, jc) if isblockop(op) else op for op in A.s["ops"]]
            out = ops[0]
            for op in ops[1:]:
                out = _j.compose(out, op)
            return out
        raise TypeError("not a block operator")
    if isinstance(A, JopAdjoint):
        return adjoint(getblock_op(A.op, jc, i))  # :1088
    if isinstance(A, JopLn):
        return JopLn(getblock_op(A.jet, i, jc))  # :1086
    return getblock_op(A.jet, i, jc)  # :1087


def op_tune_get(A, name: str):
    """Per-operator measured choice of a block operator's device handle (jh_blockop_tune_get: "fwd_walk", "fwd_trials",
    "upd_walk"); None when the operator has no device handle."""
    nat = _native_op(jet(A).s.get("_native"), jet(A).s["ops"], jet(A).rng.eltype()) if isblockop(A) else None
    return None if nat is None else nat.tune_get(name)


def op_tune_set(A, name: str, value: int):
    nat = _native_op(jet(A).s.get("_native"), jet(A).s["ops"], jet(A).rng.eltype()) if isblockop(A) else None
    if nat is None:
        raise ValueError("operator has no device handle")
    nat.tune_set(name, value)
    return A


# ------------------------------------------------------------------------------ fused chains -------
def _tall_native(op):
    """The NativeBlockOp of a tall (one-column) device-native block operator, else None.  A Jacobian whose nonlinear
    children were never pointed (or sit at points of their own) is NOT native here: the unfused path then raises the
    reference's DimensionMismatch / runs the per-child loop, and so must the fused chains, sums and LSQR."""
    if not (isinstance(op, JopLn) and isblockop(op)):
        return None
    j = op.jet
    if j.s["ops"].shape[1] != 1:
        return None
    nat = _native_op(j.s.get("_native"), j.s["ops"], j.rng.eltype())
    if nat is None:
        return None
    return _pointed_native(nat, j.s["ops"], j.mo)


def _grid_native(op):
    """The NativeBlockOp of a device-native N x K block operator with N >= 2 and K = 2 .. 4 (the shapes whose fused A'A the library has,
    jh_grid_normal.hip), else None."""
    if not (isinstance(op, JopLn) and isblockop(op)):
        return None
    j = op.jet
    if not (2 <= j.s["ops"].shape[1] <= 4 and j.s["ops"].shape[0] >= 2):
        return None
    nat = _native_op(j.s.get("_native"), j.s["ops"], j.rng.eltype())
    return None if nat is None else _pointed_native(nat, j.s["ops"], j.mo)


def _real_scale(op, allow_wide=False):
    """The real scalar a of an `a*` operator (src/Jets.jl:1159-1162) when the fused kernels reproduce Julia's arithmetic for it, else None.
    They multiply by T(a), part by part: right for a Real scalar of the elements' precision.  A Complex scalar (full product, even with
    a zero imaginary part) takes the chain; a Float64 one against 32-bit elements (promoted arithmetic) is fused only where the caller
    can say so (allow_wide: the result is then the pair (a, JH_SCALAR_* flags) for jh_blockop_mul[_adj]_scaled)."""
    if isinstance(op, JopAdjoint):
        op = op.op                                    # conj(a) == a for a real a
    if isinstance(op, JopLn) and op.jet.df is _j.constdiag_df and op.jet.df_adj is _j.constdiag_df_adj:
        a = op.jet.s["a"]
        fl = scalar_flags(a)
        T = np.dtype(domain(op).eltype())
        if T.itemsize // (2 if T.kind == "c" else 1) == 8:
            fl &= ~SCALAR_WIDE                            # nothing is wider than 64-bit elements
        if (fl & ~SCALAR_WIDE) or complex(a).imag != 0.0:
            return None
        if allow_wide:
            return float(complex(a).real), fl
        return None if fl else float(complex(a).real)
    return None


def try_fused_chain(out, x, ops: Sequence[Jop]):
    """Two-stage chains of JetComposite_df / df' (src/Jets.jl:530-540) that one kernel computes with the unfused
    chain's exact rounding sequence; returns None when the chain does not qualify:
      (A', A)   normal operator, coefficients read once            -> jh_blockop_normal_mul (tall operators; N x (2 .. 4) grids of diagonals)
      (a, A)    scalar * operator, forward                         -> jh_blockop_mul_scaled
      (A', a')  scalar * operator, adjoint: A'(conj(a) d)          -> jh_blockop_mul_adj_scaled
    for a tall all-diagonal device-native block operator A and a Real scalar a (of the elements' precision or wider: the scalar's
    type goes along, so that the fused pass has the bits of the chain)."""
    if len(ops) != 2:
        return None
    left, right = ops
    try:
        if isinstance(left, JopAdjoint) and left.op is right:
            nat = _tall_native(right)
            if nat is not None and right.jet.s["ops"].shape[0] >= 2:
                return nat.normal_mul(out, x)
            if nat is None:
                # round 6: an N x K grid of equal diagonals, K = 2 .. 4 (multi-parameter operators): one pass over the coefficients
                # (jh_grid_normal.hip); the library declines anything else (JH_ERR_UNSUPPORTED -> the reference's chain)
                nat = _grid_native(right)
                if nat is not None:
                    return nat.normal_mul(out, x)
            return None
        a = _real_scale(left, allow_wide=True)
        if a is not None and not isinstance(right, JopAdjoint):
            nat = _tall_native(right)
            if nat is not None:
                check(lib.jh_blockop_mul_scaled(nat.handle, out.handle, x.handle, a[0], a[1]))
                return out
            return None
        a = _real_scale(right, allow_wide=True)
        if a is not None and isinstance(left, JopAdjoint):
            nat = _tall_native(left.op)
            if nat is not None:
                check(lib.jh_blockop_mul_adj_scaled(nat.handle, out.handle, x.handle, a[0], a[1]))
                return out
    except JetsHipError as e:  # JH_ERR_UNSUPPORTED: not eligible for the fused kernel (mixed kinds, ragged blocks)
        if e.status == 4:
            return None
        raise
    return None


def try_fused_sum(out, x, ops: Sequence[Jop], sgns: Sequence[str], transposed: bool):
    """JetSum_df / JetSum_df' (src/Jets.jl:639-655) when every term is a tall all-diagonal device-native block operator
    A_k or a Real scalar times one (the composite (a, A_k)): one fused launch (jh_blocksum_mul_typed / jh_blocksum_mul_adj_typed).
    The scalars' Julia types go along (round 5): `1.0*A1 - 2.0*A2 + 3.0*A3` (src/Jets.jl:686) with numpy float64 scalars on Float32
    operators is WIDE -- promoted products, one rounding -- and still one pass.  Returns None when the sum does not qualify."""
    if not (1 <= len(ops) <= 4096):                          # any number of terms: the library groups them by sixteen
        return None
    nats, scales, flags = [], [], []
    for op in ops:
        op = JopLn(op)
        if isinstance(op, JopAdjoint):
            return None
        scale, fl = 1.0, 0
        if op.jet.f is _j.JetComposite_f:
            inner = op.jet.s["ops"]
            if len(inner) != 2 or isinstance(inner[0], JopAdjoint):
                return None
            a = _real_scale(inner[0], allow_wide=True)
            if a is None:
                return None
            (scale, fl), op = a, inner[1]
        nat = _tall_native(op) if not isinstance(op, JopAdjoint) else None
        if nat is None:
            return None
        nats.append(nat)
        scales.append(scale)
        flags.append(fl)
    k = len(nats)
    hs = (C.c_void_p * k)(*[n.handle for n in nats])
    sc = (C.c_double * k)(*scales)
    fg = (C.c_int32 * k)(*flags)
    sg = (C.c_double * k)(*[1.0 if s == _j.PLUS else -1.0 for s in sgns])
    fn = lib.jh_blocksum_mul_adj_typed if transposed else lib.jh_blocksum_mul_typed
    try:
        check(fn(k, hs, sc, fg, sg, out.handle, x.handle))
    except JetsHipError as e:
        if e.status == 4:
            return None
        raise
    return out
