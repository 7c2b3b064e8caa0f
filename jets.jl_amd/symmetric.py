"""Symmetric spaces / arrays (src/Jets.jl:404-516) over device storage.

A `SymmetricArray` of logical size `n` stores only the block `M` (its `parent`, a device array); an element outside the
stored block is the conjugate of the stored element `map` points at -- the layout of the spectrum of a real signal.  Like the
reference, broadcast acts on the parents (src/Jets.jl:508-512: one fused device pass) and scalar indexing goes through the
map (a host round trip per element: the slow path, as in the reference).  `norm` counts every LOGICAL element, which is what
the reference's generic AbstractArray `norm` does (test/runtests.jl:243-249): the stored elements are weighted by how many
logical elements they stand for -- one device pass over the parent and a weight vector built once per space.

Not on the block-operator hot path (a block operator never sees a symmetric space); here so that a user of the reference
finds the whole exported API.
"""
from __future__ import annotations

import builtins
import itertools

import numpy as np

from .spaces import JetSSpace, JetSpace
from . import arrays as _arr
from .arrays import DeviceArray, LinExpr

__all__ = ["SymmetricArray", "symspace"]

_weights = {}


def symspace(*_a, **_k):  # src/Jets.jl:444: operators with symmetric ranges provide their own method
    return None


class SymmetricArray:
    """SymmetricArray{T,N,F} (src/Jets.jl:446-450): `A` is the stored block (device), `n` the logical size."""

    __array_ufunc__ = None

    def __init__(self, A: DeviceArray, n, map):  # noqa: A002
        self.A, self.n, self.map = A, tuple(int(k) for k in n), map

    # -- array interface (:456-494)
    @property
    def shape(self):
        return self.n

    @property
    def dtype(self):
        return self.A.dtype

    @property
    def parent(self) -> DeviceArray:  # :454
        return self.A

    def length(self) -> int:
        return int(np.prod(self.n, dtype=np.int64))

    def _cart(self, I):
        if isinstance(I, (int, np.integer)):  # linear index, column-major (:469-472)
            I = np.unravel_index(int(I), self.n, order="F")
        I = tuple(int(k) for k in I)
        if len(I) != len(self.n) or any(not (0 <= k < m) for k, m in zip(I, self.n)):
            raise IndexError(I)
        return I

    def _stored(self, I):
        """(index into the parent, conjugate?) for the logical index I (:460-467)."""
        I = self._cart(I)
        if any(k >= m for k, m in zip(I, self.A.shape)):
            return tuple(int(k) for k in self.map(I)), True
        return I, False

    def __getitem__(self, I):
        J, cj = self._stored(I)
        v = self.A._download(int(np.ravel_multi_index(J, self.A.shape, order="F")), 1)[0]
        return np.conj(v) if cj else v

    def __setitem__(self, I, v):  # :474-482
        J, cj = self._stored(I)
        v = np.conj(v) if cj else v
        self.A._upload(np.asarray([v], dtype=self.A.dtype), int(np.ravel_multi_index(J, self.A.shape, order="F")))

    def to_numpy(self) -> np.ndarray:
        """The full logical array on the host (column-major)."""
        a = self.A.to_numpy()
        out = np.empty(self.n, dtype=a.dtype, order="F")
        for I in itertools.product(*[builtins.range(k) for k in self.n]):
            J, cj = self._stored(I)
            out[I] = np.conj(a[J]) if cj else a[J]
        return out

    # -- broadcast on the parents (:496-512)
    def _lin(self):
        return [(1.0, self)]

    def __mul__(self, a):
        if isinstance(a, (int, float, complex, np.number)):
            return _SymExpr([(a, self)])
        return NotImplemented

    __rmul__ = __mul__

    def __add__(self, other):
        return _SymExpr([(1.0, self)]) + other

    def __sub__(self, other):
        return _SymExpr([(1.0, self)]) - other

    def __neg__(self):
        return _SymExpr([(-1.0, self)])

    def assign(self, expr) -> "SymmetricArray":
        """`self .= expr` (src/Jets.jl:508-512): the broadcast runs on the stored blocks."""
        if isinstance(expr, SymmetricArray):
            expr = _SymExpr([(1.0, expr)])
        if isinstance(expr, _SymExpr):
            _arr.lincomb_(self.A, [c for c, _ in expr.terms], [x.A for _, x in expr.terms])
        elif isinstance(expr, (int, float, complex, np.number)):
            _arr.fill_(self.A, expr)
        else:
            raise TypeError(type(expr))
        return self

    def __repr__(self):
        return f"SymmetricArray({self.dtype.name}, size {self.n}, stored {self.A.shape})"


class _SymExpr:
    """a*u .+ b*v .+ ... over symmetric arrays: materialises into a SymmetricArray (similar(find_symmetricarray(bc)), :498)."""

    __array_ufunc__ = None

    def __init__(self, terms):
        self.terms = list(terms)

    def __add__(self, other):
        other = _SymExpr([(1.0, other)]) if isinstance(other, SymmetricArray) else other
        return _SymExpr(self.terms + other.terms) if isinstance(other, _SymExpr) else NotImplemented

    def __sub__(self, other):
        other = _SymExpr([(1.0, other)]) if isinstance(other, SymmetricArray) else other
        return _SymExpr(self.terms + [(-c, x) for c, x in other.terms]) if isinstance(other, _SymExpr) else NotImplemented

    def __mul__(self, a):
        if isinstance(a, (int, float, complex, np.number)):
            return _SymExpr([(a * c, x) for c, x in self.terms])
        return NotImplemented

    __rmul__ = __mul__

    def __neg__(self):
        return _SymExpr([(-c, x) for c, x in self.terms])

    def materialize(self) -> SymmetricArray:
        return similar_sym(self.terms[0][1]).assign(self)


# ------------------------------------------------------------------------------ factories (:514-516) ----
def _make(R: JetSSpace, fill):
    return SymmetricArray(fill(JetSpace(R.eltype(), *R.M)), R.n, R.map)


def zeros_sym(R: JetSSpace):
    return _make(R, _arr.zeros)


def ones_sym(R: JetSSpace):
    return _make(R, _arr.ones)


def rand_sym(R: JetSSpace, **kw):
    return _make(R, lambda S: _arr.rand(S, **kw))


def randn_sym(R: JetSSpace, **kw):
    return _make(R, lambda S: _arr.randn(S, **kw))


def similar_sym(x: SymmetricArray, T=None):
    """similar(A::SymmetricArray[, T]) (:484-486): complex -> a SymmetricArray, real -> a plain array of the logical size."""
    T = x.dtype if T is None else np.dtype(T)
    if T.kind == "c":
        return SymmetricArray(_arr.similar(x.A, T), x.n, x.map)
    return _arr.Array(JetSpace(T, *x.n))


def space_sym(x: SymmetricArray) -> JetSSpace:  # :452
    return JetSSpace(x.dtype, x.n, x.A.shape, x.map)


def _multiplicity(x: SymmetricArray) -> DeviceArray:
    """How many logical elements each stored element stands for (a real device array shaped like the parent)."""
    key = (x.n, tuple(x.A.shape), id(x.map), x.dtype.str)
    w = _weights.get(key)
    if w is None:
        cnt = np.zeros(x.A.shape, dtype=np.float64, order="F")
        for I in itertools.product(*[builtins.range(k) for k in x.n]):
            J, _ = x._stored(I)
            cnt[J] += 1
        real_t = np.float32 if x.dtype == np.dtype(np.complex64) else np.float64
        w = _arr.from_numpy(cnt.astype(real_t))
        _weights[key] = w
    return w


def norm_sym(x: SymmetricArray, p: float = 2):
    """norm(x, p) over the LOGICAL elements (the reference's generic AbstractArray norm; test/runtests.jl:243-249)."""
    from .broadcast import broadcast_

    a = _arr.abs_(x.A)                                          # |stored|, real
    w = _multiplicity(x)
    if p == np.inf:
        return _arr.norm(a, np.inf)
    if p == -np.inf:
        return _arr.norm(a, -np.inf)
    if p == 0:
        broadcast_(a, "x1 * (x0 != 0 ? 1 : 0)", [a, w])
        return _arr.norm(a, 1)
    if p == 1:
        broadcast_(a, "x1 * x0", [a, w])
        return _arr.norm(a, 1)
    if p == 2:
        broadcast_(a, "x1 * x0 * x0", [a, w])
        return type(_arr.norm(a, 1))(np.sqrt(float(_arr.norm(a, 1))))
    broadcast_(a, "x1 * pow(x0, s0)", [a, w], [float(p)])
    return type(_arr.norm(a, 1))(float(_arr.norm(a, 1)) ** (1.0 / p))


def abs_sym(x: SymmetricArray) -> np.ndarray:
    """abs.(x): a plain real array of the logical size (similar(x, Real) is an Array, :485; test/runtests.jl:288-292)."""
    return np.abs(x.to_numpy())
