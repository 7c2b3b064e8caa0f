"""BlockArray broadcast for arbitrary elementwise expressions (src/Jets.jl:889-911), fused into ONE kernel per expression.

In the reference `d .= exp.(a .* u) .+ v ./ w` is one compiled loop over the blocks whatever the expression; here the
expression is printed as C, compiled once with hiprtc for gfx950 (jh_bcast_compile) and streamed over the slabs in one
pass.  Two entry points:

    Jets.broadcast_(d, "exp(s0*x0) + x1/x2", [u, v, w], [a])          # the raw form (what the Julia binding emits)
    Jets.assign_(d, Jets.bc.exp(a * Jets.lazy(u)) + Jets.lazy(v) / Jets.lazy(w))   # a lazy expression tree, like Broadcasted

`a*u + b*v` on device vectors keeps going through jh_lincomb (arrays.LinExpr): same bits, no JIT.
"""
from __future__ import annotations

import ctypes as C
import itertools
import numbers

import numpy as np

from ._ffi import lib, check
from .spaces import dtype_code
from .arrays import _DevVec, similar

__all__ = ["broadcast_", "broadcast_many_", "pack_many", "run_packed", "lazy", "assign_", "bc", "BExpr"]

_programs = {}
_scalar_ids = itertools.count()


def _real_mask(dtype, vecs, scalars=()) -> int:
    """Bit k set: operand k is REAL of the matching precision in a complex broadcast (a real mask on a complex vector,
    src/Jets.jl:899-904); bit len(vecs) + k: scalar k is a real number (Julia's `a::Real * z` works part by part: no 0 * Inf from
    an imaginary part the scalar does not have).  Any other eltype mix is refused by the library."""
    dt = np.dtype(dtype)
    if dt.kind != "c":
        return 0
    real = np.dtype(np.float32 if dt == np.complex64 else np.float64)
    mask = sum(1 << k for k, v in enumerate(vecs) if np.dtype(v.dtype) == real)
    return mask | sum(1 << (len(vecs) + k) for k, a in enumerate(scalars) if not isinstance(a, (complex, np.complexfloating)))


def _wide_mask(dtype, scalars=()) -> int:
    """Bit k set: scalar k is Float64-based (numpy float64 / complex128 -- Julia's Float64 / ComplexF64) in a 32-bit program: promoted
    arithmetic, one rounding on the store (include/jetship.h JH_SCALAR_WIDE; _ffi.scalar_flags)."""
    if np.dtype(dtype) not in (np.dtype(np.float32), np.dtype(np.complex64)):
        return 0
    return sum(1 << k for k, a in enumerate(scalars) if isinstance(a, (np.float64, np.complex128)))


def _program(expr: str, dtype, nvec: int, nscal: int, real_mask: int = 0, wide_mask: int = 0):
    key = (expr, np.dtype(dtype).str, nvec, nscal, real_mask, wide_mask)
    h = _programs.get(key)
    if h is None:
        h = C.c_void_p()
        if wide_mask:
            check(lib.jh_bcast_compile_typed(expr.encode(), dtype_code(dtype), nvec, real_mask, nscal, wide_mask, C.byref(h)))
        elif real_mask:
            check(lib.jh_bcast_compile_mixed(expr.encode(), dtype_code(dtype), nvec, real_mask, nscal, C.byref(h)))
        else:
            check(lib.jh_bcast_compile(expr.encode(), dtype_code(dtype), nvec, nscal, C.byref(h)))
        _programs[key] = h
    return h


def pack_many(jobs):
    """[(dst, expr, vecs, scalars), ...] packed into the argument arrays of jh_bcast_apply_many.  A caller that issues the
    same batch again and again (F(m) of a tall nonlinear operator into the same vectors) keeps the pack and calls
    run_packed: no per-item host work."""
    jobs = list(jobs)
    progs = (C.c_void_p * max(len(jobs), 1))()
    dsts = (C.c_void_p * max(len(jobs), 1))()
    xs, sc, keep = [], [], []
    for k, (dst, expr, vecs, scalars) in enumerate(jobs):
        progs[k] = _program(expr, dst.dtype, len(vecs), len(scalars), _real_mask(dst.dtype, vecs, scalars), _wide_mask(dst.dtype, scalars))
        dsts[k] = dst.handle
        keep.append(dst)
        for v in vecs:
            xs.append(v.handle)
            keep.append(v)
        for a in scalars:
            a = complex(a)
            sc.append(a.real)
            sc.append(a.imag)
    xa = (C.c_void_p * max(len(xs), 1))(*xs)
    sa = (C.c_double * max(len(sc), 1))(*sc)
    return (len(jobs), progs, dsts, xa, sa, keep)


def run_packed(pack):
    if pack[0]:
        check(lib.jh_bcast_apply_many(pack[0], pack[1], pack[2], pack[3], pack[4]))


def broadcast_many_(jobs):
    """[(dst, expr, vecs, scalars), ...] in ONE trip through the ABI (jh_bcast_apply_many): items that share the program run
    as one launch, the rest are enqueued back to back -- what a tall nonlinear operator needs for F(m) and point! over
    hundreds of children."""
    run_packed(pack_many(jobs))


def broadcast_(dst: _DevVec, expr: str, vecs=(), scalars=()):
    """dst .= expr over x0..x{k-1} = elements of `vecs`, s0.. = `scalars` (converted to dst's eltype; a numpy float64 / complex128 against 32-bit elements stays a double: Julia's promotion).  dst may alias
    any operand.  Every operation is rounded as written (-ffp-contract=off)."""
    vecs, scalars = list(vecs), list(scalars)
    h = _program(expr, dst.dtype, len(vecs), len(scalars), _real_mask(dst.dtype, vecs, scalars), _wide_mask(dst.dtype, scalars))
    hs = (C.c_void_p * max(len(vecs), 1))(*[v.handle for v in vecs])
    sc = (C.c_double * max(2 * len(scalars), 1))()
    for i, a in enumerate(scalars):
        a = complex(a)
        sc[2 * i], sc[2 * i + 1] = a.real, a.imag
    check(lib.jh_bcast_apply(h, dst.handle, hs, sc))
    return dst


class BExpr:
    """A lazy elementwise expression (the role of Base.Broadcast.Broadcasted): leaves are device vectors and scalars,
    nodes are C snippets over placeholder tokens; `assign_` numbers the distinct leaves and compiles the whole tree."""

    __array_ufunc__ = None

    def __init__(self, code: str, vecs: dict, scals: list):
        self.code, self.vecs, self.scals = code, vecs, scals      # vecs: token -> vector ; scals: [(token, value)]

    # -- construction helpers
    @staticmethod
    def of(x):
        if isinstance(x, BExpr):
            return x
        if isinstance(x, _DevVec):
            tok = f"@v{x.ptr:x}_{x.length()}@"
            return BExpr(tok, {tok: x}, [])
        if isinstance(x, (numbers.Number, np.generic)):
            tok = f"@s{next(_scalar_ids)}@"
            return BExpr(tok, {}, [(tok, x)])
        raise TypeError(f"cannot broadcast over {type(x).__name__}")

    @staticmethod
    def _join(fmt: str, *args):
        args = [BExpr.of(a) for a in args]
        vecs, scals = {}, []
        for a in args:
            vecs.update(a.vecs)
            scals += [s for s in a.scals if s[0] not in {t for t, _ in scals}]
        return BExpr(fmt.format(*[a.code for a in args]), vecs, scals)

    def __add__(self, o): return BExpr._join("({} + {})", self, o)
    def __radd__(self, o): return BExpr._join("({} + {})", o, self)
    def __sub__(self, o): return BExpr._join("({} - {})", self, o)
    def __rsub__(self, o): return BExpr._join("({} - {})", o, self)
    def __mul__(self, o): return BExpr._join("({} * {})", self, o)
    def __rmul__(self, o): return BExpr._join("({} * {})", o, self)
    def __truediv__(self, o): return BExpr._join("({} / {})", self, o)
    def __rtruediv__(self, o): return BExpr._join("({} / {})", o, self)
    def __neg__(self): return BExpr._join("(-{})", self)

    def __pow__(self, p):
        if isinstance(p, int) and 1 <= p <= 4:                    # x^2 is x*x in Julia too (literal_pow)
            out = self
            for _ in range(p - 1):
                out = BExpr._join("({} * {})", out, self)
            return out
        return BExpr._join("pow({}, {})", self, p)

    # -- materialisation
    def program(self):
        code = self.code
        vecs = []
        for k, (tok, v) in enumerate(self.vecs.items()):
            code = code.replace(tok, f"x{k}")
            vecs.append(v)
        scal = []
        for k, (tok, a) in enumerate(self.scals):
            code = code.replace(tok, f"s{k}")
            scal.append(a)
        return code, vecs, scal

    def materialize(self):
        """A fresh vector shaped like the first vector operand (x = a*u .+ ...; src/Jets.jl:889-897)."""
        if not self.vecs:
            raise ValueError("an expression without vector operands has no shape; assign_ it into a vector")
        first = next(iter(self.vecs.values()))
        return assign_(similar(first), self)


def lazy(x) -> BExpr:
    """Wrap a device vector (or scalar) as a leaf of a lazy broadcast expression."""
    return BExpr.of(x)


def assign_(dst: _DevVec, expr) -> _DevVec:
    """dst .= expr  -- one fused kernel for the whole tree."""
    code, vecs, scal = BExpr.of(expr).program()
    return broadcast_(dst, code, vecs, scal)


class _Funcs:
    """Elementwise functions for lazy expressions: Jets.bc.exp(x), Jets.bc.maximum(x, y), ..."""

    def __getattr__(self, name):
        unary = {"exp", "log", "sqrt", "sin", "cos", "tan", "tanh", "sinh", "cosh", "abs", "abs2", "conj", "real", "imag", "sign",
                 "floor", "ceil", "log2", "log10", "exp2", "atan", "asin", "acos", "erf"}
        binary = {"maximum": "fmax", "minimum": "fmin", "pow": "pow", "atan2": "atan2", "hypot": "hypot"}
        if name in unary:
            return lambda x: BExpr._join(name + "({})", x)
        if name in binary:
            return lambda x, y: BExpr._join(binary[name] + "({}, {})", x, y)
        raise AttributeError(name)


bc = _Funcs()
