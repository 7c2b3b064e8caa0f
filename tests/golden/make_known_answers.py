#!/usr/bin/env python3
"""Known answers for the block loops that do NOT come from the oracle (tests/golden/known_answers.npz).

    python tests/golden/make_known_answers.py          # rewrites known_answers.npz next to this script (pure Python, ~10 s)

The reference holds no vectors for JetBlock_df!/df'!, JetComposite or JetSum, Julia cannot run here, and the
other golden files are oracle output.  This script is an INDEPENDENT derivation of what the reference computes, written from
the reference's source lines and nothing else:

  * arithmetic: exact rationals (fractions.Fraction) + an IEEE-754 round-to-nearest-even written here from the standard
    (binary32 / binary64, signed zeros, subnormals) -- no numpy / C floating point takes part in any expected value;
  * loops: the statements of src/Jets.jl, one Python line per reference line:
      JetBlock_df!   1010-1032  rows outer, columns inner; zero blocks skipped (1022); ncol > 1: `_d .+= mul!(dtmp, op, _m)`
                                accumulates into d AS FOUND (1024); ncol == 1: `mul!(_d, op, _m)` overwrites (1026)
      JetBlock_df'!  1034-1057  columns outer, rows inner; nrow > 1: `_m .= 0` (1042) then `_m .+= mul!(mtmp, op', _d)` (1049):
                                product rounded, THEN added, rows in order; nrow == 1: direct write (1051), zero block => untouched
      JetComposite   522-540    right-to-left chain through zeros() temporaries, `d .= g(m)`
      JetSum         639-655    `d .= 0`, then `broadcast!(sgn, d, d, mul!(_d, op, m))` term by term; sign flattening 667-676
      chains         530-540    (round 6) composites of depth 3 to 7 around a tall operator -- A' o W o A, (W o A)' o (W o A), a * (A' o A),
                                M' o A' o (b W) o A o (a M) -- and sums whose terms are such chains: every stage into its own zeros()
    children: diagonal (test/runtests.jl:3-4  d .= diagonal .* m / m .= conj.(diagonal) .* d), identity, scalar a (1159-1160),
    JopZeroBlock (941-951), complex product = Julia's Complex *: (ar*br - ai*bi, ar*bi + ai*br), every operation rounded.

Two kinds of cases:
  exact-integer    every product and partial sum is an integer below 2^24: the answer is plain integer arithmetic, independent
                   of rounding and of summation order -- pins indexing, block layout, skip / accumulate / overwrite rules;
  order-revealing  e.g. products (2^24, 1, 1, ...): the reference's sequential sum stays at 2^24 (ties to even) while any other
                   order gives 2^24 + k; random cases in full softfloat emulation -- pin the ORDER and the ROUNDING SEQUENCE.

Both the CPU oracle (tests/test_known_answers.py) and the HIP path (tests/test_gpu_known_answers.py) must reproduce every
expected array bit for bit.
"""
from __future__ import annotations

import os
import struct
from fractions import Fraction

import numpy as np  # storage only (np.frombuffer / np.savez): no numpy arithmetic below

HERE = os.path.dirname(os.path.abspath(__file__))

# ------------------------------------------------------------------------------------------------ softfloat ---------
FMT = {"f32": (24, -126, 127, "<I", "<f4", 32), "f64": (53, -1022, 1023, "<Q", "<f8", 64)}


class F:
    """A finite IEEE value: sign bit + exact magnitude.  (Zero keeps its sign.)"""

    __slots__ = ("s", "m")

    def __init__(self, s: int, m: Fraction):
        self.s, self.m = s, m

    def __repr__(self):
        return f"{'-' if self.s else '+'}{float(self.m)!r}"


def rnd(s: int, mag: Fraction, fmt: str) -> F:
    """Round the exact value (-1)^s * mag to the nearest representable value, ties to even (IEEE 754-2008 section 4.3.1)."""
    p, emin, emax, *_ = FMT[fmt]
    if mag == 0:
        return F(s, Fraction(0))
    # exponent e with 2^e <= mag < 2^(e+1)
    e = mag.numerator.bit_length() - mag.denominator.bit_length()
    if Fraction(2) ** e > mag:
        e -= 1
    elif Fraction(2) ** (e + 1) <= mag:
        e += 1
    e = max(e, emin)                                   # subnormals share emin's quantum
    quantum = Fraction(2) ** (e - p + 1)
    q = mag / quantum
    n = q.numerator // q.denominator
    rem = q - n
    if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and (n & 1)):
        n += 1
    val = n * quantum
    if val >= Fraction(2) ** (emax + 1):
        raise OverflowError("known-answer inputs must stay finite")
    return F(s, val)


def fmul(a: F, b: F, fmt: str) -> F:
    return rnd(a.s ^ b.s, a.m * b.m, fmt)


def fadd(a: F, b: F, fmt: str) -> F:
    x = (-a.m if a.s else a.m) + (-b.m if b.s else b.m)
    if x == 0:                                          # IEEE 754 section 6.3: exact zero sum is +0 unless both are -0
        return F(1 if (a.s and b.s and a.m == 0 and b.m == 0) else 0, Fraction(0))
    return rnd(1 if x < 0 else 0, abs(x), fmt)


def fneg(a: F) -> F:
    return F(a.s ^ 1, a.m)


def to_bits(a: F, fmt: str) -> int:
    p, emin, emax, _, _, width = FMT[fmt]
    sign = a.s << (width - 1)
    if a.m == 0:
        return sign
    e = a.m.numerator.bit_length() - a.m.denominator.bit_length()
    if Fraction(2) ** e > a.m:
        e -= 1
    elif Fraction(2) ** (e + 1) <= a.m:
        e += 1
    if e < emin:                                        # subnormal
        frac = a.m / Fraction(2) ** (emin - p + 1)
        assert frac.denominator == 1
        return sign | int(frac)
    frac = a.m / Fraction(2) ** (e - p + 1)
    assert frac.denominator == 1 and (1 << (p - 1)) <= int(frac) < (1 << p), "value is not representable"
    return sign | ((e + emax) << (p - 1)) | (int(frac) - (1 << (p - 1)))


def from_bits(bits: int, fmt: str) -> F:
    p, emin, emax, _, _, width = FMT[fmt]
    s = bits >> (width - 1)
    ex = (bits >> (p - 1)) & ((1 << (width - p)) - 1)
    fr = bits & ((1 << (p - 1)) - 1)
    if ex == 0:
        return F(s, fr * Fraction(2) ** (emin - p + 1))
    assert ex != (1 << (width - p)) - 1, "no infinities / NaNs in the fixtures"
    return F(s, ((1 << (p - 1)) + fr) * Fraction(2) ** (ex - emax - p + 1))


def from_int(v: int) -> F:
    return F(1 if v < 0 else 0, Fraction(abs(v)))


# complex element = (re, im) pair of F; Julia's *(z::Complex, w::Complex)
def cmul(a, b, fmt):
    ar, ai = a
    br, bi = b
    re = fadd(fmul(ar, br, fmt), fneg(fmul(ai, bi, fmt)), fmt)
    im = fadd(fmul(ar, bi, fmt), fmul(ai, br, fmt), fmt)
    return (re, im)


def cadd(a, b, fmt):
    return (fadd(a[0], b[0], fmt), fadd(a[1], b[1], fmt))


def cconj(a):
    return (a[0], fneg(a[1]))


class Arith:
    """Element arithmetic of one eltype: real ("f32", "f64") or complex ("c32", "c64")."""

    def __init__(self, dtype: str):
        self.dtype = dtype
        self.cplx = dtype[0] == "c"
        self.fmt = {"f32": "f32", "f64": "f64", "c32": "f32", "c64": "f64"}[dtype]

    def mul(self, a, b):
        return cmul(a, b, self.fmt) if self.cplx else fmul(a, b, self.fmt)

    def add(self, a, b):
        return cadd(a, b, self.fmt) if self.cplx else fadd(a, b, self.fmt)

    def sub(self, a, b):                               # x - y (Julia's `-` broadcast): one rounding, as x + (-y)
        return self.add(a, (fneg(b[0]), fneg(b[1])) if self.cplx else fneg(b))

    def conj(self, a):
        return cconj(a) if self.cplx else a

    def zero(self):
        z = F(0, Fraction(0))
        return (z, F(0, Fraction(0))) if self.cplx else z

    def pack(self, vec) -> np.ndarray:
        _, _, _, code, npd, _ = FMT[self.fmt]
        flat = []
        for x in vec:
            flat.extend([x[0], x[1]] if self.cplx else [x])
        raw = b"".join(struct.pack(code, to_bits(x, self.fmt)) for x in flat)
        arr = np.frombuffer(raw, dtype=npd).copy()
        return arr.view({"f32": np.complex64, "f64": np.complex128}[self.fmt]) if self.cplx else arr


# ------------------------------------------------------------------------------------------------ child operators ---
class Child:
    """kind in {"zero", "identity", "scale", "diag"}; adjoint = the block is JopAdjoint(op)."""

    def __init__(self, kind, n, coeff=None, scale=None, adjoint=False):
        self.kind, self.n, self.coeff, self.scale, self.adjoint = kind, n, coeff, scale, adjoint

    def apply(self, ar: Arith, x, transposed: bool):
        """mul!(out, op, x) (transposed = False) or mul!(out, op', x): a fresh vector."""
        cj = self.adjoint != transposed                 # (op')' = op
        if self.kind == "identity":
            return list(x)                              # d .= m
        if self.kind == "scale":                        # d .= a*m ; m .= conj(a)*d   (1159-1160)
            a = ar.conj(self.scale) if cj else self.scale
            return [ar.mul(a, v) for v in x]
        if self.kind == "diag":                         # d .= diagonal .* m ; m .= conj.(diagonal) .* d
            return [ar.mul(ar.conj(c) if cj else c, v) for c, v in zip(self.coeff, x)]
        raise AssertionError("zero blocks are never applied by the linear loops (1022, 1047)")


def block_df(ar, ops, d_blocks, m_blocks):              # src/Jets.jl:1010-1032; d_blocks updated in place (lists of elements)
    nrow, ncol = len(ops), len(ops[0])
    for i in range(nrow):                               # 1015
        for j in range(ncol):                           # 1020
            if ops[i][j].kind != "zero":                # 1022
                dtmp = ops[i][j].apply(ar, m_blocks[j], False)
                if ncol > 1:
                    d_blocks[i] = [ar.add(x, y) for x, y in zip(d_blocks[i], dtmp)]    # _d .+= dtmp   (1024)
                else:
                    d_blocks[i] = dtmp                  # mul!(_d, op, _m)   (1026)
    return d_blocks


def block_df_adj(ar, ops, m_blocks, d_blocks):          # src/Jets.jl:1034-1057
    nrow, ncol = len(ops), len(ops[0])
    for j in range(ncol):                               # 1039
        if nrow > 1:
            m_blocks[j] = [ar.zero() for _ in m_blocks[j]]                              # _m .= 0   (1042)
        for i in range(nrow):                           # 1045
            if ops[i][j].kind != "zero":                # 1047
                mtmp = ops[i][j].apply(ar, d_blocks[i], True)
                if nrow > 1:
                    m_blocks[j] = [ar.add(x, y) for x, y in zip(m_blocks[j], mtmp)]    # _m .+= mtmp   (1049)
                else:
                    m_blocks[j] = mtmp                  # 1051
    return m_blocks


# ------------------------------------------------------------------------------------------------ inputs ------------
class Lcg:
    def __init__(self, seed):
        self.x = seed & ((1 << 64) - 1)

    def next(self):
        self.x = (self.x * 6364136223846793005 + 1442695040888963407) & ((1 << 64) - 1)
        return self.x >> 11

    def real(self, fmt, lo_exp=-3, hi_exp=3, signed=True) -> F:
        p = FMT[fmt][0]
        man = (1 << (p - 1)) | (self.next() & ((1 << (p - 1)) - 1))
        e = lo_exp + self.next() % (hi_exp - lo_exp + 1)
        return F(self.next() & 1 if signed else 0, man * Fraction(2) ** (e - p + 1))

    def elem(self, ar: Arith, **kw):
        return (self.real(ar.fmt, **kw), self.real(ar.fmt, **kw)) if ar.cplx else self.real(ar.fmt, **kw)

    def ints(self, n, lo, hi):
        return [lo + self.next() % (hi - lo + 1) for _ in range(n)]


def ivec(ar: Arith, values):
    return [(from_int(v), from_int(0)) if ar.cplx else from_int(v) for v in values]


CASES = {}


def store(case, name, ar, vec):
    CASES[f"{case}/{name}"] = ar.pack(vec)


def describe(case, ops, dtype, block_rows, block_cols):
    """The operator as data the tests rebuild it from: kinds / adjoint flags as small integer matrices (column-major flat)."""
    kinds = {"zero": 0, "identity": 1, "scale": 2, "diag": 3}
    nrow, ncol = len(ops), len(ops[0])
    CASES[f"{case}/shape"] = np.array([nrow, ncol], dtype=np.int64)
    CASES[f"{case}/kind"] = np.array([[kinds[ops[i][j].kind] for j in range(ncol)] for i in range(nrow)], dtype=np.int64)
    CASES[f"{case}/adjoint"] = np.array([[int(ops[i][j].adjoint) for j in range(ncol)] for i in range(nrow)], dtype=np.int64)
    CASES[f"{case}/row_len"] = np.array(block_rows, dtype=np.int64)
    CASES[f"{case}/col_len"] = np.array(block_cols, dtype=np.int64)
    CASES[f"{case}/dtype"] = np.array([["f32", "f64", "c32", "c64"].index(dtype)], dtype=np.int64)


def store_ops(case, ar, ops):
    for i, row in enumerate(ops):
        for j, op in enumerate(row):
            if op.kind == "diag":
                store(case, f"coeff_{i}_{j}", ar, op.coeff)
            if op.kind == "scale":
                store(case, f"scale_{i}_{j}", ar, [op.scale])


def run_linear_case(case, dtype, ops, m_blocks, d_found, m_found, d_for_adjoint=None):
    """forward from the dirty range vector `d_found`, adjoint (of d_for_adjoint, default: the forward's output) into the dirty
    domain vector `m_found`."""
    ar = Arith(dtype)
    nrow, ncol = len(ops), len(ops[0])
    describe(case, ops, dtype, [len(b) for b in d_found], [len(b) for b in m_blocks])
    store_ops(case, ar, ops)
    for j in range(ncol):
        store(case, f"m_{j}", ar, m_blocks[j])
        store(case, f"m_found_{j}", ar, m_found[j])
    for i in range(nrow):
        store(case, f"d_found_{i}", ar, d_found[i])
    d = block_df(ar, ops, [list(b) for b in d_found], m_blocks)
    for i in range(nrow):
        store(case, f"fwd_{i}", ar, d[i])
    din = d if d_for_adjoint is None else d_for_adjoint
    if d_for_adjoint is not None:
        for i in range(nrow):
            store(case, f"d_in_{i}", ar, din[i])
    mt = block_df_adj(ar, ops, [list(b) for b in m_found], din)
    for j in range(ncol):
        store(case, f"adj_{j}", ar, mt[j])
    return d, mt


def main():
    rng = Lcg(20260203)

    # ---- 1. exact integers, tall 7 x 1, n = 64 (vector kernels) and n = 7 (scalar kernels); Float32 and Float64 ------------
    for dtype in ("f32", "f64"):
        for n in (64, 7):
            ar = Arith(dtype)
            nrow = 7
            ops = [[Child("diag", n, coeff=ivec(ar, rng.ints(n, -9, 9)))] for _ in range(nrow)]
            m = [ivec(ar, rng.ints(n, -9, 9))]
            d_found = [ivec(ar, rng.ints(n, -99, 99)) for _ in range(nrow)]         # dirty: a tall forward must overwrite it
            m_found = [ivec(ar, rng.ints(n, -99, 99))]                              # dirty: the adjoint zeroes it first (1042)
            run_linear_case(f"int_tall_{dtype}_n{n}", dtype, ops, m, d_found, m_found)

    # ---- 2. order-revealing, tall 11 x 1, Float32: products (2^24, 1, 1, ...) and friends ---------------------------------
    ar = Arith("f32")
    n, nrow = 64, 11
    big = 1 << 24
    cols = []                                           # per element: the list of products p_i wanted, realised as a_i = p_i, d_i = 1
    for e in range(n):
        kind = e % 8
        if kind == 0:
            p = [big] + [1] * (nrow - 1)                # sequential: stays 2^24; any tree that adds the ones first: 2^24 + 10
        elif kind == 1:
            p = [1] * (nrow - 1) + [big]                # sequential: 10 + 2^24 = 2^24 + 10 (exact: even)
        elif kind == 2:
            p = [big, 1, -big] + [1] * (nrow - 3)       # sequential: (2^24 + 1 -> 2^24) - 2^24 = 0, then + 8
        elif kind == 3:
            p = [big + 2, 1, 1, 1] + [3] * (nrow - 4)   # ties to even going up and down along the way
        elif kind == 4:
            p = [1, big, 1, -big, 1] + [0] * (nrow - 5)
        elif kind == 5:
            p = [big, 3] + [1] * (nrow - 2)             # 2^24 + 3 -> 2^24 + 4 (tie to even, up)
        elif kind == 6:
            p = [-big, -1, -1] + [2] * (nrow - 3)
        else:
            p = rng.ints(nrow, -5, 5)
        cols.append(p)
    ops = [[Child("diag", n, coeff=ivec(ar, [cols[e][i] for e in range(n)]))] for i in range(nrow)]
    ones = [ivec(ar, [1] * n) for _ in range(nrow)]
    m = [ivec(ar, [1] * n)]
    describe("order_tall_f32", ops, "f32", [n] * nrow, [n])
    store_ops("order_tall_f32", ar, ops)
    store("order_tall_f32", "m_0", ar, m[0])
    store("order_tall_f32", "m_found_0", ar, ivec(ar, rng.ints(n, -99, 99)))
    for i in range(nrow):
        store("order_tall_f32", f"d_found_{i}", ar, ivec(ar, [0] * n))
        store("order_tall_f32", f"d_in_{i}", ar, ones[i])
    d = block_df(ar, ops, [ivec(ar, [0] * n) for _ in range(nrow)], m)
    for i in range(nrow):
        store("order_tall_f32", f"fwd_{i}", ar, d[i])
    mt = block_df_adj(ar, ops, [ivec(ar, [0] * n)], ones)
    store("order_tall_f32", "adj_0", ar, mt[0])
    # sanity of the construction itself: the sequential answer differs from the exact sum where it is meant to
    assert to_bits(mt[0][0], "f32") == to_bits(from_int(big), "f32") and to_bits(mt[0][1], "f32") == to_bits(from_int(big + 10), "f32")
    assert to_bits(mt[0][2], "f32") == to_bits(from_int(nrow - 3), "f32")

    # ---- 3. random values in full softfloat emulation: tall 13 x 1 (n = 32), all four eltypes ----------------------------
    for dtype in ("f32", "f64", "c32", "c64"):
        ar = Arith(dtype)
        n, nrow = 32, 13
        ops = [[Child("diag", n, coeff=[rng.elem(ar) for _ in range(n)])] for _ in range(nrow)]
        m = [[rng.elem(ar) for _ in range(n)]]
        d_found = [[rng.elem(ar) for _ in range(n)] for _ in range(nrow)]
        m_found = [[rng.elem(ar) for _ in range(n)]]
        d, mt = run_linear_case(f"rand_tall_{dtype}", dtype, ops, m, d_found, m_found)
        # JetComposite (A', A): mul!(zeros(range), A, m) then mul!(zeros(domain), A', .), `d .= g(m)`   (530-534): the same
        # two roundings per term as the unfused pair -- stored separately so that a fused kernel is pinned by its own name
        store(f"rand_tall_{dtype}", "normal_0", ar, mt[0])

    # ---- 4. 3 x 4 grid with zero blocks and mixed kinds (the shape of test/runtests.jl:622-695), ragged blocks ------------
    for dtype in ("f32", "c64"):
        ar = Arith(dtype)
        row_len, col_len = [12, 12, 12], [12, 12, 12, 12]          # elementwise children are square: all blocks 12
        def dg():
            return Child("diag", 12, coeff=[rng.elem(ar) for _ in range(12)])
        ops = [[dg(), Child("zero", 12), Child("identity", 12), Child("diag", 12, coeff=[rng.elem(ar) for _ in range(12)], adjoint=True)],
               [Child("zero", 12), Child("zero", 12), Child("zero", 12), Child("zero", 12)],          # a row of zero blocks: d_2 stays AS FOUND (1022)
               [Child("scale", 12, scale=rng.elem(ar)), dg(), Child("zero", 12), Child("scale", 12, scale=rng.elem(ar), adjoint=True)]]
        m = [[rng.elem(ar) for _ in range(12)] for _ in range(4)]
        d_found = [[rng.elem(ar) for _ in range(12)] for _ in range(3)]                             # dirty: accumulated into (1024)
        m_found = [[rng.elem(ar) for _ in range(12)] for _ in range(4)]                             # dirty: zeroed (1042), column 2 too
        run_linear_case(f"grid_{dtype}", dtype, ops, m, d_found, m_found)

    # ---- 5. wide 1 x 3 with a zero block: forward accumulates into d as found; adjoint (nrow == 1) writes DIRECTLY (1051) and
    #         leaves the zero block's column untouched (1047): its dirty content survives ------------------------------------
    ar = Arith("f64")
    ops = [[Child("diag", 10, coeff=[rng.elem(ar) for _ in range(10)]), Child("zero", 10), Child("identity", 10)]]
    m = [[rng.elem(ar) for _ in range(10)] for _ in range(3)]
    run_linear_case("wide_f64", "f64", ops, m, [[rng.elem(ar) for _ in range(10)]], [[rng.elem(ar) for _ in range(10)] for _ in range(3)])

    # ---- 6. JetSum: A1 - (A2 - A3) flattens to signs (+, -, +) (667-676); d .= 0 then term by term (639-646) -------------
    for dtype in ("f32", "f64"):
        ar = Arith(dtype)
        n, nrow = 16, 5
        terms = [[[Child("diag", n, coeff=[rng.elem(ar) for _ in range(n)])] for _ in range(nrow)] for _ in range(3)]
        m = [[rng.elem(ar) for _ in range(n)]]
        # element 0 of m is -0 times ... : make the first product of term 1 a negative zero to pin `0 + (-0) = +0` (d .= 0 first)
        m[0][0] = F(1, Fraction(0))
        din = [[rng.elem(ar) for _ in range(n)] for _ in range(nrow)]
        signs = ["+", "-", "+"]
        case = f"sum_{dtype}"
        CASES[f"{case}/shape"] = np.array([nrow, 1, 3], dtype=np.int64)
        CASES[f"{case}/dtype"] = np.array([["f32", "f64", "c32", "c64"].index(dtype)], dtype=np.int64)
        store(case, "m_0", ar, m[0])
        for i in range(nrow):
            store(case, f"d_in_{i}", ar, din[i])
        for t in range(3):
            for i in range(nrow):
                store(case, f"coeff_{t}_{i}", ar, terms[t][i][0].coeff)
        dsum = [[ar.zero() for _ in range(n)] for _ in range(nrow)]                                 # d .= 0   (640)
        for t in range(3):
            tmp = block_df(ar, terms[t], [[ar.zero()] * n for _ in range(nrow)], m)                 # mul!(_d, op, m)
            f = ar.add if signs[t] == "+" else ar.sub
            dsum = [[f(x, y) for x, y in zip(dsum[i], tmp[i])] for i in range(nrow)]                # broadcast!(sgn, d, d, _d)
        for i in range(nrow):
            store(case, f"fwd_{i}", ar, dsum[i])
        msum = [ar.zero() for _ in range(n)]                                                        # m .= 0   (649)
        for t in range(3):
            tmp = block_df_adj(ar, terms[t], [[ar.zero()] * n], din)[0]
            f = ar.add if signs[t] == "+" else ar.sub
            msum = [f(x, y) for x, y in zip(msum, tmp)]
        store(case, "adj_0", ar, msum)

    # ---- 7. tall 9 x 1 with rows of EVERY elementwise kind (zero, identity, scalar, diagonal, adjointed ones): the forward leaves
    #         the zero rows as found (1022), the adjoint skips them (1047); A'A goes through the zeros() temporary (530-534) -----
    for dtype in ("f32", "c64"):
        ar = Arith(dtype)
        n = 32
        def dg(adj=False):
            return Child("diag", n, coeff=[rng.elem(ar) for _ in range(n)], adjoint=adj)
        col = [dg(), Child("zero", n), Child("identity", n), Child("scale", n, scale=rng.elem(ar)), dg(adj=True), Child("zero", n), dg(),
               Child("scale", n, scale=rng.elem(ar), adjoint=True), Child("identity", n, adjoint=True)]
        ops = [[c] for c in col]
        nrow = len(ops)
        m = [[rng.elem(ar) for _ in range(n)]]
        m[0][3] = (F(1, Fraction(0)), F(1, Fraction(0))) if ar.cplx else F(1, Fraction(0))       # a negative zero through every kind
        d_found = [[rng.elem(ar) for _ in range(n)] for _ in range(nrow)]
        m_found = [[rng.elem(ar) for _ in range(n)]]
        case = f"mixed_tall_{dtype}"
        run_linear_case(case, dtype, ops, m, d_found, m_found)
        tmp = block_df(ar, ops, [[ar.zero() for _ in range(n)] for _ in range(nrow)], m)
        store(case, "normal_0", ar, block_df_adj(ar, ops, [[ar.zero() for _ in range(n)]], tmp)[0])

    # ---- 8. (round 3) LONG JetSums: 6, 8 and 11 terms.  The fused kernels take up to eight terms per launch and continue the
    #         left-to-right sum from what the output holds afterwards; the reference adds term by term into d .= 0 (639-646) and
    #         m .= 0 (648-655) whatever the count.  Signs as the nested differences of tests/known_answers.py: SUM_EXPRESSIONS
    #         flatten them (667-676).  Drawn AFTER every earlier case, so none of the older arrays changes. ----------------------
    for case, dtype, signs in (("sum6_f32", "f32", "+--+-+"), ("sum8_f32", "f32", "+--++-+-"), ("sum8_f64", "f64", "+--++-+-"),
                               ("sum11_f32", "f32", "+-+---+-+-+")):
        ar = Arith(dtype)
        n, nrow, nt = 16, 3, len(signs)
        terms = [[[Child("diag", n, coeff=[rng.elem(ar) for _ in range(n)])] for _ in range(nrow)] for _ in range(nt)]
        m = [[rng.elem(ar) for _ in range(n)]]
        m[0][0] = F(1, Fraction(0))                                                                 # -0: `0 + (-0) = +0`, then `(+0) - (-0) = +0` ...
        din = [[rng.elem(ar) for _ in range(n)] for _ in range(nrow)]
        CASES[f"{case}/shape"] = np.array([nrow, 1, nt], dtype=np.int64)
        CASES[f"{case}/dtype"] = np.array([["f32", "f64", "c32", "c64"].index(dtype)], dtype=np.int64)
        CASES[f"{case}/signs"] = np.array([1 if c == "+" else -1 for c in signs], dtype=np.int64)
        store(case, "m_0", ar, m[0])
        for i in range(nrow):
            store(case, f"d_in_{i}", ar, din[i])
        for t in range(nt):
            for i in range(nrow):
                store(case, f"coeff_{t}_{i}", ar, terms[t][i][0].coeff)
        dsum = [[ar.zero() for _ in range(n)] for _ in range(nrow)]                                 # d .= 0   (640)
        for t in range(nt):
            tmp = block_df(ar, terms[t], [[ar.zero()] * n for _ in range(nrow)], m)                 # mul!(_d, op, m)
            f = ar.add if signs[t] == "+" else ar.sub
            dsum = [[f(x, y) for x, y in zip(dsum[i], tmp[i])] for i in range(nrow)]                # broadcast!(sgn, d, d, _d)
        for i in range(nrow):
            store(case, f"fwd_{i}", ar, dsum[i])
        msum = [ar.zero() for _ in range(n)]                                                        # m .= 0   (649)
        for t in range(nt):
            tmp = block_df_adj(ar, terms[t], [[ar.zero()] * n], din)[0]
            f = ar.add if signs[t] == "+" else ar.sub
            msum = [f(x, y) for x, y in zip(msum, tmp)]
        store(case, "adj_0", ar, msum)

    # ---- 9. (round 6) CHAINS of depth 3 to 7 through a tall operator with rows of every kind, and sums whose terms are chains.
    #         JetComposite_df! (530-534): `dg = mapreduce(i -> (_m -> mul!(zeros(range(JopLn(ops[i]))), JopLn(ops[i]), _m)), o, 1:n); d .= dg(m)` --
    #         every stage into its own zeros(), right to left; JetComposite_df'! (536-540) the adjoints in the opposite order.  Stages besides the
    #         block operator: a diagonal over the block range (`W`: d .= diagonal .* m on the whole range vector, test/runtests.jl:3-4), a diagonal on
    #         the domain (`M`), a REAL scalar (`d .= a * m`, 1159: a::Real * z multiplies part by part).  n = 18 elements per block: rows of Float32
    #         are 72 bytes -- off the 16-byte grid.  Drawn AFTER every earlier case, so none of the older arrays changes. ----------------------------
    def rscale(ar, a, x):                                                       # d .= a * m for a Real a (Julia: Complex(a * re, a * im))
        return [(fmul(a, v[0], ar.fmt), fmul(a, v[1], ar.fmt)) if ar.cplx else fmul(a, v, ar.fmt) for v in x]

    def diag_stage(ar, coeff, x, conj):                                         # d .= diagonal .* m  /  m .= conj.(diagonal) .* d
        return [ar.mul(ar.conj(c) if conj else c, v) for c, v in zip(coeff, x)]

    def flat(blocks):
        return [v for b in blocks for v in b]

    def unflat(vec, nrow, n):
        return [vec[i * n:(i + 1) * n] for i in range(nrow)]

    def run_chain(ar, ops, nrow, n, stages, x):
        """x (a flat list: the domain vector or the whole range vector) through `stages` in application order; each stage writes a fresh vector."""
        cur = list(x)
        for st in stages:
            if st[0] == "A":                                                    # mul!(zeros(range(A)), A, cur)   (531)
                cur = flat(block_df(ar, ops, [[ar.zero() for _ in range(n)] for _ in range(nrow)], [cur]))
            elif st[0] == "At":                                                 # mul!(zeros(domain(A)), A', cur)   (537)
                cur = block_df_adj(ar, ops, [[ar.zero() for _ in range(n)]], unflat(cur, nrow, n))[0]
            elif st[0] == "D":
                cur = diag_stage(ar, st[1], cur, st[2])
            elif st[0] == "s":
                cur = rscale(ar, st[1], cur)
            else:
                raise AssertionError(st)
        return cur

    for dtype in ("f32", "f64", "c32", "c64"):
        ar = Arith(dtype)
        n = 18
        def dg(adj=False):
            return Child("diag", n, coeff=[rng.elem(ar) for _ in range(n)], adjoint=adj)
        rscal = (rng.real(ar.fmt), F(0, Fraction(0))) if ar.cplx else rng.real(ar.fmt)
        col = [dg(), Child("zero", n), Child("identity", n), Child("scale", n, scale=rscal), dg(adj=True), dg(), dg()]
        ops = [[c] for c in col]
        nrow = len(ops)
        case = f"chain_{dtype}"
        describe(case, ops, dtype, [n] * nrow, [n])
        store_ops(case, ar, ops)
        w = [[rng.elem(ar) for _ in range(nrow * n)] for _ in range(2)]          # two weight vectors over the whole range
        cdom = [rng.elem(ar) for _ in range(n)]                                 # a diagonal on the domain
        s0, s1 = from_int(3), F(1, Fraction(5, 8))                              # real scalars, exact in every element type
        m = [rng.elem(ar) for _ in range(n)]
        m[2] = (F(1, Fraction(0)), F(1, Fraction(0))) if ar.cplx else F(1, Fraction(0))       # a negative zero through every stage
        din = [rng.elem(ar) for _ in range(nrow * n)]
        store(case, "w_0", ar, w[0])
        store(case, "w_1", ar, w[1])
        store(case, "c_0", ar, cdom)
        CASES[f"{case}/scalars"] = np.array([3.0, -0.625], dtype=np.float64)
        store(case, "m_0", ar, m)
        store(case, "d_in", ar, din)
        A, At = ("A",), ("At",)
        W0, W0c, W1 = ("D", w[0], False), ("D", w[0], True), ("D", w[1], False)
        M, Mc = ("D", cdom, False), ("D", cdom, True)
        S0, S1 = ("s", s0), ("s", s1)
        chains = {
            "AtWA": [A, W0, At],                                                # A' o W o A                        (depth 3)
            "WAtWA": [A, W0, W0c, At],                                          # (W o A)' o (W o A)                (depth 4)
            "aAtA": [A, At, S0],                                                # a * (A' o A)                      (depth 3)
            "MtAtsWAsM": [M, S0, A, W1, S1, At, Mc],                            # M' o A' o (b W1) o A o (a M)      (depth 7)
        }
        for name, stages in chains.items():
            store(case, f"y_{name}", ar, run_chain(ar, ops, nrow, n, stages, m))
        store(case, "f_sWcA", ar, run_chain(ar, ops, nrow, n, [A, W0c, S1], m))                  # (b W0') o A: domain -> range
        store(case, "a_sWcA", ar, run_chain(ar, ops, nrow, n, [S1, W0, At], din))                # its adjoint: A' o W0 o b  (conj(b) == b)
        # JetSum over chains (639-646): d .= 0; d .= d + (A'WA) m; d .= d + (a I) m; d .= d - (A'A) m
        t1 = run_chain(ar, ops, nrow, n, [A, W0, At], m)
        t2 = rscale(ar, s1, list(m))                                            # (b * I) m: the composite (b, I) -- I into zeros(), then the scalar stage
        t3 = run_chain(ar, ops, nrow, n, [A, At], m)
        acc = [ar.zero() for _ in range(n)]
        acc = [ar.add(x, y) for x, y in zip(acc, t1)]
        acc = [ar.add(x, y) for x, y in zip(acc, t2)]
        acc = [ar.sub(x, y) for x, y in zip(acc, t3)]
        store(case, "y_sum", ar, acc)
        # ... and a sum on the range: W0 o A - (b W1) o A  (domain -> range), with its adjoint applied to d_in
        f1 = run_chain(ar, ops, nrow, n, [A, W0], m)
        f2 = run_chain(ar, ops, nrow, n, [A, W1, S1], m)
        accr = [ar.zero() for _ in range(nrow * n)]
        accr = [ar.add(x, y) for x, y in zip(accr, f1)]
        accr = [ar.sub(x, y) for x, y in zip(accr, f2)]
        store(case, "f_sum", ar, accr)
        g1 = run_chain(ar, ops, nrow, n, [W0c, At], din)
        g2 = run_chain(ar, ops, nrow, n, [S1, ("D", w[1], True), At], din)
        accd = [ar.zero() for _ in range(n)]
        accd = [ar.add(x, y) for x, y in zip(accd, g1)]
        accd = [ar.sub(x, y) for x, y in zip(accd, g2)]
        store(case, "a_sum", ar, accd)

    np.savez_compressed(os.path.join(HERE, "known_answers.npz"), **CASES)
    print(f"wrote {len(CASES)} arrays in {len({k.split('/')[0] for k in CASES})} cases to tests/golden/known_answers.npz")


if __name__ == "__main__":
    main()
