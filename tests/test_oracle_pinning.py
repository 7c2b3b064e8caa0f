"""CPU: pin the oracle against the reference's own test identities.

The reference (pure Julia) cannot run in this image and stores no golden vectors; its tests are
identities between the operator API and a closed form written with plain arrays/matrices
(test/runtests.jl).  Each test below re-encodes one of those identities with numpy as the
independent closed form (the role Julia's `B*m`, `B'*d`, `norm(_x)` play there), plus the
literal-valued checks the reference does hold (test/runtests.jl:518-526).
"""
import math

import numpy as np
import pytest

from oracle import jets_oracle as jo

RNG = np.random.default_rng(20241016)
REAL = [np.float32, np.float64]
ALL = [np.float32, np.float64, np.complex64, np.complex128]


def rnd(dt, *shape):
    dt = np.dtype(dt)
    x = RNG.random(shape)
    if dt.kind == "c":
        x = x + 1j * RNG.random(shape)
    return np.asfortranarray(x.astype(dt))


def tol(dt):
    return 3.5e-4 if np.dtype(dt) in (np.dtype(np.float32), np.dtype(np.complex64)) else 1.5e-8   # Julia isapprox default: sqrt(eps)


# ------------------------------------------------------------------ layout (src/Jets.jl:739-750, 820-823)
def test_block_ranges_are_cumulative_one_based():
    # R = JetBSpace([JetSpace(Float64,2), JetSpace(Float64,2,2), JetSpace(Float64,2,3)])  (test/runtests.jl:513)
    start, stop = jo.bspace_indices([2, 4, 6])
    assert (start, stop) == ([1, 3, 7], [2, 6, 12])
    # config 4: 1024 blocks of 256^3 -> needs Int64
    start, stop = jo.bspace_indices([256 ** 3] * 1024)
    assert stop[-1] == 17_179_869_184 and start[-1] == stop[-1] - 256 ** 3 + 1
    start, stop = jo.bspace_indices([0, 3, 0])                       # empty blocks: start = stop + 1
    assert (start, stop) == ([1, 1, 4], [0, 3, 3])


def test_linear_index_lookup():
    lens = [2, 4, 6]
    assert jo.barr_locate(lens, 1) == (1, 1)
    assert jo.barr_locate(lens, 2) == (1, 2)
    assert jo.barr_locate(lens, 3) == (2, 1)
    assert jo.barr_locate(lens, 12) == (3, 6)
    with pytest.raises(IndexError):
        jo.barr_locate(lens, 13)


# ------------------------------------------------------------------ "block arrays" (test/runtests.jl:512-551)
def test_pi_fill_literals_and_norms():
    x = [np.ones(2), np.ones((2, 2), order="F"), np.ones((2, 3), order="F")]
    jo.barr_fill([x[0]], math.pi)                                     # setblock!(x,1,pi)
    jo.barr_fill([x[1]], 2 * math.pi)
    x[2][...] = 3 * math.pi * np.ones((2, 3))
    _x = jo.barr_convert(x)
    assert np.array_equal(_x, np.concatenate([np.full(2, math.pi), np.full(4, 2 * math.pi), np.full(6, 3 * math.pi)]))
    assert jo.barr_norm(x) == pytest.approx(np.linalg.norm(_x), rel=1e-15)            # :524
    assert jo.barr_norm(x, 0) == np.linalg.norm(_x, 0)                                 # :525
    assert jo.barr_norm(x, math.inf) == np.linalg.norm(_x, np.inf)                     # :526


@pytest.mark.parametrize("dt", ALL)
@pytest.mark.parametrize("p", [2, 1, 0, math.inf, -math.inf, 3])
def test_norm_matches_flat_norm(dt, p):
    x = [rnd(dt, 2), rnd(dt, 2, 2), rnd(dt, 2, 3), rnd(dt, 257)]
    flat = jo.barr_convert(x).astype(np.complex128)
    assert jo.barr_norm(x, p) == pytest.approx(np.linalg.norm(flat, p), rel=tol(dt))


@pytest.mark.parametrize("dt", REAL)
def test_extrema(dt):
    x = [rnd(dt, 2) - 0.5, rnd(dt, 2, 2) - 0.5, rnd(dt, 2, 3) - 0.5]                   # :528-540
    flat = jo.barr_convert(x)
    assert jo.barr_extrema(x) == (flat.min(), flat.max())


@pytest.mark.parametrize("dt", ALL)
def test_dot_conjugates_first_argument(dt):
    x = [rnd(dt, 2), rnd(dt, 2, 2), rnd(dt, 2, 3)]
    y = [rnd(dt, 2), rnd(dt, 2, 2), rnd(dt, 2, 3)]
    _x, _y = jo.barr_convert(x).astype(np.complex128), jo.barr_convert(y).astype(np.complex128)
    assert complex(jo.barr_dot(x, y)) == pytest.approx(np.vdot(_x, _y), rel=tol(dt))  # dot(x,x) ~ dot(_x,_x)  (:550)


# ------------------------------------------------------------------ "block arrays, broadcasting" (553-600)
@pytest.mark.parametrize("dt", ALL)
def test_broadcast_lincomb_is_left_to_right_in_eltype(dt):
    shapes = [(2,), (2, 2), (2, 3)]
    u, v, w = ([rnd(dt, *s) for s in shapes] for _ in range(3))
    a, b, c = 0.3, 0.7, 0.9
    x = jo.barr_lincomb([np.empty_like(t) for t in u], [a, b, c], [u, v, w])
    T = np.dtype(dt).type
    for i in range(3):                                                                 # :564-568 element by element
        expect = (T(a) * u[i] + T(b) * v[i]) + T(c) * w[i]
        if np.dtype(dt).kind != "c":
            assert np.array_equal(x[i], expect)                                        # same IEEE op sequence
        else:
            assert np.allclose(x[i], expect, rtol=tol(dt))


# ------------------------------------------------------------------ block operators
def dense(dt, nr, nc):
    return rnd(dt, nr, nc)


@pytest.mark.parametrize("dt", ALL)
def test_tall_and_skinny(dt):
    """A*m == [B1 m; B2 m; B3 m];  A'd == B1'd1 + B2'd2 + B3'd3   (test/runtests.jl:720-726)."""
    B = [dense(dt, 5, 5) for _ in range(3)]
    ops = [[jo.Block("dense", 5, 5, coeff=b)] for b in B]
    m = rnd(dt, 5)
    d = jo.block_df(ops, [np.zeros(5, dtype=dt) for _ in range(3)], [m])
    assert np.allclose(np.concatenate(d), np.concatenate([b @ m for b in B]), rtol=tol(dt))
    dd = [rnd(dt, 5) for _ in range(3)]
    mt = jo.block_df_adj(ops, [rnd(dt, 5)], dd)                                         # dirty output is zeroed (1042)
    assert np.allclose(mt[0], sum(b.conj().T @ x for b, x in zip(B, dd)), rtol=tol(dt))


@pytest.mark.parametrize("dt", ALL)
def test_tall_diagonal_is_exact(dt):
    """Diagonal blocks (JopFoo, test/runtests.jl:3-8): one multiply per element, ordered accumulate."""
    n, N = 33, 7
    g = [rnd(dt, n) for _ in range(N)]
    ops = [[jo.Block("diag", n, coeff=x)] for x in g]
    m = rnd(dt, n)
    d = jo.block_df(ops, [np.zeros(n, dtype=dt) for _ in range(N)], [m])
    dd = [rnd(dt, n) for _ in range(N)]
    mt = jo.block_df_adj(ops, [np.zeros(n, dtype=dt)], dd)
    if np.dtype(dt).kind != "c":
        for i in range(N):
            assert np.array_equal(d[i], g[i] * m)
        acc = np.zeros(n, dtype=dt)
        for i in range(N):                                                             # sequential, product rounded first
            acc = acc + g[i] * dd[i]
        assert np.array_equal(mt[0], acc)
    else:
        assert np.allclose(np.concatenate(d), np.concatenate([x * m for x in g]), rtol=tol(dt))
        assert np.allclose(mt[0], sum(np.conj(x) * y for x, y in zip(g, dd)), rtol=tol(dt))


@pytest.mark.parametrize("dt", ALL)
def test_short_and_fat(dt):
    """A*m == B1 m1 + B2 m2 + B3 m3;  A'd == [B1'd; B2'd; B3'd]   (test/runtests.jl:744-750)."""
    B = [dense(dt, 5, 5) for _ in range(3)]
    ops = [[jo.Block("dense", 5, 5, coeff=b) for b in B]]
    m = [rnd(dt, 5) for _ in range(3)]
    d = jo.block_df(ops, [np.zeros(5, dtype=dt)], m)
    assert np.allclose(d[0], sum(b @ x for b, x in zip(B, m)), rtol=tol(dt))
    dd = rnd(dt, 5)
    mt = jo.block_df_adj(ops, [rnd(dt, 5) for _ in range(3)], [dd])                     # nrow == 1: direct write (1051)
    assert np.allclose(np.concatenate(mt), np.concatenate([b.conj().T @ dd for b in B]), rtol=tol(dt))


@pytest.mark.parametrize("dt", ALL)
def test_mixed_3x4_with_zero_blocks_and_adjoint_block(dt):
    """The linear skeleton of test/runtests.jl:622-684: 3x4 blocks, Z22 and Z34 zero, A24 an adjoint."""
    n = 10
    B = {k: dense(dt, n, n) for k in ("11", "12", "13", "14", "21", "23", "24", "31", "32", "33")}
    blk = lambda k: jo.Block("dense", n, n, coeff=B[k])
    Z = lambda: jo.Block("zero", n, n)
    ops = [[blk("11"), blk("12"), blk("13"), blk("14")],
           [blk("21"), Z(), blk("23"), jo.Block("dense", n, n, coeff=B["24"], adjoint=True)],
           [blk("31"), blk("32"), blk("33"), Z()]]
    m = [rnd(dt, n) for _ in range(4)]
    d = jo.block_df(ops, [np.zeros(n, dtype=dt) for _ in range(3)], m)
    H = lambda a: a.conj().T
    assert np.allclose(d[0], B["11"] @ m[0] + B["12"] @ m[1] + B["13"] @ m[2] + B["14"] @ m[3], rtol=tol(dt))   # :664
    assert np.allclose(d[1], B["21"] @ m[0] + B["23"] @ m[2] + H(B["24"]) @ m[3], rtol=tol(dt))                 # :665
    assert np.allclose(d[2], B["31"] @ m[0] + B["32"] @ m[1] + B["33"] @ m[2], rtol=tol(dt))                    # :666
    dd = [rnd(dt, n) for _ in range(3)]
    mt = jo.block_df_adj(ops, [rnd(dt, n) for _ in range(4)], dd)                       # mul!(rand(domain(L)), L', dd)  (:684)
    assert np.allclose(mt[0], H(B["11"]) @ dd[0] + H(B["21"]) @ dd[1] + H(B["31"]) @ dd[2], rtol=tol(dt))
    assert np.allclose(mt[1], H(B["12"]) @ dd[0] + H(B["32"]) @ dd[2], rtol=tol(dt))
    assert np.allclose(mt[2], H(B["13"]) @ dd[0] + H(B["23"]) @ dd[1] + H(B["33"]) @ dd[2], rtol=tol(dt))
    assert np.allclose(mt[3], H(B["14"]) @ dd[0] + B["24"] @ dd[1], rtol=tol(dt))
    # reference quirk (src/Jets.jl:1024): forward accumulates into a dirty d when ncol > 1
    d0 = [rnd(dt, n) for _ in range(3)]
    d1 = jo.block_df(ops, [x.copy() for x in d0], m)
    assert np.allclose(d1[0], d0[0] + d[0], rtol=10 * tol(dt))


@pytest.mark.parametrize("dt", ALL)
def test_singleton(dt):
    """test/runtests.jl:704-710."""
    b = dense(dt, 5, 5)
    ops = [[jo.Block("dense", 5, 5, coeff=b)]]
    m, d = rnd(dt, 5), rnd(dt, 5)
    assert np.allclose(jo.block_df(ops, [np.zeros(5, dtype=dt)], [m])[0], b @ m, rtol=tol(dt))
    assert np.allclose(jo.block_df_adj(ops, [np.zeros(5, dtype=dt)], [d])[0], b.conj().T @ d, rtol=tol(dt))


def test_zero_block_is_skipped_not_zeroed():
    """src/Jets.jl:1022: in a one-column operator a zero block leaves its range block untouched."""
    n = 6
    ops = [[jo.Block("identity", n)], [jo.Block("zero", n, n)]]
    d = [np.full(n, 5.0), np.full(n, 7.0)]
    m = rnd(np.float64, n)
    jo.block_df(ops, d, [m])
    assert np.array_equal(d[0], m) and np.array_equal(d[1], np.full(n, 7.0))
    # but called directly, JopZeroBlock's df! zeroes (src/Jets.jl:942)
    assert np.array_equal(jo.child_mul(jo.Block("zero", n, n), np.full(n, 3.0), m), np.zeros(n))


@pytest.mark.parametrize("dt", ALL)
def test_scalar_times_operator_kernel(dt):
    """a*A applies d .= a*m / m .= conj(a)*d (src/Jets.jl:1159-1160; test/runtests.jl:789-795)."""
    a = 3.14 if np.dtype(dt).kind != "c" else 3.14 - 0.5j
    blk = jo.Block("scale", 10, scale=a)
    m = rnd(dt, 10)
    T = np.dtype(dt).type
    assert np.allclose(jo.child_mul(blk, np.empty(10, dtype=dt), m), T(a) * m, rtol=tol(dt))
    assert np.allclose(jo.child_mul_adj(blk, np.empty(10, dtype=dt), m), np.conj(T(a)) * m, rtol=tol(dt))


@pytest.mark.parametrize("dt", ALL)
def test_composite_normal_operator_order(dt):
    """(A' o A) m == A'(A m): right-to-left application (src/Jets.jl:530-534; test/runtests.jl:296-316)."""
    n, N = 12, 4
    g = [rnd(dt, n) for _ in range(N)]
    ops = [[jo.Block("diag", n, coeff=x)] for x in g]
    m = rnd(dt, n)
    y = jo.normal_df(ops, [rnd(dt, n)], [m])
    d = jo.block_df(ops, [np.zeros(n, dtype=dt) for _ in range(N)], [m])
    mt = jo.block_df_adj(ops, [np.zeros(n, dtype=dt)], d)
    assert np.array_equal(y[0], mt[0])
    assert np.allclose(y[0], sum(np.abs(x.astype(np.complex128)) ** 2 for x in g) * m, rtol=10 * tol(dt))


@pytest.mark.parametrize("dt", ALL)
def test_dot_product_test(dt):
    """test/runtests.jl:901-918 incl. masks and the complex case."""
    n = 10
    g = rnd(dt, n)
    ops = [[jo.Block("diag", n, coeff=g)]]
    m, d = rnd(dt, n), rnd(dt, n)
    lhs, rhs = jo.dot_product_test(ops, [m], [d])
    assert abs(lhs - rhs) <= tol(dt) * abs(rhs)
    mmask, dmask = np.ones(n, dtype=dt), np.ones(n, dtype=dt)
    mmask[0] = 0
    dmask[0] = 0
    lhs, rhs = jo.dot_product_test(ops, [m], [d], mmask=[mmask], dmask=[dmask])
    assert abs(lhs - rhs) <= tol(dt) * abs(rhs)
    if np.dtype(dt).kind == "c":
        assert isinstance(lhs, complex)


def test_config1_4x4_identity_float64():
    """BASELINE.json configs[0]: 4x4 JopBlock of identity JopLn on JetSpace(Float64,128), dot-product test."""
    n = 128
    ops = [[jo.Block("identity", n) for _ in range(4)] for _ in range(4)]
    m = [rnd(np.float64, n) for _ in range(4)]
    d = [rnd(np.float64, n) for _ in range(4)]
    lhs, rhs = jo.dot_product_test(ops, m, d)
    assert abs(lhs - rhs) / abs(lhs + rhs) < 1e-14
    out = jo.block_df(ops, [np.zeros(n) for _ in range(4)], m)
    s = ((m[0] + m[1]) + m[2]) + m[3]
    for i in range(4):
        assert np.array_equal(out[i], s)


# ------------------------------------------------------------------ counter-based generator
def _mix64(z):
    M = (1 << 64) - 1
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
    return z ^ (z >> 31)


def test_rng_known_answers_from_an_independent_big_int_implementation():
    G, M = 0x9E3779B97F4A7C15, (1 << 64) - 1
    for seed, stream, base in [(1, 0, 0), (2, 0, 5), (3, 7, 2 ** 33 + 11)]:
        key = _mix64((seed * G + stream) & M)
        # Float32 lane j: 24 bits of hash(j >> 1) -- the top 24 for an even lane, bits 39..16 for an odd one (round 4: one hash per two values)
        h32 = lambda j: _mix64((key + ((j >> 1) + 1) * G) & M)
        want32 = [(((h32(base + k) >> 16) & 0xFFFFFF) if (base + k) & 1 else (h32(base + k) >> 40)) * 2.0 ** -24 for k in range(64)]
        want64 = [((_mix64((key + (base + k + 1) * G) & M)) >> 11) * 2.0 ** -53 for k in range(64)]
        assert np.array_equal(jo.rng_u01(np.float32, seed, stream, base, 64), np.array(want32, dtype=np.float32))
        assert np.array_equal(jo.rng_u01(np.float64, seed, stream, base, 64), np.array(want64, dtype=np.float64))
    c = jo.rng_u01(np.complex64, 1, 0, 3, 8)                          # complex element k = lanes 2k (re), 2k+1 (im)
    f = jo.rng_u01(np.float32, 1, 0, 6, 16)
    assert np.array_equal(c.real, f[0::2]) and np.array_equal(c.imag, f[1::2])
    u = jo.rng_u01(np.float32, 9, 9, 0, 200000)
    assert 0.0 <= u.min() and u.max() < 1.0 and abs(u.mean() - 0.5) < 5e-3


# ------------------------------------------------------------------ nonlinear blocks (JopBar, test/runtests.jl:19-24)
def _bar_f(m):
    return m * m                                                                        # d .= m.^2


def _bar_J(mo, dm):
    return (2 * mo) * dm                                                                # dd .= 2 .* mo .* dm


@pytest.mark.parametrize("dt", ALL)
def test_nonlinear_singleton_tall_and_fat(dt):
    """The JopBar halves of test/runtests.jl:711-716 (singleton), 727-733 (tall), 751-757 (short-and-fat)."""
    n = 5
    exact = np.dtype(dt).kind != "c"
    eq = (lambda a, b: np.array_equal(a, b)) if exact else (lambda a, b: np.allclose(a, b, rtol=tol(dt)))
    # singleton
    m, d = rnd(dt, n), rnd(dt, n)
    ops = [[jo.Block("square", n, coeff=m)]]
    assert eq(jo.block_f(ops, [rnd(dt, n)], [m])[0], _bar_f(m))                          # F*m == G*m  (:713)
    assert eq(jo.block_df(ops, [rnd(dt, n)], [m])[0], _bar_J(m, m))                      # J*m         (:715)
    assert eq(jo.block_df_adj(ops, [rnd(dt, n)], [d])[0], np.conj(2 * m) * d)            # J'*d        (:716)
    # tall: F*m == [G1 m; G2 m; G3 m], J*m likewise, J'd == sum_i Ji'd_i  (:728-733)
    ops = [[jo.Block("square", n, coeff=m)] for _ in range(3)]
    out = jo.block_f(ops, [rnd(dt, n) for _ in range(3)], [m])
    assert all(eq(o, _bar_f(m)) for o in out)
    out = jo.block_df(ops, [rnd(dt, n) for _ in range(3)], [m])
    assert all(eq(o, _bar_J(m, m)) for o in out)
    dd = [rnd(dt, n) for _ in range(3)]
    mt = jo.block_df_adj(ops, [rnd(dt, n)], dd)[0]
    acc = np.zeros(n, dtype=dt)
    for x in dd:
        acc = acc + np.conj(2 * m) * x
    assert eq(mt, acc)
    # short-and-fat: F*m == sum_j Gj m_j (accumulated into d as found), J*m likewise, J'd == [Jj' d]  (:752-757)
    mb = [rnd(dt, n) for _ in range(3)]
    ops = [[jo.Block("square", n, coeff=x) for x in mb]]
    d0 = rnd(dt, n)
    acc = d0.copy()
    for x in mb:
        acc = acc + _bar_f(x)
    assert eq(jo.block_f(ops, [d0.copy()], mb)[0], acc)
    acc = np.zeros(n, dtype=dt)
    for x in mb:
        acc = acc + _bar_J(x, x)
    assert eq(jo.block_df(ops, [np.zeros(n, dtype=dt)], mb)[0], acc)
    out = jo.block_df_adj(ops, [rnd(dt, n) for _ in range(3)], [d])
    assert all(eq(o, np.conj(2 * x) * d) for o, x in zip(out, mb))


def test_nonlinear_forward_does_not_skip_zero_blocks():
    """src/Jets.jl:998-1004 has no `iszero` test (1022 does): a zero block's `d .= 0` (942) runs in f!."""
    n = 6
    m = rnd(np.float64, n)
    ops = [[jo.Block("square", n, coeff=m)], [jo.Block("zero", n, n)]]
    d0 = [rnd(np.float64, n), rnd(np.float64, n)]
    out = jo.block_f(ops, [x.copy() for x in d0], [m])
    assert np.array_equal(out[0], m * m) and np.array_equal(out[1], np.zeros(n))
    out = jo.block_df(ops, [x.copy() for x in d0], [m])
    assert np.array_equal(out[1], d0[1])                                                # the linear loop skips it
    wide = [[jo.Block("zero", n, n), jo.Block("square", n, coeff=np.zeros(n))]]
    out = jo.block_f(wide, [np.full(n, -0.0)], [np.zeros(n), np.zeros(n)])
    assert not np.signbit(out[0]).any()                                                 # -0.0 + 0.0 == +0.0


def test_nonlinear_jacobian_passes_the_dot_product_test():
    n = 40
    for dt in ALL:
        mo = [rnd(dt, n) for _ in range(2)]
        ops = [[jo.Block("square", n, coeff=mo[0]), jo.Block("diag", n, coeff=rnd(dt, n))],
               [jo.Block("zero", n, n), jo.Block("square", n, coeff=mo[1])]]
        lhs, rhs = jo.dot_product_test(ops, [rnd(dt, n) for _ in range(2)], [rnd(dt, n) for _ in range(2)])
        assert abs(lhs - rhs) <= 10 * tol(dt) * abs(lhs + rhs)


def test_multiple_linearizations_literal_values():
    """test/runtests.jl:203-211: J1*dm == 2 .* [1,2] .* dm == [2, 8]; J2*dm == 2 .* [3,4] .* dm == [6, 16]."""
    dm = np.array([1.0, 2.0])
    for mo, want in (([1.0, 2.0], [2.0, 8.0]), ([3.0, 4.0], [6.0, 16.0])):
        out = jo.block_df([[jo.Block("square", 2, coeff=np.array(mo))]], [np.zeros(2)], [dm])[0]
        assert out.tolist() == want
    assert jo.block_f([[jo.Block("square", 2, coeff=np.zeros(2))]], [np.zeros(2)], [np.array([3.0, 4.0])])[0].tolist() == [9.0, 16.0]


# ---- NaN in the ordered reductions: the reference folds with Julia's `max` / `min` (src/Jets.jl:835-838), which answer NaN when either
# argument is one; its extrema folds the blocks' (stdlib, NaN-propagating) extrema with `<` / `>` (:870-878), so a NaN in the first
# block is the answer and a later block that holds one drops out of the comparison altogether
@pytest.mark.parametrize("dt", REAL)
def test_nan_in_norm_inf_and_extrema_follows_julias_max_min(dt):
    nan = np.array(np.nan, dtype=dt)
    x = [np.array([1, -5, 2], dtype=dt), np.array([3, nan, -9], dtype=dt), np.array([4, 0.5], dtype=dt)]
    assert math.isnan(jo.barr_norm(x, math.inf)) and math.isnan(jo.barr_norm(x, -math.inf))
    assert jo.barr_extrema(x) == (-5.0, 4.0)                          # the middle block compares false both ways: its -9 and 3 are not seen
    y = [x[1], x[0], x[2]]
    mn, mx = jo.barr_extrema(y)
    assert math.isnan(mn) and math.isnan(mx)                          # NaN in the first block sticks
    assert jo.barr_norm([np.array([-0.0], dtype=dt)], -math.inf) == 0.0 and not np.signbit(jo.barr_norm([np.array([-0.0], dtype=dt)], -math.inf))
