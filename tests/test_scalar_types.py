"""The TYPE of a scalar decides Julia's arithmetic (src/Jets.jl:1159-1160 `d .= a * m`, `m .= conj(a) * d`; 889-911 broadcast): a Real
multiplies a complex element part by part, a Complex takes the full product even with a zero imaginary part, and a Float64 scalar against
Float32 elements is promoted arithmetic rounded once on the store.  The oracle (CPU, here) and the HIP path (-m gpu) are both checked
against Julia's formulas spelled out with real numpy operations (tests/helpers.py: julia_scalar_term / julia_lincomb), on data that
holds signed zeros, infinities and values whose products round differently in the two precisions."""
import numpy as np
import pytest

from .helpers import assert_same_values as assert_bits_equal   # bit for bit, a NaN matching any NaN (payloads differ between x86 and CDNA)
from .helpers import julia_lincomb, julia_scalar_term

SPECIALS = [0.0, -0.0, np.inf, -np.inf, 1.0, -1.0, 1e-30, 3.0e38]


def _data(dt, n, seed):
    rng = np.random.default_rng(seed)
    dt = np.dtype(dt)
    if dt.kind == "c":
        x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(dt)
        k = 0
        for re in SPECIALS:                      # every pairing of special parts
            for im in SPECIALS:
                x[k] = complex(re, im)
                k += 1
    else:
        x = rng.standard_normal(n).astype(dt)
        x[:len(SPECIALS)] = SPECIALS
    return x


def _store(tr, ti, dt):
    R = np.float32 if np.dtype(dt) in (np.dtype(np.float32), np.dtype(np.complex64)) else np.float64
    with np.errstate(all="ignore"):
        if ti is None:
            return tr.astype(R)
        out = np.empty(tr.shape, dtype=dt)
        out.real, out.imag = tr.astype(R), ti.astype(R)
        return out


def scalars_for(dt):
    """scalars of every type class against elements of type dt"""
    real = [2.5, np.float32(0.1), np.float64(0.1), np.float64(3.14), 3]
    if np.dtype(dt).kind != "c":
        return real
    return real + [2 + 0j, complex(0.1, -0.0), np.complex64(0.1 + 0.3j), np.complex128(0.1 + 0.3j), np.complex128(3.14 + 0j), 0.7 - 1.3j]


@pytest.mark.parametrize("dt", [np.float32, np.float64, np.complex64, np.complex128])
def test_oracle_scale_blocks_and_lincomb_follow_the_scalars_type(oracle, dt):
    n = 200
    x, y = _data(dt, n, 5), _data(dt, n, 6)[::-1].copy()
    for a in scalars_for(dt):
        want = _store(*julia_scalar_term(a, x)[:2], dt)
        got = oracle.block_df([[oracle.Block("scale", n, scale=a)]], [np.zeros(n, dtype=dt)], [x])[0]
        assert_bits_equal(got, want, f"a * m, a = {a!r} ({type(a).__name__}), {np.dtype(dt)}")
        want = _store(*julia_scalar_term(a.conjugate() if isinstance(a, (complex, np.complexfloating)) else a, x)[:2], dt)
        got = oracle.block_df_adj([[oracle.Block("scale", n, scale=a)]], [np.zeros(n, dtype=dt)], [x])[0]
        assert_bits_equal(got, want, f"conj(a) * d, a = {a!r} ({type(a).__name__}), {np.dtype(dt)}")
        for b in scalars_for(dt)[::2]:
            got = oracle.barr_lincomb([np.empty(n, dtype=dt)], [a, b], [[x], [y]])[0]
            assert_bits_equal(got, julia_lincomb([a, b], [x, y]), f"a*x + b*y, a = {a!r}, b = {b!r}, {np.dtype(dt)}")
    # the type matters: a Complex 2 + 0im against a Real 2 on a signed zero and an infinity, Float64 0.1 against Float32(0.1)
    if np.dtype(dt).kind == "c":
        z = np.array([complex(-0.0, -1.0), complex(1.0, np.inf)], dtype=dt)
        as_real = oracle.block_df([[oracle.Block("scale", 2, scale=2.0)]], [np.zeros(2, dtype=dt)], [z])[0]
        as_cplx = oracle.block_df([[oracle.Block("scale", 2, scale=2 + 0j)]], [np.zeros(2, dtype=dt)], [z])[0]
        assert np.signbit(as_real[0].real) and not np.signbit(as_cplx[0].real)
        assert as_real[1].real == 2.0 and np.isnan(as_cplx[1].real)
    if np.dtype(dt) == np.dtype(np.float32):
        narrow = oracle.barr_lincomb([np.empty(n, dtype=dt)], [0.1], [[x]])[0]
        wide = oracle.barr_lincomb([np.empty(n, dtype=dt)], [np.float64(0.1)], [[x]])[0]
        assert (narrow != wide).sum() > n // 10, "Float64(0.1) * x rounded once differs from Float32(0.1) * x in many last bits"


@pytest.mark.gpu
@pytest.mark.parametrize("dt", [np.float32, np.float64, np.complex64, np.complex128])
def test_hip_scale_blocks_and_lincomb_follow_the_scalars_type(Jets, oracle, dt):
    """a * A through JH_OP_SCALE blocks -- next to a diagonal in a tall operator, in a 2 x 2 grid and as the stage of `a * G` (the fused
    kernels for scalars of the elements' precision, the per-block loop / the chain for wide and Complex-typed ones) -- and `a*u + b*v`
    through jh_lincomb_typed, against Julia's formulas and against the oracle given the same Python objects."""
    J = Jets
    n = 256
    spc = J.JetSpace(dt, n)
    x, y, g = _data(dt, n, 5), _data(dt, n, 6)[::-1].copy(), _data(dt, n, 7)
    dx, dy = J.from_numpy(x, spc), J.from_numpy(y, spc)
    G = J.JopDiagonal(J.from_numpy(g, spc))
    zero = np.zeros(n, dtype=dt)
    for a in scalars_for(dt):
        S = J.JopLn(dom=spc, rng=spc, df=J.constdiag_df, df_adj=J.constdiag_df_adj, s={"a": a})
        so, go = oracle.Block("scale", n, scale=a), oracle.Block("diag", n, coeff=g)
        what = f"a = {a!r} ({type(a).__name__}), {np.dtype(dt)}"
        # Julia's formula for the scalar rows themselves
        want_f = _store(*julia_scalar_term(a, x)[:2], dt)
        # tall [a*I ; G]
        A = J.blockop([[S], [G]])
        d = J.mul(A, dx).to_numpy()
        assert_bits_equal(d[:n], want_f, f"[a*I; G] forward row 1 vs Julia's formula, {what}")
        assert_bits_equal(d, np.concatenate(oracle.block_df([[so], [go]], [zero.copy(), zero.copy()], [x])), f"[a*I; G] forward vs oracle, {what}")
        dd = np.concatenate([x, y])
        mt = J.mul(A.H, J.from_numpy(dd, J.range(A))).to_numpy()
        assert_bits_equal(mt, oracle.block_df_adj([[so], [go]], [zero.copy()], [x, y])[0], f"[a*I; G]' vs oracle, {what}")
        J.close(A)
        # 2 x 2 grid [[a*I, G], [G, a*I]]
        B = J.blockop([[S, G], [G, S]])
        d2 = J.mul(B, J.from_numpy(dd, J.domain(B))).to_numpy()
        assert_bits_equal(d2, np.concatenate(oracle.block_df([[so, go], [go, so]], [zero.copy(), zero.copy()], [x, y])), f"grid forward vs oracle, {what}")
        m2 = J.mul(B.H, J.from_numpy(dd, J.range(B))).to_numpy()
        assert_bits_equal(m2, np.concatenate(oracle.block_df_adj([[so, go], [go, so]], [zero.copy(), zero.copy()], [x, y])), f"grid adjoint vs oracle, {what}")
        J.close(B)
        # a * T for a tall all-diagonal T: the composite (fused into the forward for a Real scalar of the elements' precision)
        T = J.blockop([[G], [G]])
        aT = a * T
        d3 = J.mul(aT, dx).to_numpy()
        with np.errstate(all="ignore"):
            inner = oracle.block_df([[go], [go]], [zero.copy(), zero.copy()], [x])
        want3 = np.concatenate([_store(*julia_scalar_term(a, blk)[:2], dt) for blk in inner])
        assert_bits_equal(d3, want3, f"(a * T) m vs Julia's formula, {what}")
        # (a * T)' d = T' (conj(a) d): the scalar stage rounds into a range-sized temporary (`m .= conj(a) * d`, 1160), then the ordered row sum
        ac = a.conjugate() if isinstance(a, (complex, np.complexfloating)) else a
        scaled = [_store(*julia_scalar_term(ac, blk)[:2], dt) for blk in (x, y)]
        m3 = J.mul(aT.H, J.from_numpy(dd, J.range(aT))).to_numpy()
        assert_bits_equal(m3, oracle.block_df_adj([[go], [go]], [zero.copy()], scaled)[0], f"(a * T)' d vs Julia's formula + the oracle's row sum, {what}")
        J.close(T)
        for b in scalars_for(dt)[::2]:
            got = (a * dx + b * dy).materialize().to_numpy()
            assert_bits_equal(got, julia_lincomb([a, b], [x, y]), f"a*x + b*y vs Julia's formula, b = {b!r}, {what}")
            assert_bits_equal(got, oracle.barr_lincomb([np.empty(n, dtype=dt)], [a, b], [[x], [y]])[0], f"a*x + b*y vs oracle, b = {b!r}, {what}")


def test_typed_broadcast_programs_compile_for_every_mix():
    """hiprtc cross-compiles without a GPU (jh_bcast_check_typed): wide and narrow, real and complex scalars next to complex and real
    operands in one expression -- the generated code's mixed-precision operators must resolve for every combination."""
    import ctypes as C
    import os

    lib = C.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "jets.jl_amd", "libjetship.so"))
    lib.jh_last_error.restype = C.c_char_p
    cases = [("s0*x0 + s1*x1", 0, 2, 0, 2, 1), ("s0*x0 + s1*x1", 0, 2, 0, 2, 3),
             ("s0*x0 + s1*x1", 2, 2, 0, 2, 1), ("s0*x0 + s1*x1", 2, 2, 4, 2, 1), ("s0*x0 + s1*x1", 2, 2, 12, 2, 3),
             ("(x0 / s0) - (s1 / x1) + conj(x0) * 2", 2, 2, 0, 2, 2), ("s0*x0 + x1", 2, 2, 2, 1, 1), ("s0*x0 + x1", 3, 2, 0, 1, 1),
             ("(s0 - x0) * (s1 + x1) / s0", 0, 2, 0, 2, 2)]
    for expr, dt, nvec, real_mask, nscal, wide_mask in cases:
        assert lib.jh_bcast_check_typed(expr.encode(), dt, nvec, real_mask, nscal, wide_mask) == 0, (expr, dt, real_mask, wide_mask, lib.jh_last_error())


@pytest.mark.gpu
@pytest.mark.parametrize("dt", [np.float32, np.complex64, np.float64])
def test_hip_compiled_broadcast_promotes_like_julia(Jets, dt):
    """`x .= a .* u .+ b .* v` and friends through the hiprtc-compiled broadcast with Float64 scalars against 32-bit elements: Float64
    arithmetic wherever such a scalar has entered, one rounding on the store -- the same bits as the typed lincomb and as Julia's formulas;
    and an expression with a division and a difference against numpy's own promotion (strong numpy scalars, real dtypes)."""
    J = Jets
    n = 1000
    spc = J.JetSpace(dt, n)
    x, y = _data(dt, n, 11), _data(dt, n, 12)[::-1].copy()
    dx, dy = J.from_numpy(x, spc), J.from_numpy(y, spc)
    out = J.zeros(spc)
    scal = scalars_for(dt)
    for a in scal:
        for b in scal[::2]:
            J.broadcast_(out, "s0*x0 + s1*x1", [dx, dy], [a, b])
            assert_bits_equal(out.to_numpy(), julia_lincomb([a, b], [x, y]), f"s0*x0 + s1*x1, a = {a!r} ({type(a).__name__}), b = {b!r}, {np.dtype(dt)}")
    if np.dtype(dt).kind != "c":
        xs, ys = np.abs(x[np.isfinite(x)][:512]) + dt(0.5), np.abs(y[np.isfinite(y)][:512]) + dt(0.25)
        xs, ys = xs[:min(xs.size, ys.size)], ys[:min(xs.size, ys.size)]
        spc2 = J.JetSpace(dt, xs.size)
        d2 = J.zeros(spc2)
        for a, b in ((np.float64(0.1), np.float32(0.3)), (0.1, np.float64(0.3)), (np.float64(1.7), np.float64(0.3)), (0.1, 0.3)):
            J.broadcast_(d2, "(x0 - s0) / (x1 + s1) - s0 * x1", [J.from_numpy(xs, spc2), J.from_numpy(ys, spc2)], [a, b])
            aa = a if isinstance(a, np.generic) else dt(a)      # a plain Python number is taken in the element type
            bb = b if isinstance(b, np.generic) else dt(b)
            with np.errstate(all="ignore"):
                want = ((xs - aa) / (ys + bb) - aa * ys).astype(dt)
            assert_bits_equal(d2.to_numpy(), want, f"(x0 - s0) / (x1 + s1) - s0 * x1, a = {a!r}, b = {b!r}, {np.dtype(dt)}")


@pytest.mark.gpu
@pytest.mark.parametrize("dt", [np.float32, np.complex64])
@pytest.mark.parametrize("n", [1 << 20, 1 << 22, 1 << 24])
def test_hip_wide_scalar_times_tall_operator_fused_at_every_launch_shape(Jets, oracle, dt, n):
    """`(a * A) m` and `(a * A)' d` for a Float64 scalar against 32-bit elements on the fused kernels' WIDE instantiations
    (jh_blockop_mul_scaled / _adj_scaled), at block sizes that select each of their launch shapes: the bits of Julia's promoted product
    rounded once, and NOT those of Float32(a) * x."""
    J = Jets
    if np.dtype(dt).itemsize * n * 3 > (1 << 29):
        pytest.skip("more than 512 MiB of coefficients for a shape check")
    spc = J.JetSpace(dt, n)
    g = [oracle.rng_u01(dt, 21, i, 0, n) for i in range(3)]
    x = oracle.rng_u01(dt, 22, 0, 0, n)
    T = J.blockop([[J.JopDiagonal(J.from_numpy(gi, spc))] for gi in g])
    a = np.float64(0.1)
    aT = a * T
    ops = [[oracle.Block("diag", n, coeff=gi)] for gi in g]
    inner = oracle.block_df(ops, [np.zeros(n, dtype=dt) for _ in range(3)], [x])
    want = np.concatenate([_store(*julia_scalar_term(a, blk)[:2], dt) for blk in inner])
    got = J.mul(aT, J.from_numpy(x, spc)).to_numpy()
    assert_bits_equal(got, want, f"(a * T) m, n = {n}, {np.dtype(dt)}")
    narrow = np.concatenate([_store(*julia_scalar_term(0.1, blk)[:2], dt) for blk in inner])
    assert (got.view(np.uint32) != narrow.view(np.uint32)).mean() > 0.1, "the wide product differs from Float32(0.1) * x in many last bits"
    scaled = [_store(*julia_scalar_term(a, blk)[:2], dt) for blk in inner]
    m = J.mul(aT.H, J.from_numpy(np.concatenate(inner), J.range(aT))).to_numpy()
    assert_bits_equal(m, oracle.block_df_adj(ops, [np.zeros(n, dtype=dt)], scaled)[0], f"(a * T)' d, n = {n}, {np.dtype(dt)}")
    J.close(T)


@pytest.mark.gpu
@pytest.mark.parametrize("dt", [np.float32, np.complex64])
@pytest.mark.parametrize("nterms", [3, 6, 11, 19])
def test_hip_jetsum_with_wide_scalars_stays_fused_with_the_chains_bits(Jets, oracle, monkeypatch, dt, nterms):
    """The reference's own docstring example `A = 1.0*A1 - 2.0*A2 + 3.0*A3` (src/Jets.jl:686, 703) on Float32 operators: numpy float64
    scalars are Julia's Float64 -- "wide" against 32-bit elements.  Round 4 sent such sums to the unfused chain ((5K + 1) range-sized
    streams); round 5's WIDE instantiations of the sum kernels keep them ONE pass ((K + 1) streams), every launch shape (4 / 8 / 16
    streams per launch, 19 = 16 + 3 continuing the left-to-right sum), with the bits (a) of the unfused chain on the device and (b) of
    Julia's formulas spelled out in numpy: product in the element type, scalar stage promoted and rounded once, signed add in the
    element type, terms in order; adjoint: scalar stage on d, ordered row sum per term, signed add."""
    import sys

    J = Jets
    blk = sys.modules[J.blockop.__module__]
    n, nrow = 3 * 4096 + 64, 5
    spc = J.JetSpace(dt, n)
    scal = [np.float64(1.0), np.float64(2.0), np.float64(3.14), np.float64(0.1), np.float32(0.7), 3, np.float64(-1.7)]       # wide, narrow and integer scalars mixed
    coeffs = [[oracle.rng_u01(dt, 31 + t, i, 0, n) for i in range(nrow)] for t in range(nterms)]
    ops = [J.blockop([[J.JopDiagonal(J.from_numpy(g, spc))] for g in coeffs[t]]) for t in range(nterms)]
    S, signs = None, []
    for t in range(nterms):
        a = scal[t % len(scal)]
        term = ops[t] if t == 4 else a * ops[t]                  # one bare operator among the scaled ones
        if S is None:
            S = term
            signs.append(1)
        elif t % 3 == 1:
            S = S - term
            signs.append(-1)
        else:
            S = S + term
            signs.append(1)
    x = oracle.rng_u01(dt, 41, 0, 0, n)
    x[:len(SPECIALS)] = np.asarray(SPECIALS, dtype=np.float32).astype(dt)
    din = [oracle.rng_u01(dt, 42, i, 0, n) for i in range(nrow)]

    fused_calls = []
    real_try = blk.try_fused_sum

    def counting(out, xx, o, s, transposed):
        r = real_try(out, xx, o, s, transposed)
        fused_calls.append(r is not None)
        return r

    monkeypatch.setattr(blk, "try_fused_sum", counting)
    d_fused = J.mul(S, J.from_numpy(x, spc)).to_numpy()
    m_fused = J.mul(S.H, J.from_numpy(np.concatenate(din), J.range(S))).to_numpy().ravel(order="F")
    assert fused_calls == [True, True], "a sum with Float64 scalars on 32-bit operators must take the fused route"
    monkeypatch.setattr(blk, "try_fused_sum", lambda *a, **k: None)
    d_chain = J.mul(S, J.from_numpy(x, spc)).to_numpy()
    m_chain = J.mul(S.H, J.from_numpy(np.concatenate(din), J.range(S))).to_numpy().ravel(order="F")
    monkeypatch.undo()
    assert_bits_equal(d_fused, d_chain, f"fused JetSum forward vs the unfused chain, {nterms} terms, {np.dtype(dt)}")
    assert_bits_equal(m_fused, m_chain, f"fused JetSum adjoint vs the unfused chain, {nterms} terms, {np.dtype(dt)}")

    # Julia's formulas, term by term
    zero = lambda: np.zeros(n, dtype=dt)
    want_d = [zero() for _ in range(nrow)]
    want_m = zero()
    with np.errstate(all="ignore"):
        for t in range(nterms):
            a = 1 if t == 4 else scal[t % len(scal)]
            obl = [[oracle.Block("diag", n, coeff=g)] for g in coeffs[t]]
            prod = oracle.block_df(obl, [zero() for _ in range(nrow)], [x])
            for i in range(nrow):
                term = prod[i] if t == 4 else _store(*julia_scalar_term(a, prod[i])[:2], dt)
                want_d[i] = (want_d[i] + term) if signs[t] > 0 else (want_d[i] - term)
            scaled = din if t == 4 else [_store(*julia_scalar_term(a, b)[:2], dt) for b in din]
            _m = oracle.block_df_adj(obl, [zero()], scaled)[0]
            want_m = (want_m + _m) if signs[t] > 0 else (want_m - _m)
    assert_bits_equal(d_fused, np.concatenate(want_d), f"fused JetSum forward vs Julia's formulas, {nterms} terms, {np.dtype(dt)}")
    assert_bits_equal(m_fused, want_m, f"fused JetSum adjoint vs Julia's formulas, {nterms} terms, {np.dtype(dt)}")
    for A in ops:
        J.close(A)
