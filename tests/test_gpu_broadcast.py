"""GPU: BlockArray broadcast for arbitrary elementwise expressions (src/Jets.jl:889-911) through the hiprtc-compiled fused
kernels (jh_bcast_*).  Re-encodes test/runtests.jl:553-600 (a*u .+ b*v .+ c*w, y .= x, x .* y) and widens to functions.

Bar: + - * / and sqrt on real eltypes are BIT-EXACT against numpy evaluating the same operations in the same order and
type (every operation rounded as written, no FMA); transcendental functions within 2e-6 (Float32) / 1e-14 (Float64)
relative; complex arithmetic within 1e-6 / 1e-14.
"""
import numpy as np
import pytest

from .helpers import DTYPES, assert_bits_equal, u01

pytestmark = pytest.mark.gpu


def _blockspace(Jets, dt, lens):
    return Jets.JetBSpace([Jets.JetSpace(dt, n) for n in lens])


@pytest.mark.parametrize("dt", [np.float32, np.float64])
@pytest.mark.parametrize("lens", [[2, 4, 6], [1024, 1024], [3, 5, 1000, 17], [1 << 20]])
def test_arithmetic_broadcast_is_bit_exact(Jets, oracle, dt, lens):
    """x = a*u .+ b*v .+ c*w (test/runtests.jl:561), then a longer tree with / and sqrt."""
    R = _blockspace(Jets, dt, lens)
    n = sum(lens)
    u, v, w = (Jets.rand(R, seed=s, stream=0) for s in (1, 2, 3))
    hu, hv, hw = (u01(oracle, dt, s, 0, n) for s in (1, 2, 3))
    T = np.dtype(dt).type
    a, b, c = T(0.37), T(-1.25), T(2.0)
    x = Jets.zeros(R)
    Jets.broadcast_(x, "s0*x0 + s1*x1 + s2*x2", [u, v, w], [a, b, c])
    assert isinstance(x, Jets.BlockArray)                                             # :562
    assert_bits_equal(x.to_numpy(), (a * hu + b * hv) + c * hw, "a*u .+ b*v .+ c*w")
    for i in range(len(lens)):                                                        # blockwise, like :564-568
        r = x.indices[i]
        assert_bits_equal(Jets.getblock(x, i).to_numpy().ravel(order="F"), ((a * hu + b * hv) + c * hw)[r.start:r.stop], f"block {i}")
    y = Jets.zeros(R)
    Jets.broadcast_(y, "(x0 - s0) / (x1 + s1) + sqrt(x2) * x0", [u, v, w], [a, c])
    assert_bits_equal(y.to_numpy(), (hu - a) / (hv + c) + np.sqrt(hw) * hu, "(u - a) / (v + c) + sqrt(w) * u")
    Jets.broadcast_(y, "s0", [], [3.14])                                              # x .= 3.14  (:583)
    assert_bits_equal(y.to_numpy(), np.full(n, T(3.14)), "x .= 3.14")


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_functions_within_tolerance(Jets, oracle, dt):
    n = 100_000
    spc = Jets.JetSpace(dt, n)
    u, v = Jets.rand(spc, seed=4, stream=0), Jets.rand(spc, seed=5, stream=0)
    hu, hv = u01(oracle, dt, 4, 0, n), u01(oracle, dt, 5, 0, n)
    tol = 2e-6 if np.dtype(dt) == np.float32 else 1e-14
    out = Jets.zeros(spc)
    cases = {"exp(-x0*x0) * cos(x1)": np.exp(-hu * hu) * np.cos(hv), "log(x0 + s0) + tanh(x1)": np.log(hu + 1) + np.tanh(hv),
             "fmax(x0, x1) - fmin(x0, x1)": np.maximum(hu, hv) - np.minimum(hu, hv), "pow(x0, s0) + sin(x1)": hu ** 1 + np.sin(hv),
             "abs(x0 - x1) + sign(x0 - x1)": np.abs(hu - hv) + np.sign(hu - hv), "abs2(x0) + conj(x1) + real(x0) + imag(x1)": hu * hu + hv + hu}
    for expr, want in cases.items():
        Jets.broadcast_(out, expr, [u, v], [1.0])
        np.testing.assert_allclose(out.to_numpy(), want.astype(dt), rtol=tol, atol=tol, err_msg=expr)


@pytest.mark.parametrize("dt", [np.complex64, np.complex128])
def test_complex_broadcast(Jets, oracle, dt):
    lens = [7, 64, 1001]
    R = _blockspace(Jets, dt, lens)
    n = sum(lens)
    u, v = Jets.rand(R, seed=6, stream=0), Jets.rand(R, seed=7, stream=0)
    hu, hv = u01(oracle, dt, 6, 0, n), u01(oracle, dt, 7, 0, n)
    tol = 1e-6 if np.dtype(dt) == np.complex64 else 1e-14
    out = Jets.zeros(R)
    a = 0.5 - 0.25j
    Jets.broadcast_(out, "s0*x0 + conj(x1)*x0 - x0/x1", [u, v], [a])
    np.testing.assert_allclose(out.to_numpy(), a * hu + np.conj(hv) * hu - hu / hv, rtol=tol, atol=tol)
    Jets.broadcast_(out, "exp(x0) * abs2(x1) + abs(x0) + 2*x1 - x0/3", [u, v])
    np.testing.assert_allclose(out.to_numpy(), np.exp(hu) * np.abs(hv) ** 2 + np.abs(hu) + 2 * hv - hu / 3, rtol=10 * tol, atol=10 * tol)
    # the arithmetic-only product keeps the explicit formula's bits (re*re - im*im, re*im + im*re), like jh_hadamard
    had = Jets.hadamard_(Jets.zeros(R), u, v)
    Jets.broadcast_(out, "x0*x1", [u, v])
    assert_bits_equal(out.to_numpy(), had.to_numpy(), "x0*x1 == jh_hadamard")


def test_aliasing_views_and_unaligned_blocks(Jets, oracle):
    """dst may be an operand; a block view at an odd element offset takes the one-element-per-lane kernel."""
    dt = np.float32
    R = _blockspace(Jets, dt, [3, 5, 1024, 7])
    n = 3 + 5 + 1024 + 7
    u, v = Jets.rand(R, seed=8, stream=0), Jets.rand(R, seed=9, stream=0)
    hu, hv = u01(oracle, dt, 8, 0, n), u01(oracle, dt, 9, 0, n)
    Jets.broadcast_(u, "x0*x1 + x0", [u, v])                                          # in place
    want = hu * hv + hu
    assert_bits_equal(u.to_numpy(), want, "u .= u .* v .+ u")
    b1, c1 = Jets.getblock(u, 1), Jets.getblock(v, 1)                                 # 5 elements at byte offset 12
    Jets.broadcast_(b1, "x0 - x1", [b1, c1])
    want[3:8] = want[3:8] - hv[3:8]
    assert_bits_equal(u.to_numpy(), want, "broadcast into an unaligned block view leaves its neighbours alone")
    b2 = Jets.getblock(u, 2)                                                          # 1024 elements at byte offset 32: vector kernel
    Jets.broadcast_(b2, "x0 * s0", [b2], [2.0])
    want[8:1032] = want[8:1032] * np.float32(2)
    assert_bits_equal(u.to_numpy(), want, "aligned block view")


def test_lazy_expression_tree(Jets, oracle):
    dt, n = np.float64, 5000
    spc = Jets.JetSpace(dt, n)
    u, v = Jets.rand(spc, seed=10, stream=0), Jets.rand(spc, seed=11, stream=0)
    hu, hv = u01(oracle, dt, 10, 0, n), u01(oracle, dt, 11, 0, n)
    L = Jets.lazy
    e = 2.0 * L(u) ** 2 - L(v) / (1.0 + L(u)) + Jets.bc.sqrt(L(v))
    code, vecs, scal = e.program()
    assert len(vecs) == 2 and code.count("x0") == 3                                   # u appears three times, one operand
    z = e.materialize()
    assert_bits_equal(z.to_numpy(), 2.0 * (hu * hu) - hv / (1.0 + hu) + np.sqrt(hv), "lazy tree")
    w = Jets.zeros(spc)
    Jets.assign_(w, Jets.bc.maximum(L(u), 0.5) * -L(v))
    assert_bits_equal(w.to_numpy(), np.maximum(hu, 0.5) * -hv, "maximum / unary minus")
    Jets.assign_(w, 7)                                                                # scalar only
    assert (w.to_numpy() == 7).all()
    programs = len(Jets.broadcast._programs)
    Jets.assign_(w, 2.0 * L(v) ** 2 - L(u) / (1.0 + L(v)) + Jets.bc.sqrt(L(u)))      # same tree, other operands/values: cached program
    assert len(Jets.broadcast._programs) == programs


def test_errors_are_reported_not_crashes(Jets):
    spc = Jets.JetSpace(np.float32, 64)
    u, out = Jets.rand(spc), Jets.zeros(spc)
    with pytest.raises(Jets.JetsHipError) as ei:
        Jets.broadcast_(out, "x0 +* nonsense(", [u])
    assert "x0 +* nonsense(" in str(ei.value) and "error" in str(ei.value)
    with pytest.raises(Jets.JetsHipError, match="DimensionMismatch"):
        Jets.broadcast_(out, "x0", [Jets.rand(Jets.JetSpace(np.float32, 63))])
    with pytest.raises(Jets.JetsHipError, match="dtype"):
        Jets.broadcast_(out, "x0", [Jets.rand(Jets.JetSpace(np.float64, 64))])
    with pytest.raises(Jets.JetsHipError):
        Jets.broadcast_(out, "x0", [u] * 9)                                           # more than 8 vector operands
    Jets.broadcast_(out, "x0", [u])                                                   # the context is still healthy
    assert_bits_equal(out.to_numpy(), u.to_numpy(), "copy")


@pytest.mark.parametrize("dt", [np.float32, np.float64, np.complex64, np.complex128])
def test_many_broadcasts_in_one_launch_equal_item_by_item(Jets, dt):
    """jh_bcast_apply_many: items that share the program, the length and 16-byte alignment run as ONE launch over
    (packs, items) with device tables -- same bits as one launch per item; anything else (an operand that is another item's
    destination, mixed programs, odd lengths) keeps the item-by-item order."""
    n, count = 1000, 37
    R = Jets.JetBSpace([Jets.JetSpace(dt, n)] * count)
    x, y = Jets.rand(R, seed=11, stream=0), Jets.rand(R, seed=12, stream=0)
    expr = "s0*x0*x1 + s1"
    one, many = Jets.zeros(R), Jets.zeros(R)
    for k in range(count):
        Jets.broadcast_(one.arrays[k], expr, [x.arrays[k], y.arrays[k]], [0.5 + k, -0.25 * k])
    Jets.broadcast_many_((many.arrays[k], expr, [x.arrays[k], y.arrays[k]], [0.5 + k, -0.25 * k]) for k in range(count))
    assert many.to_numpy().tobytes() == one.to_numpy().tobytes()
    # in place on its own destination: still one launch, still the same bits
    z1, z2 = Jets.rand(R, seed=13, stream=0), Jets.rand(R, seed=13, stream=0)
    for k in range(count):
        Jets.broadcast_(z1.arrays[k], "x0 + x1*x1", [z1.arrays[k], x.arrays[k]])
    Jets.broadcast_many_((z2.arrays[k], "x0 + x1*x1", [z2.arrays[k], x.arrays[k]], []) for k in range(count))
    assert z2.to_numpy().tobytes() == z1.to_numpy().tobytes()
    # a chain (item k reads item k-1's destination) must keep its order
    c1, c2 = Jets.rand(R, seed=14, stream=0), Jets.rand(R, seed=14, stream=0)
    for k in range(1, count):
        Jets.broadcast_(c1.arrays[k], "x0 + x1", [c1.arrays[k - 1], x.arrays[k]])
    Jets.broadcast_many_((c2.arrays[k], "x0 + x1", [c2.arrays[k - 1], x.arrays[k]], []) for k in range(1, count))
    assert c2.to_numpy().tobytes() == c1.to_numpy().tobytes()
    # mixed programs and lengths fall back as a whole
    S = Jets.JetBSpace([Jets.JetSpace(dt, 7 + k) for k in range(6)])
    u, v1, v2 = Jets.rand(S, seed=15, stream=0), Jets.zeros(S), Jets.zeros(S)
    for k in range(6):
        Jets.broadcast_(v1.arrays[k], "x0*x0" if k % 2 else "x0 + x0", [u.arrays[k]])
    Jets.broadcast_many_((v2.arrays[k], "x0*x0" if k % 2 else "x0 + x0", [u.arrays[k]], []) for k in range(6))
    assert v2.to_numpy().tobytes() == v1.to_numpy().tobytes()


@pytest.mark.parametrize("cdt,rdt", [(np.complex64, np.float32), (np.complex128, np.float64)])
def test_real_operands_in_a_complex_broadcast(Jets, cdt, rdt):
    """Mixed element types (src/Jets.jl:899-904 pairs blocks whatever their eltypes): a REAL mask / weight on a complex vector.
    Real (x) complex is Julia's rule -- a*(x + iy) = (a*x) + i(a*y), a + z adds to the real part -- which is numpy's too, so the
    fused kernel must give numpy's bits; 16-byte path, the one-element path (a view at an odd offset) and the batched entry."""
    J = Jets
    n = 4099
    rs = np.random.RandomState(7)
    hz = (rs.standard_normal(n) + 1j * rs.standard_normal(n)).astype(cdt)
    hw = (rs.standard_normal(n) - 1j * rs.standard_normal(n)).astype(cdt)
    ha = rs.standard_normal(n).astype(rdt)
    hz[5] = cdt(complex(-0.0, -3.0))                                   # signed zeros through the mixed product
    ha[5] = rdt(0.0)
    z, w, a = J.from_numpy(hz), J.from_numpy(hw), J.from_numpy(ha)
    out = J.zeros(J.JetSpace(cdt, n))
    def rc_mul(r, c):                                                   # Julia: *(x::Real, z::Complex) = Complex(x*real(z), x*imag(z))
        o = np.empty(c.shape, cdt)                                     # (numpy promotes the real to complex first: other signed zeros)
        o.real, o.imag = r * c.real, r * c.imag
        return o

    J.broadcast_(out, "x0*x1", [a, z])
    assert out.to_numpy().tobytes() == rc_mul(ha, hz).tobytes()
    assert np.signbit(out.to_numpy()[5].real) and np.signbit(out.to_numpy()[5].imag)   # 0 * (-0 - 3i) = -0 - 0i
    J.broadcast_(out, "x1*x0 + x2", [a, z, w])                         # operand order free: bit 0 is the real one
    assert out.to_numpy().tobytes() == (rc_mul(ha, hz) + hw).astype(cdt).tobytes()
    J.broadcast_(out, "(x0 + x1)*s0 - x0", [a, z], [2.0 - 0.5j])
    want = ((ha + hz) * cdt(2.0 - 0.5j) - ha).astype(cdt)
    np.testing.assert_allclose(out.to_numpy(), want, rtol=3e-6 if cdt == np.complex64 else 1e-14)
    # a block vector with a real mask of the same block structure; the one-element kernel through views at odd offsets
    R = J.JetBSpace([J.JetSpace(cdt, 33), J.JetSpace(cdt, 1000)])
    Rm = J.JetBSpace([J.JetSpace(rdt, 33), J.JetSpace(rdt, 1000)])
    bz, bm = J.from_numpy(hz[:1033], R), J.from_numpy(ha[:1033], Rm)
    bo = J.zeros(R)
    J.broadcast_(J.getblock(bo, 1), "x0*x1", [J.getblock(bm, 1), J.getblock(bz, 1)])   # block 1 starts at element 33: odd alignment
    assert bo.to_numpy()[33:].tobytes() == rc_mul(ha[33:1033], hz[33:1033]).tobytes()
    assert not bo.to_numpy()[:33].any()
    # wrong mixes are refused
    with pytest.raises(J.JetsHipError):
        J.broadcast_(J.zeros(J.JetSpace(rdt, n)), "x0*x1", [a, z])     # a complex operand in a real program
    other = J.from_numpy(ha.astype(np.float64 if rdt == np.float32 else np.float32))
    with pytest.raises(J.JetsHipError):
        J.broadcast_(out, "x0*x1", [other, z])                         # precision mismatch


def test_dot_product_test_with_real_masks_on_a_complex_operator(Jets):
    """dot_product_test(A, m, d; mmask, dmask) (src/Jets.jl:1211-1226) with REAL masks on complex vectors (test/runtests.jl:915-917
    uses complex operators; a real mute mask is the common case in practice)."""
    J = Jets
    dt = np.complex64
    spc = J.JetSpace(dt, 2048)
    A = J.blockop([[J.JopDiagonal(J.rand(spc, seed=31, stream=i))] for i in range(4)])
    m, d = J.rand(J.domain(A), seed=32), J.rand(J.range(A), seed=33)
    mmask = J.from_numpy((np.arange(2048) % 3 != 0).astype(np.float32))
    dmask = J.from_numpy((np.arange(4 * 2048) % 5 != 0).astype(np.float32), J.JetBSpace([J.JetSpace(np.float32, 2048)] * 4))
    lhs, rhs = J.dot_product_test(A, m, d, mmask=mmask, dmask=dmask)
    assert abs(lhs - rhs) <= 1e-5 * abs(lhs + rhs)
