"""GPU: the reference's own operator test sets, re-encoded through the host mirror on device arrays.

Each test follows one `@testset` of /root/reference/test/runtests.jl with the same toy operators written as
closures over HIP-backed arrays (JopFoo -> JopDiagonal, JopBaz -> JopDense, JopBar -> the x^2 closure below),
seeded inputs instead of unseeded `rand`, and the same closed forms in numpy.  `approx` is Julia's `isapprox`
tightened from rtol = sqrt(eps) to 1e-12 (Float64).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

F64 = np.float64
RNG = np.random.default_rng(1234)


def approx(x, y, rtol=1e-12):
    x = x.to_numpy() if hasattr(x, "to_numpy") else np.asarray(x)
    y = y.to_numpy() if hasattr(y, "to_numpy") else np.asarray(y)
    x, y = x.ravel(order="F"), y.ravel(order="F")
    return np.linalg.norm(x - y) <= rtol * max(np.linalg.norm(x), np.linalg.norm(y), 1e-300)


@pytest.fixture(scope="module")
def T(Jets):
    """The reference's toy operators (test/runtests.jl:3-56) over device arrays."""
    J = Jets

    class Toys:
        @staticmethod
        def dev(a):
            return J.from_numpy(np.asfortranarray(a))

        @staticmethod
        def JopFoo(diag):                                    # test/runtests.jl:3-8
            return J.JopDiagonal(Toys.dev(diag))

        @staticmethod
        def JopBaz(A):                                       # :27-33
            return J.JopDense(Toys.dev(A))

        @staticmethod
        def JopBar(n):                                       # :20-25   f: d .= m.^2 ; df: dd .= 2 .* mo .* dm
            spc = J.JetSpace(F64, n)
            return J.JopNl(f=Toys.bar_f, df=Toys.bar_df, dom=spc, rng=spc)

        @staticmethod
        def bar_f(d, m, **kw):
            return J.hadamard_(d, m, m)

        @staticmethod
        def bar_df(dd, dm, mo=None, **kw):
            J.hadamard_(dd, mo, dm)
            return J.lincomb_(dd, [2.0], [dd])

    return Toys


def test_composition_linear(Jets, T):
    """test/runtests.jl:296-326."""
    B1, B2, B3, B4 = (RNG.random((10, 10)) for _ in range(4))
    A1, A2, A3, A4 = map(T.JopBaz, (B1, B2, B3, B4))
    A21, A321, A4321 = A2 @ A1, A3 @ A2 @ A1, A4 @ A3 @ A2 @ A1
    hm = RNG.random(10)
    m = T.dev(hm)
    assert approx(A21 * m, B2 @ (B1 @ hm))
    assert approx(A321 * m, B3 @ (B2 @ (B1 @ hm)))
    d = A4321 * m
    hd = B4 @ (B3 @ (B2 @ (B1 @ hm)))
    assert approx(d, hd)
    assert len(Jets.state(A4321)["ops"]) == 4
    assert approx(A21.H * d, B1.T @ (B2.T @ hd))
    assert approx(A321.H * d, B1.T @ (B2.T @ (B3.T @ hd)))
    assert approx(A4321.H * d, B1.T @ (B2.T @ (B3.T @ (B4.T @ hd))))
    assert Jets.domain(A4321) == Jets.JetSpace(F64, 10) and Jets.eltype(A4321) == np.dtype(F64)
    C4321 = A4 @ A3 @ A21.H
    assert approx(C4321 * m, B4 @ (B3 @ (B1.T @ (B2.T @ hm))))


def test_composition_nonlinear_and_jacobians(Jets, T):
    """test/runtests.jl:358-390."""
    F1, F2, F3, F4 = (T.JopBar(10) for _ in range(4))
    F21, F321, F4321 = F2 @ F1, F3 @ F2 @ F1, F4 @ F3 @ F2 @ F1
    hm = RNG.random(10)
    m = T.dev(hm)
    assert approx(F21 * m, (hm ** 2) ** 2)
    assert approx(F321 * m, hm ** 8)
    assert approx(F4321 * m, hm ** 16)
    m1 = Jets.ones(Jets.JetSpace(F64, 10))
    J1 = Jets.jacobian_(F1, m1)
    J21 = Jets.jacobian_(F2, F1 * m1) @ J1
    L21 = Jets.jacobian_(F21, m1)
    L4321 = Jets.jacobian_(F4321, m1)
    dm = Jets.ones(Jets.JetSpace(F64, 10))
    assert approx(J21 * dm, L21 * dm)
    assert approx(L4321 * dm, 16 * np.ones(10))             # d/dm m^16 at m = 1
    dd = L4321 * dm
    assert approx(L4321.H * dd, 16 * 16 * np.ones(10))


def test_composition_linear_nonlinear_adjoints(Jets, T):
    """test/runtests.jl:392-423."""
    B2, g4 = RNG.random((10, 10)), RNG.random(10)
    A2, A4 = T.JopBaz(B2), T.JopFoo(g4)
    F1, F3 = T.JopBar(10), T.JopBar(10)
    F4321 = A4 @ F3 @ A2.H @ F1
    hm = RNG.random(10)
    m = T.dev(hm)
    assert approx(F4321 * m, g4 * (B2.T @ hm ** 2) ** 2)
    L = Jets.jacobian_(F4321, m)
    hdm = RNG.random(10)
    inner = B2.T @ hm ** 2
    expect = g4 * (2 * inner * (B2.T @ (2 * hm * hdm)))
    assert approx(L * T.dev(hdm), expect)


def test_sum_linear_and_with_compositions(Jets, T):
    """test/runtests.jl:453-488."""
    B1, B2, B3 = (RNG.random((10, 10)) for _ in range(3))
    A1, A2, A3 = map(T.JopBaz, (B1, B2, B3))
    hm, hd = RNG.random(10), RNG.random(10)
    m, d = T.dev(hm), T.dev(hd)
    assert approx((A1 + A2) * m, B1 @ hm + B2 @ hm)
    assert approx((A1 + A2 - A3) * m, B1 @ hm + B2 @ hm - B3 @ hm)
    A12 = A1 + A2
    A123 = A12 + A3
    assert approx(A123 * m, B1 @ hm + B2 @ hm + B3 @ hm)
    assert approx((A123 - A12) * m, B3 @ hm, rtol=1e-11)
    assert approx(A123.H * d, B1.T @ hd + B2.T @ hd + B3.T @ hd)
    a1, a2, a3 = 0.3, 0.7, 0.2
    S = a1 * A1 + a2 * A2 - a3 * A3
    assert approx(S * m, a1 * (B1 @ hm) + a2 * (B2 @ hm) - a3 * (B3 @ hm))
    S2 = (a1 * A1 + a2 * A2) + a3 * A3
    assert approx(S2.H * d, a1 * (B1.T @ hd) + a2 * (B2.T @ hd) + a3 * (B3.T @ hd))


def test_sum_linear_plus_nonlinear(Jets, T):
    """test/runtests.jl:500-510."""
    B1 = RNG.random((10, 10))
    A1, F2 = T.JopBaz(B1), T.JopBar(10)
    F12 = A1 + F2
    hm = RNG.random(10)
    m = T.dev(hm)
    assert approx(F12 * m, B1 @ hm + hm ** 2)
    J12 = Jets.jacobian(F12, m)
    assert approx(J12 * m, B1 @ hm + 2 * hm * hm)


def test_block_array_reshaped_and_blockop_vector_form(Jets, T):
    """test/runtests.jl:611-618."""
    A = [RNG.random((10, 10)) for _ in range(5)]
    _A = Jets.blockop([T.JopBaz(a) for a in A])
    hm = RNG.random(10)
    _y = _A * T.dev(hm)
    for i in range(5):
        assert approx(Jets.getblock(_y, i), A[i] @ hm)


def test_block_operator_3x4_with_nonlinear_zero_and_composite_blocks(Jets, T):
    """test/runtests.jl:622-695 -- the full set: dense, nonlinear, zero, adjoint and composite blocks."""
    B = {k: RNG.random((10, 10)) for k in ("11", "13", "14", "21", "23", "24", "32", "33")}
    A11, A13, A14, A21, A23, A32, A33 = (T.JopBaz(B[k]) for k in ("11", "13", "14", "21", "23", "32", "33"))
    A24 = T.JopBaz(B["24"]).H
    F12, F23, F31 = T.JopBar(10), T.JopBar(10), T.JopBar(10)
    spc = Jets.JetSpace(F64, 10)
    Z22, Z34 = Jets.JopZeroBlock(spc, spc), Jets.JopZeroBlock(spc, spc)
    assert Jets.iszero(Z22) and not Jets.iszero(A11) and not Jets.iszero(F12)
    C24 = A24 @ T.JopBar(10)
    F = Jets.blockop([[A11, F12, A13, A14], [A21, Z22, F23, C24], [F31, A32, A33, Z34]])
    assert isinstance(F, Jets.JopNl) and Jets.nblocks_op(F) == (3, 4)
    hm = RNG.random(40)
    m = Jets.from_numpy(hm, Jets.domain(F))
    d = (F * m).to_numpy()
    s = [hm[0:10], hm[10:20], hm[20:30], hm[30:40]]
    assert approx(d[0:10], B["11"] @ s[0] + s[1] ** 2 + B["13"] @ s[2] + B["14"] @ s[3])           # :664
    assert approx(d[10:20], B["21"] @ s[0] + s[2] ** 2 + B["24"].T @ s[3] ** 2)                     # :665
    assert approx(d[20:30], s[0] ** 2 + B["32"] @ s[1] + B["33"] @ s[2])                            # :666

    J = Jets.jacobian_(F, m)
    hdm = RNG.random(40)
    dm = Jets.from_numpy(hdm, Jets.domain(J))
    dd = J * dm
    t = [hdm[0:10], hdm[10:20], hdm[20:30], hdm[30:40]]
    e0 = B["11"] @ t[0] + 2 * s[1] * t[1] + B["13"] @ t[2] + B["14"] @ t[3]
    e1 = B["21"] @ t[0] + 2 * s[2] * t[2] + B["24"].T @ (2 * s[3] * t[3])
    e2 = 2 * s[0] * t[0] + B["32"] @ t[1] + B["33"] @ t[2]
    assert approx(dd, np.concatenate([e0, e1, e2]))

    # L = @blockop of the explicit jacobians (:672-681)
    J12 = Jets.jacobian_(F12, Jets.from_numpy(s[1]))
    J23 = Jets.jacobian_(F23, Jets.from_numpy(s[2]))
    J24 = Jets.jacobian_(C24, Jets.from_numpy(s[3]))
    J31 = Jets.jacobian_(F31, Jets.from_numpy(s[0]))
    L = Jets.blockop([[A11, J12, A13, A14], [A21, Z22, J23, J24], [J31, A32, A33, Z34]])
    assert isinstance(L, Jets.JopLn)
    assert approx(dd, L * dm)
    hdd = dd.to_numpy()
    u = [hdd[0:10], hdd[10:20], hdd[20:30]]
    a0 = B["11"].T @ u[0] + B["21"].T @ u[1] + 2 * s[0] * u[2]
    a1 = 2 * s[1] * u[0] + B["32"].T @ u[2]
    a2 = B["13"].T @ u[0] + 2 * s[2] * u[1] + B["33"].T @ u[2]
    a3 = B["14"].T @ u[0] + 2 * s[3] * (B["24"] @ u[1])
    expect_adj = np.concatenate([a0, a1, a2, a3])
    assert approx(L.H * dd, expect_adj) and approx(J.H * dd, expect_adj)                            # :682
    dirty = Jets.from_numpy(RNG.random(40), Jets.domain(L))
    assert approx(Jets.mul_(dirty, L.H, dd), expect_adj)                                            # :684
    assert Jets.eltype(L) == np.dtype(F64)
    K = Jets.convert_op(L)                                                                          # :686  K = convert(Array, L)
    assert K.shape == (30, 40) and approx(L * dm, K @ dm.to_numpy())                                # :688
    assert approx(L.H * dd, K.T @ hdd)
    _J12 = Jets.getblock_op(J, 0, 1)
    hx = RNG.random(10)
    assert approx(J12 * Jets.from_numpy(hx), _J12 * Jets.from_numpy(hx))                            # :691-694


def test_block_operator_singleton_tall_wide_with_dense_and_nonlinear(Jets, T):
    """test/runtests.jl:704-758."""
    Bm = RNG.random((5, 5))
    B = T.JopBaz(Bm)
    A = Jets.blockop([[B]])
    hm, hd = RNG.random(5), RNG.random(5)
    assert approx(A * T.dev(hm), Bm @ hm) and approx(A.H * T.dev(hd), Bm.T @ hd)
    F = T.JopBar(5)
    G = Jets.blockop([[F]])
    m = T.dev(hm)
    assert approx(F * m, G * m)
    assert approx(Jets.jacobian_(G, m).H * T.dev(hd), 2 * hm * hd)

    Bs = [RNG.random((5, 5)) for _ in range(3)]
    At = Jets.blockop([[T.JopBaz(b)] for b in Bs])                                                  # tall (:721-726)
    assert approx(At * m, np.concatenate([b @ hm for b in Bs]))
    hd15 = RNG.random(15)
    d15 = Jets.from_numpy(hd15, Jets.range(At))
    assert approx(At.H * d15, sum(b.T @ hd15[5 * i:5 * i + 5] for i, b in enumerate(Bs)))
    Ft = Jets.blockop([[T.JopBar(5)] for _ in range(3)])                                            # :728-733
    assert approx(Ft * m, np.concatenate([hm ** 2] * 3))
    Jt = Jets.jacobian_(Ft, m)
    assert approx(Jt * m, np.concatenate([2 * hm * hm] * 3))
    assert approx(Jt.H * d15, sum(2 * hm * hd15[5 * i:5 * i + 5] for i in range(3)))

    Aw = Jets.blockop([[T.JopBaz(b) for b in Bs]])                                                  # wide (:745-750)
    hm15 = RNG.random(15)
    m15 = Jets.from_numpy(hm15, Jets.domain(Aw))
    assert approx(Aw * m15, sum(b @ hm15[5 * j:5 * j + 5] for j, b in enumerate(Bs)))
    assert approx(Aw.H * T.dev(hd), np.concatenate([b.T @ hd for b in Bs]))
    Fw = Jets.blockop([[T.JopBar(5) for _ in range(3)]])                                            # :752-757
    assert approx(Fw * m15, sum(hm15[5 * j:5 * j + 5] ** 2 for j in range(3)))
    Jw = Jets.jacobian_(Fw, m15)
    assert approx(Jw.H * T.dev(hd), np.concatenate([2 * hm15[5 * j:5 * j + 5] * hd for j in range(3)]))


def test_getblock_of_adjoint_block_operator_values(Jets, T):
    """test/runtests.jl:760-787."""
    Bs = [[RNG.random((5, 5)) for _ in range(3)] for _ in range(2)]
    ops = [[T.JopBaz(b) for b in row] for row in Bs]
    C = Jets.blockop(ops).H
    hx = RNG.random(5)
    for i in range(2):
        for j in range(3):
            Cji = Jets.getblock_op(C, j, i)
            assert isinstance(Cji, Jets.JopAdjoint)
            assert approx(Cji * T.dev(hx), Bs[i][j].T @ hx)


def test_scalar_times_operator(Jets, T):
    """test/runtests.jl:789-795."""
    Bm = RNG.random((10, 10))
    hm = RNG.random(10)
    assert approx((3.14 * T.JopBaz(Bm)) * T.dev(hm), 3.14 * (Bm @ hm))


def test_vectorized_operators(Jets, T):
    """test/runtests.jl:797-838."""
    g = RNG.random((10, 11))
    A = Jets.JopDiagonal(T.dev(g))                                     # JopFoo2: N-d diagonal
    hx = RNG.random((10, 11))
    x = T.dev(hx)
    Bv = Jets.vec_op(A)
    assert Bv.jet.f is Jets.JetVec_f and Jets.domain(A).vec().size() == (110,)
    d = A * x
    _d = Bv * Jets.vec(x)
    assert d.shape == (10, 11) and _d.shape == (110,)
    assert approx(d, g * hx) and np.array_equal(_d.to_numpy(), d.to_numpy().ravel(order="F"))
    assert Jets.reshape(_d, Jets.range(A)).shape == (10, 11)
    Ab = Jets.blockop([Jets.JopDiagonal(T.dev(g)), Jets.JopDiagonal(T.dev(2 * g))])                # :819-838
    db = Ab * x
    _db = Jets.vec_op(Ab) * Jets.vec(x)
    assert np.array_equal(db.to_numpy(), _db.to_numpy())
    a = Ab.H * db
    _a = Jets.vec_op(Ab.H) * db
    assert a.shape == (10, 11) and np.array_equal(_a.to_numpy().ravel(order="F"), a.to_numpy().ravel(order="F"))


def test_dot_product_test_dense_and_complex(Jets, T):
    """test/runtests.jl:901-918 plus a rectangular dense operator."""
    A = T.JopFoo(RNG.random(10))
    lhs, rhs = Jets.dot_product_test(A, T.dev(RNG.random(10)), T.dev(RNG.random(10)))
    assert abs(lhs - rhs) <= 1e-13 * abs(lhs)
    Ac = T.JopFoo(RNG.random(10) + 1j * RNG.random(10))
    lhs, rhs = Jets.dot_product_test(Ac, T.dev(RNG.random(10) + 1j * RNG.random(10)), T.dev(RNG.random(10) + 1j * RNG.random(10)))
    assert np.iscomplexobj(lhs) and abs(lhs - rhs) <= 1e-13 * abs(lhs)
    D = T.JopBaz(RNG.random((7, 4)))
    lhs, rhs = Jets.dot_product_test(D, T.dev(RNG.random(4)), T.dev(RNG.random(7)))
    assert abs(lhs - rhs) <= 1e-13 * abs(lhs)
    Dc = T.JopBaz((RNG.random((300, 170)) + 1j * RNG.random((300, 170))).astype(np.complex64))
    lhs, rhs = Jets.dot_product_test(Dc, T.dev((RNG.random(170) + 1j * RNG.random(170)).astype(np.complex64)),
                                     T.dev((RNG.random(300) + 1j * RNG.random(300)).astype(np.complex64)))
    assert abs(lhs - rhs) <= 1e-5 * abs(lhs)


def test_dense_forward_is_bit_exact_and_adjoint_within_tolerance(Jets, oracle):
    """jh_gemv vs the oracle's dense child (test/runtests.jl:27-28): forward accumulates columns in order
    (bit-exact); adjoint reduces in fp64 (tolerance)."""
    for dt, tol in ((np.float32, 1e-6), (np.float64, 1e-14), (np.complex64, 1e-6), (np.complex128, 1e-14)):
        nr, nc = 129, 67
        hA = (RNG.random((nr, nc)) - 0.5).astype(dt)
        if np.dtype(dt).kind == "c":
            hA = (hA + 1j * (RNG.random((nr, nc)) - 0.5)).astype(dt)
        hx = (RNG.random(nc) - 0.5).astype(dt)
        hy = (RNG.random(nr) - 0.5).astype(dt)
        A = Jets.JopDense(Jets.from_numpy(np.asfortranarray(hA)))
        blk = oracle.Block("dense", nr, nc, coeff=np.asfortranarray(hA))
        fwd = (A * Jets.from_numpy(hx)).to_numpy()
        assert fwd.tobytes() == oracle.child_mul(blk, np.empty(nr, dtype=dt), hx).tobytes()
        adj = (A.H * Jets.from_numpy(hy)).to_numpy()
        ref = hA.astype(np.complex128).conj().T @ hy.astype(np.complex128)
        assert np.linalg.norm(adj - ref) <= tol * np.linalg.norm(hA.astype(np.complex128)) * np.linalg.norm(hy.astype(np.complex128))


@pytest.mark.parametrize("dt,tol", [(np.float32, 2e-6), (np.float64, 1e-14), (np.complex64, 2e-6)])
@pytest.mark.parametrize("nr,nc", [(4096, 4096), (64, 70000), (300000, 24), (1000, 1003), (5, 3)])
def test_dense_operator_shapes(Jets, dt, tol, nr, nc):
    """jh_gemv over the shapes that exercise each code path: square, few rows (column split), few columns (row split),
    odd sizes (unaligned columns -> scalar path), tiny."""
    rng = np.random.default_rng(nr * 7 + nc)
    hA = (rng.random((nr, nc)) - 0.5).astype(dt)
    if np.dtype(dt).kind == "c":
        hA = (hA + 1j * (rng.random((nr, nc)) - 0.5)).astype(dt)
    hx = (rng.random(nc) - 0.5).astype(dt)
    hy = (rng.random(nr) - 0.5).astype(dt)
    A = Jets.JopDense(Jets.from_numpy(np.asfortranarray(hA)))
    A64 = hA.astype(np.complex128)
    fwd = (A * Jets.from_numpy(hx)).to_numpy().astype(np.complex128)
    ref = A64 @ hx.astype(np.complex128)
    scale = np.linalg.norm(np.abs(A64) @ np.abs(hx.astype(np.complex128)))
    assert np.linalg.norm(fwd - ref) <= tol * max(nc, 64) ** 0.5 * scale
    adj = (A.H * Jets.from_numpy(hy)).to_numpy().astype(np.complex128)
    refa = A64.conj().T @ hy.astype(np.complex128)
    scalea = np.linalg.norm(np.abs(A64).T @ np.abs(hy.astype(np.complex128)))
    assert np.linalg.norm(adj - refa) <= tol * scalea
    lhs, rhs = Jets.dot_product_test(A, Jets.from_numpy(hx), Jets.from_numpy(hy))
    assert abs(lhs - rhs) <= 50 * tol * max(abs(lhs), scalea * np.linalg.norm(hx) / max(nc, 1) ** 0.5)



# ---------------------------------------------------------------------------------- the remaining value-level test sets
def test_linear_operator_set(Jets, T):
    """test/runtests.jl:126-168."""
    diag, hm = RNG.random(10), RNG.random(10)
    A = T.JopFoo(diag)
    m = T.dev(hm)
    d = A * m
    assert approx(d, diag * hm)
    a = A.H * d
    assert approx(a, diag * d.to_numpy())
    Jets.fill_(d, 0)
    Jets.mul_(d, A, m)
    assert approx(d, diag * hm)
    Jets.fill_(a, 0)
    Jets.mul_(a, A.H, d)
    assert approx(a, diag * d.to_numpy())
    assert Jets.size(A) == (10, 10) and Jets.shape(A) == ((10,), (10,)) and Jets.size(A, 1) == 10 and Jets.shape(A, 2) == (10,)
    assert Jets.domain(A) == Jets.JetSpace(F64, 10) and Jets.range(A) == Jets.JetSpace(F64, 10) and Jets.eltype(A) == np.dtype(F64)
    assert approx(Jets.convert_op(A), np.diag(diag))                                   # :151
    assert approx(Jets.state(A)["diagonal"], diag) and approx(Jets.state(A, "diagonal"), diag)
    # JopFooBar (:35-39): a linear operator with df! only (df'! defaults to df!), 2-D space
    hA = RNG.random((5, 5))
    spc = Jets.JetSpace(F64, 5, 5)
    B = Jets.JopLn(df=lambda d, m, *, A, **kw: Jets.hadamard_(d, A, m), dom=spc, rng=spc, s={"A": T.dev(hA)})
    assert approx(Jets.convert_op(B), np.diag(hA.ravel(order="F")))                   # :157
    m, d = Jets.rand(Jets.domain(B)), Jets.rand(Jets.range(B))
    for mk in (Jets.jacobian, Jets.jacobian_):                                        # :160-167: the jacobian of a linear operator is itself
        assert approx(mk(B, Jets.rand(Jets.domain(B))) * m, B * m)
        assert approx(mk(B.H, Jets.rand(Jets.domain(B.H))).H * d, B.H * d)


def test_nonlinear_operator_and_upstate_sets(Jets, T):
    """test/runtests.jl:170-201."""
    F = T.JopBar(10)
    m = Jets.rand(Jets.domain(F))
    hm = m.to_numpy()
    d = F * m
    assert approx(d, hm ** 2)
    Jets.fill_(d, 0)
    Jets.mul_(d, F, m)
    assert approx(d, hm ** 2)
    J = Jets.jacobian_(F, m)
    assert approx(Jets.point(J), hm)
    d = J * m
    assert approx(d, 2 * hm * hm)
    assert approx(J.H * d, 2 * hm * d.to_numpy())
    assert Jets.size(F) == (10, 10) and Jets.shape(F) == ((10,), (10,)) and Jets.domain(F) == Jets.JetSpace(F64, 10)
    # JopRosenbrock (:41-50): upstate! rewrites part of the state at every point!
    spc = Jets.JetSpace(F64, 2)

    def f(d, m, **kw):
        h = m.to_numpy()
        return Jets.copyto_(d, Jets.from_numpy(np.array([1 - h[0], 10 * (h[1] - h[0] ** 2)])))

    def upstate(m, s):
        Jm = s["J"].to_numpy()
        Jm[1, 0] = -20.0 * m.to_numpy()[0]
        Jets.copyto_(s["J"], Jets.from_numpy(np.asfortranarray(Jm)))

    R = Jets.JopNl(f=f, df=lambda d, m, *, J, **kw: Jets.mul_(d, Jets.JopDense(J), m),
                   df_adj=lambda m, d, *, J, **kw: Jets.mul_(m, Jets.JopDense(J).H, d), upstate=upstate, dom=spc, rng=spc,
                   s={"J": T.dev(np.array([[-1.0, 0.0], [0.0, 10.0]]))})
    hm = RNG.random(2)
    Jr = Jets.jacobian_(R, T.dev(hm))
    assert approx(Jets.state(Jr)["J"], np.array([[-1.0, 0.0], [-20 * hm[0], 10.0]]))   # :200
    assert approx(Jr * T.dev(np.array([1.0, 2.0])), np.array([[-1.0, 0.0], [-20 * hm[0], 10.0]]) @ np.array([1.0, 2.0]))


def test_composition_and_sum_with_matrix_operands(Jets, T):
    """test/runtests.jl:328-356 and 490-498: a plain matrix takes part in `o`, `+` and `-`."""
    B1, B2, B3, B4 = (RNG.random((10, 10)) for _ in range(4))
    A1, A2, A4 = T.JopBaz(B1), T.JopBaz(B2), T.JopBaz(B4)
    A3 = T.dev(B3)                                                                    # A3 = rand(10,10): a matrix, not an operator
    A21, A321 = A2 @ A1, A3 @ A2 @ A1
    A4321 = A4 @ A3 @ A2 @ A1
    hm = RNG.random(10)
    m = T.dev(hm)
    assert approx(A21 * m, B2 @ (B1 @ hm)) and approx(A321 * m, B3 @ (B2 @ (B1 @ hm)))
    d = A4321 * m
    hd = d.to_numpy()
    assert approx(d, B4 @ (B3 @ (B2 @ (B1 @ hm))))
    assert approx(A21.H * d, B1.T @ (B2.T @ hd)) and approx(A321.H * d, B1.T @ (B2.T @ (B3.T @ hd)))
    assert approx(A4321.H * d, B1.T @ (B2.T @ (B3.T @ (B4.T @ hd))))
    assert Jets.domain(A4321) == Jets.JetSpace(F64, 10) and Jets.eltype(A4321) == np.dtype(F64)
    assert approx(Jets.convert_op(A4321) @ hm, A4321 * m)                             # :354-355
    S12 = A1 + A3                                                                     # :493   A1 + A2 (matrix)
    S123 = A1 + A3 - A4
    assert approx(S12 * m, B1 @ hm + B3 @ hm) and approx(S123 * m, B1 @ hm + B3 @ hm - B4 @ hm)


def test_composition_operator_times_block_operator(Jets, T):
    """test/runtests.jl:425-436: getblock of (block operator o operator) composes per block."""
    A1 = T.JopFoo(RNG.random(2))
    A2 = Jets.blockop([T.JopBar(2), T.JopBar(2)])
    A = A2 @ A1
    A11, A21 = Jets.getblock_op(A, 0, 0), Jets.getblock_op(A, 1, 0)
    m = Jets.rand(Jets.domain(A))
    Am = A * m
    assert approx(Jets.getblock(Am, 0), A11 * m) and approx(Jets.getblock(Am, 1), A21 * m)


# ---------------------------------------------------------------------------------- symmetric spaces (test/runtests.jl:218-294)
def _indexmap(I):                                           # 0-based twin of test/runtests.jl:218-224
    return I if I[0] < 4 else (I[0] - 4, I[1])


def test_symmetric_space(Jets):
    """test/runtests.jl:227-257."""
    C128 = np.complex128
    R = Jets.JetSSpace(C128, (8, 4), (4, 4), _indexmap)
    assert tuple(R.size()) == (8, 4) and R.eltype() == np.dtype(C128)
    assert np.array_equal(Jets.ones(R).to_numpy(), np.ones((8, 4), C128)) and np.array_equal(Jets.zeros(R).to_numpy(), np.zeros((8, 4), C128))
    assert Jets.rand(R).shape == (8, 4) and Jets.Array(R).shape == (8, 4) and Jets.Array(R).dtype == np.dtype(C128)
    _ = Jets.randn(R)
    x = Jets.rand(R)
    z = Jets.similar(x)
    assert isinstance(z, Jets.SymmetricArray) and z.shape == (8, 4)
    y = x.A                                                                           # the stored block
    hy = y.to_numpy()
    assert Jets.norm(x) == pytest.approx(np.sqrt(2 * np.linalg.norm(hy) ** 2), rel=1e-12)         # :243
    assert Jets.norm(x, 2) == pytest.approx(float(Jets.norm(x)), rel=1e-14)
    assert Jets.norm(x, 1) == pytest.approx(2 * np.abs(hy).sum(), rel=1e-12)
    assert Jets.norm(x, np.inf) == pytest.approx(np.abs(hy).max(), rel=1e-14)
    x[0, 0] = 0
    x[5, 0] = 0                                                                       # x[1,1] = x[6,1] = 0  (:247)
    assert Jets.norm(x, 0) == pytest.approx(2 * np.count_nonzero(x.A.to_numpy()), rel=1e-14)
    assert Jets.space(Jets.rand(R)) == R
    assert Jets.JetSSpace(C128, (0, 0), R.M, R.map) == R.similar((0, 0))
    for i in range(32):                                                               # :252-256, linear indices (column-major)
        x[i] = (i + 1) + (i + 1) * 1j
        assert x[i] == (i + 1) + (i + 1) * 1j
    full = x.to_numpy()
    assert full[6, 1] == np.conj(x.A.to_numpy()[2, 1])                                # outside the stored block: the conjugate


def test_symmetric_spaces_broadcast(Jets):
    """test/runtests.jl:259-294."""
    R = Jets.JetSSpace(np.complex128, (8, 4), (4, 4), _indexmap)
    u, v, w = Jets.rand(R), Jets.rand(R), Jets.rand(R)
    a, b, c = RNG.random(3)
    x = (a * u + b * v + c * w).materialize()
    assert isinstance(x, Jets.SymmetricArray)
    want = a * u.A.to_numpy() + b * v.A.to_numpy() + c * w.A.to_numpy()
    assert approx(x.A, want)
    y = Jets.zeros(R)
    y.assign(x)
    assert approx(y.A, want) and np.allclose(y.to_numpy(), x.to_numpy())
    y.assign(-1.0 * y)                                                                # y .*= -1
    assert approx(y.parent, -want)
    z = Jets.abs_(x)
    assert isinstance(z, np.ndarray) and z.dtype == np.float64 and z.shape == (8, 4)  # typeof(z) == Array{Float64,2}
    full = x.to_numpy()
    assert np.allclose(z, np.abs(full))
