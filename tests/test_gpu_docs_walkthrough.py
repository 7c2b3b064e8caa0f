"""GPU: the reference's user guide (docs/src/index.md) walked through with the Python mirror on device arrays -- every code
example of the sections on jets, operators, compositions, linear combinations, block operators and `vec` + lsqr, with the
guide's own "ground truth" comparisons.  Closures are written the way the guide writes them, over HIP-backed arrays
(broadcasts through the JIT-fused kernels)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def approx(x, y, rtol=1e-12):
    x = x.to_numpy() if hasattr(x, "to_numpy") else np.asarray(x)
    y = y.to_numpy() if hasattr(y, "to_numpy") else np.asarray(y)
    return np.linalg.norm(x.ravel(order="F") - y.ravel(order="F")) <= rtol * max(np.linalg.norm(x), np.linalg.norm(y), 1e-300)


def test_vector_spaces_and_convenience_methods(Jets):
    """docs/src/index.md:44-84."""
    R1, R2, R3 = Jets.JetSpace(np.float32, 10), Jets.JetSpace(np.float64, 10, 20), Jets.JetSpace(np.complex64, 10, 20, 2)
    x1, x2, x3 = Jets.rand(R1), Jets.rand(R2), Jets.rand(R3)
    assert (x1.shape, x1.dtype) == ((10,), np.float32) and (x2.shape, x2.dtype) == ((10, 20), np.float64)
    assert (x3.shape, x3.dtype) == ((10, 20, 2), np.complex64)
    assert R2.eltype() == np.float64 and R2.ndims() == 2 and R2.length() == 200 and tuple(R2.size()) == (10, 20)
    assert Jets.ones(R2).to_numpy().sum() == 200 and Jets.zeros(R2).to_numpy().sum() == 0 and Jets.Array(R2).shape == (10, 20)
    assert Jets.reshape(Jets.rand(Jets.JetSpace(np.float64, 200)), R2).shape == (10, 20)
    assert tuple(Jets.vec(R3).size()) == (400,)


def _power_jet(Jets, a=2.0, n=5):
    """docs/src/index.md:108-128: f(x) = x^a with its (self-adjoint) linearization."""
    def foo(d, m, *, a, **kw):                                   # foo!(d, m; a, kwargs...) = d .= m.^a
        return Jets.broadcast_(d, "pow(x0, s0)", [m], [a])

    def dfoo(dd, dm, *, mo, a, **kw):                            # dfoo!(dd, dm; mo, a, kwargs...) = dd .= a * mo.^(a-1) .* dm
        return Jets.broadcast_(dd, "s0 * pow(x0, s0 - 1) * x1", [mo, dm], [a])

    spc = Jets.JetSpace(np.float64, n)
    return Jets.Jet(dom=spc, rng=spc, f=foo, df=dfoo, s={"a": a})


def test_jets_and_operators(Jets):
    """docs/src/index.md:135-171."""
    myjet = _power_jet(Jets)
    assert Jets.state(myjet)["a"] == 2.0 and Jets.shape(myjet, 1) == (5,) and Jets.size(myjet) == (5, 5)
    F = Jets.JopNl(myjet)
    m = Jets.rand(Jets.domain(F))
    d1 = Jets.mul_(Jets.Array(Jets.range(F)), F, m)                                    # mul!(d1, F, m)
    d2 = F * m
    d3 = m.to_numpy() ** 2                                                            # ground truth
    assert approx(d1, d3) and approx(d2, d3)
    mo = Jets.rand(Jets.domain(F))
    dF = Jets.JopLn(myjet, mo)                                                        # JopLn(myjet, mo)
    dF2 = Jets.jacobian(F, mo)
    M1, M2 = Jets.convert_op(dF), Jets.convert_op(dF2)
    assert approx(M1, np.diag(2 * mo.to_numpy())) and approx(M2, M1)
    dm = Jets.rand(Jets.domain(dF))
    dd1 = Jets.mul_(Jets.Array(Jets.range(dF)), dF, dm)
    dd3 = 2 * mo.to_numpy() * (2 - 1) * dm.to_numpy()                                 # ground truth
    assert approx(dd1, dd3) and approx(dF * dm, dd3)
    d = Jets.rand(Jets.range(dF))
    a1 = Jets.mul_(Jets.Array(Jets.domain(dF)), dF.H, d)                              # mul!(a1, dF', d): self-adjoint dfoo!
    a3 = 2 * mo.to_numpy() * (2 - 1) * d.to_numpy()
    assert approx(a1, a3) and approx(dF.H * d, a3)
    Jets.state_(myjet, {"a": 3.0})                                                    # state!(jet, s)
    assert approx(F * m, m.to_numpy() ** 3)


def test_compositions_and_linear_combinations_with_a_matrix_operand(Jets):
    """docs/src/index.md:179-200: A3 is a plain matrix, not an operator."""
    rng = np.random.default_rng(5)
    spc = Jets.JetSpace(np.float64, 10)
    h1, h2, hA3 = rng.random(10), rng.random(10), rng.random((10, 10))
    A1, A2 = Jets.JopDiagonal(Jets.from_numpy(h1)), Jets.JopDiagonal(Jets.from_numpy(h2))
    A3 = Jets.from_numpy(np.asfortranarray(hA3))                                      # a device MATRIX
    A = A3 @ A2 @ A1                                                                  # A3 o A2 o A1
    m = Jets.rand(Jets.domain(A))
    hm = m.to_numpy()
    assert approx(A * m, hA3 @ (h2 * (h1 * hm)))                                      # A*m ≈ A3*(A2*(A1*m))
    assert approx(A.H * m, h1 * (h2 * (hA3.T @ hm)))
    B = hA3 @ A1                                                                      # a HOST matrix composes too
    assert approx(B * m, hA3 @ (h1 * hm))
    L = 1.0 * A1 - 2.0 * A2 + 3.0 * A3                                                # operator linear combination
    assert approx(L * m, 1.0 * (h1 * hm) - 2.0 * (h2 * hm) + 3.0 * (hA3 @ hm))
    assert approx(L.H * m, h1 * hm - 2.0 * (h2 * hm) + 3.0 * (hA3.T @ hm))
    L2 = A3 + A1 - hA3                                                                # matrix on the left, host matrix on the right
    assert approx(L2 * m, h1 * hm, rtol=1e-10)
    C = A1 @ A3                                                                       # operator o matrix  (:575)
    assert approx(C * m, h1 * (hA3 @ hm))


def test_block_operators_spaces_and_vectors(Jets):
    """docs/src/index.md:204-232."""
    spc = Jets.JetSpace(np.float64, 10)
    A = Jets.blockop([[Jets.JopDiagonal(Jets.rand(spc)) for _ in range(3)] for _ in range(2)])
    A12 = Jets.getblock_op(A, 0, 1)
    assert isinstance(A12, Jets.JopLn) and Jets.nblocks_op(A) == (2, 3) and Jets.nblocks_op(A, 1) == 2 and Jets.nblocks_op(A, 2) == 3
    d, m = Jets.rand(Jets.range(A)), Jets.rand(Jets.domain(A))
    assert Jets.nblocks(d) == 2 and Jets.nblocks(m) == 3
    d2 = Jets.getblock(d, 1)                                                          # a reference, not a copy
    new = np.random.default_rng(6).random(10)
    Jets.setblock_(d, 1, new)
    assert np.array_equal(d2.to_numpy(), new) and np.array_equal(d.to_numpy()[10:], new)
    _d = Jets.rand(Jets.JetSpace(np.float64, 20))
    dr = Jets.reshape(_d, Jets.range(A))                                              # reshape(_d, range(A)) shares memory
    Jets.fill_(Jets.getblock(dr, 0), 7.0)
    assert (_d.to_numpy()[:10] == 7.0).all()


def test_vectorized_operator_with_lsqr(Jets):
    """docs/src/index.md:239-245: m = reshape(lsqr(vec(A), vec(d)), range(A)); A*m ≈ d -- with an orthogonal 2-D operator
    standing in for JopDct (a separable transform is a JetPack operator, not part of Jets.jl)."""
    rng = np.random.default_rng(7)
    n1, n2 = 16, 8
    Q, _ = np.linalg.qr(rng.random((n1 * n2, n1 * n2)))
    spc = Jets.JetSpace(np.float64, n1, n2)
    dQ = Jets.from_numpy(np.asfortranarray(Q))

    def df(d, m, *, A, **kw):
        return Jets.reshape(Jets.mul_(Jets.vec(d), Jets.JopDense(A), Jets.vec(m)), spc)

    def df_adj(m, d, *, A, **kw):
        return Jets.reshape(Jets.mul_(Jets.vec(m), Jets.JopDense(A).H, Jets.vec(d)), spc)

    A = Jets.JopLn(dom=spc, rng=spc, df=df, df_adj=df_adj, s={"A": dQ})
    d = Jets.rand(Jets.range(A))
    x = Jets.lsqr(Jets.vec_op(A), Jets.vec(d), atol=1e-14, btol=1e-14, maxiter=50).x
    m = Jets.reshape(x, Jets.range(A))
    assert m.shape == (n1, n2) and approx(A * m, d, rtol=1e-10)                       # A*m ≈ d  # true


def test_readme_usage_example(Jets):
    """The snippet in README.md, at a small size."""
    R = Jets.JetSpace(np.float32, 16, 16, 16)
    A = Jets.blockop([[Jets.JopDiagonal(Jets.rand(R))] for _ in range(8)])
    m = Jets.rand(Jets.domain(A))
    d = A * m
    mt = A.H * d
    y = (A.H @ A) * m
    assert np.array_equal(y.to_numpy(), mt.to_numpy())
    x = Jets.lsqr(A, d, maxiter=20).x
    assert approx(x, m, rtol=1e-3)
    F = Jets.blockop([[Jets.JopElementwise(R, "exp(x0)", "exp(x0)")], [Jets.JopSquare(R)]])
    J = Jets.jacobian_(F, m)
    dd = J * m
    hm = m.to_numpy().astype(np.float64)
    assert approx(dd, np.concatenate([(np.exp(hm) * hm).ravel(order="F"), (2 * hm * hm).ravel(order="F")]), rtol=1e-5)
    Jets.broadcast_(mt, "s0*x0 + sqrt(abs(x1))", [m, y], [0.5])
    assert approx(mt, 0.5 * hm + np.sqrt(np.abs(y.to_numpy().astype(np.float64))), rtol=1e-5)
