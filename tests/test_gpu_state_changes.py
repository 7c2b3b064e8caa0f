"""GPU: `state!` on a child takes effect on the next block `mul!` (src/Jets.jl:272 merges, :391 splats the CURRENT state into
the closure), also through every fused path that captured raw coefficient pointers; and the fused chains refuse an
un-pointed Jacobian exactly like the unfused path (advisor findings, round 1)."""
import numpy as np
import pytest

from .helpers import assert_bits_equal, u01

pytestmark = pytest.mark.gpu


def _tall(Jets, oracle, nrow, n, seed):
    spc = Jets.JetSpace(np.float32, n)
    diags = [Jets.rand(spc, seed=seed, stream=i) for i in range(nrow)]
    kids = [Jets.JopDiagonal(g) for g in diags]
    return Jets.blockop([[k] for k in kids]), kids, [u01(oracle, np.float32, seed, i, n) for i in range(nrow)]


def test_state_change_of_a_child_reaches_every_native_path(Jets, oracle):
    nrow, n = 5, 4096
    A, kids, ha = _tall(Jets, oracle, nrow, n, 31)
    m = Jets.rand(Jets.domain(A), seed=2, stream=0)
    hm = u01(oracle, np.float32, 2, 0, n)
    d = A * m                                                           # builds and caches the device handle
    assert_bits_equal(d.to_numpy(), np.concatenate([a * hm for a in ha]), "before state!")
    # state!(child 3, diagonal = new): the reference's next mul! sees the new array
    new = Jets.rand(Jets.JetSpace(np.float32, n), seed=77, stream=0)
    hnew = u01(oracle, np.float32, 77, 0, n)
    Jets.state_(kids[3], {"diagonal": new})
    ha2 = list(ha)
    ha2[3] = hnew
    ops = [[oracle.Block("diag", n, coeff=a)] for a in ha2]
    want_fwd = np.concatenate(oracle.block_df(ops, [np.zeros(n, np.float32) for _ in range(nrow)], [hm]))
    assert_bits_equal((A * m).to_numpy(), want_fwd, "forward after state!")
    hd = [want_fwd[i * n:(i + 1) * n] for i in range(nrow)]
    want_adj = oracle.block_df_adj(ops, [np.zeros(n, np.float32)], hd)[0]
    assert_bits_equal((A.H * (A * m)).to_numpy().ravel(order="F"), want_adj, "adjoint after state!")
    assert_bits_equal(((A.H @ A) * m).to_numpy().ravel(order="F"), want_adj, "fused A'A after state!")
    # LSQR (native loop) solves with the NEW operator
    x_true = Jets.rand(Jets.domain(A), seed=4, stream=0)
    res = Jets.lsqr(A, A * x_true, atol=0.0, btol=0.0, conlim=0.0, maxiter=30)
    hx = u01(oracle, np.float32, 4, 0, n)
    assert np.linalg.norm(res.x.to_numpy().ravel() - hx) <= 1e-3 * np.linalg.norm(hx)
    # an unrelated state! keeps the handle (no rebuild): same object, same results
    cell = A.jet.s["_native"]
    before = cell.value
    Jets.state_(kids[0], {"note": 1})
    assert_bits_equal((A * m).to_numpy(), want_fwd, "forward after an unrelated state!")
    assert cell.value is before


def test_fused_chains_refuse_an_unpointed_jacobian(Jets):
    """A tall operator of JopElementwise children that was never pointed: the unfused Jacobian raises the reference's
    DimensionMismatch; the fused A'A, the fused sum and LSQR must not run on the all-zero diagonal instead."""
    spc = Jets.JetSpace(np.float32, 2048)
    F = Jets.blockop([[Jets.JopElementwise(spc, "x0*x0*x0", "3*x0*x0")] for _ in range(3)])
    Jn = Jets.JopLn(Jets.jet(F))
    m = Jets.rand(spc, seed=5, stream=0)
    with pytest.raises(Exception):
        Jets.mul(Jn, m)
    with pytest.raises(Exception):
        Jets.mul(Jn.H @ Jn, m)
    with pytest.raises(Exception):
        Jets.lsqr(Jn, Jets.rand(Jets.range(Jn), seed=6, stream=0), maxiter=2)
    # once pointed, all three run and agree
    Jets.point_(Jets.jet(F), m)
    y1 = Jn.H * (Jn * m)
    y2 = (Jn.H @ Jn) * m
    assert_bits_equal(y1.to_numpy(), y2.to_numpy(), "fused vs chained after point!")
