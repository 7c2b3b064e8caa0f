"""GPU: operators that mix small DENSE children (adjointed or not) with the elementwise kinds run the reference's block loops in
ONE launch (k_block_loop_small) -- the shape of the reference's own 3 x 4 test operator (test/runtests.jl:622-695: JopBaz
children, one adjointed, Jacobians of JopBar, zero blocks).  Every thread forms a dense child's dot product sequentially, as the
oracle does, so forward AND adjoint are bit-exact (the per-child kernels' adjoint is tolerance parity)."""
import numpy as np
import pytest

from .helpers import DTYPES, assert_bits_equal, u01, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _the_loop_at_any_size(Jets):
    """Late round 5: operators whose dense children TOGETHER reach 512 KiB take the batched list route instead (knob small_loop_max_kib; tolerance parity
    there -- tests/test_gpu_dense_lists.py); these tests are about the one-launch loop and its bits, so they pin it."""
    Jets.tune(small_loop_max_kib=1 << 40)
    yield
    Jets.tune(small_loop_max_kib=512)


def _reference_3x4(J, oracle, dt, n, seed=5):
    """[A11 J12 A13 A14; A21 Z22 J23 A24'; J31 A32 A33 Z34] with dense A (n x n), diagonal J (the Jacobian of JopBar), zero Z."""
    layout = [["dense", "diag", "dense", "dense"], ["dense", "zero", "diag", "dense_adj"], ["diag", "dense", "dense", "zero"]]
    spc = J.JetSpace(dt, n)
    dev, ora = [], []
    for i, row in enumerate(layout):
        dr, orow = [], []
        for j, k in enumerate(row):
            st = 10 * i + j
            if k.startswith("dense"):
                hA = np.asfortranarray(u01(oracle, dt, seed, st, n * n).reshape((n, n), order="F"))
                op = J.JopDense(J.from_numpy(hA))
                dr.append(op.H if k.endswith("adj") else op)
                orow.append(oracle.Block("dense", n, n, coeff=hA, adjoint=k.endswith("adj")))
            elif k == "diag":
                dr.append(J.JopDiagonal(J.rand(spc, seed=seed, stream=st)))
                orow.append(oracle.Block("diag", n, coeff=u01(oracle, dt, seed, st, n)))
            else:
                dr.append(J.JopZeroBlock(spc, spc)); orow.append(oracle.Block("zero", n))
        dev.append(dr); ora.append(orow)
    return J.blockop(dev), ora


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("n", [10, 64, 128])       # 128 x 128 x 16 B = 256 KiB: the largest child of the one-launch loop
def test_reference_3x4_shape_in_one_launch_bit_exact(Jets, oracle, dt, n):
    J = Jets
    A, ora = _reference_3x4(J, oracle, dt, n)
    hm = [u01(oracle, dt, 2, j, n) for j in range(4)]
    hd_found = [u01(oracle, dt, 3, i, n) for i in range(3)]
    m = J.from_numpy(np.concatenate(hm), J.domain(A))
    d = J.from_numpy(np.concatenate(hd_found), J.range(A))                  # dirty: accumulated into (1024)
    J.mul_(d, A, m)
    want = oracle.block_df(ora, [b.copy() for b in hd_found], hm)
    assert_bits_equal(d.to_numpy(), np.concatenate(want), "forward")
    hmt = [u01(oracle, dt, 4, j, n) for j in range(4)]
    mt = J.from_numpy(np.concatenate(hmt), J.domain(A))                     # dirty: zeroed (1042)
    J.mul_(mt, A.H, d)
    want_m = oracle.block_df_adj(ora, [b.copy() for b in hmt], want)
    assert_bits_equal(mt.to_numpy(), np.concatenate(want_m), "adjoint")
    lhs, rhs = J.dot_product_test(A, J.rand(J.domain(A), seed=7), J.rand(J.range(A), seed=8))
    assert abs(lhs - rhs) <= (1e-4 if np.dtype(dt).itemsize // (2 if np.dtype(dt).kind == "c" else 1) == 4 else 1e-11) * abs(lhs + rhs)
    # the per-block loop (knob small_loop = 0) computes the same operator within the tolerance of its fp64 wave reductions
    try:
        J.tune(small_loop=0)
        d2 = J.from_numpy(np.concatenate(hd_found), J.range(A))
        J.mul_(d2, A, m)
        assert rel_err(d2.to_numpy(), np.concatenate(want)) < 1e-5      # its adjointed child reduces in fp64 across a wave: tolerance
        mt2 = J.zeros(J.domain(A))
        J.mul_(mt2, A.H, d)
        assert rel_err(mt2.to_numpy(), np.concatenate(want_m)) < 1e-5
    finally:
        J.tune(small_loop=1)


def test_wide_and_tall_mixes_with_adjointed_children(Jets, oracle):
    """1 x K and N x 1 operators of dense and adjointed-dense children of different shapes (so no uniform batch applies)."""
    J = Jets
    dt = np.float64
    shapes = [(6, 9, False), (9, 6, True), (6, 6, False)]                   # block j maps R^{c_j} -> R^6: (nr, nc, adjoint)
    dev, ora, col_len = [], [], []
    for j, (nr, nc, adj) in enumerate(shapes):
        hA = np.asfortranarray(u01(oracle, dt, 11, j, nr * nc).reshape((nr, nc), order="F"))
        op = J.JopDense(J.from_numpy(hA))
        dev.append(op.H if adj else op)
        ora.append(oracle.Block("dense", nr, nc, coeff=hA, adjoint=adj))
        col_len.append(nr if adj else nc)
    W = J.blockop([dev])                                                    # 1 x 3
    hm = [u01(oracle, dt, 12, j, n) for j, n in enumerate(col_len)]
    hd = [u01(oracle, dt, 13, 0, 6)]
    d = J.from_numpy(hd[0], J.range(W))
    J.mul_(d, W, J.from_numpy(np.concatenate(hm), J.domain(W)))
    want = oracle.block_df([ora], [hd[0].copy()], hm)
    assert_bits_equal(d.to_numpy(), want[0], "wide forward")
    mt = J.zeros(J.domain(W))
    J.mul_(mt, W.H, d)
    want_m = oracle.block_df_adj([ora], [np.zeros(n, dt) for n in col_len], want)
    assert_bits_equal(mt.to_numpy(), np.concatenate(want_m), "wide adjoint (single row: direct writes, 1051)")
    T = J.blockop([[op.H] for op in dev])                                   # 3 x 1, every child adjointed once more
    ora_t = [[oracle.Block("dense", b.nr, b.nc, coeff=b.coeff, adjoint=not b.adjoint)] for b in ora]
    hx = u01(oracle, dt, 14, 0, 6)
    dt_ = J.zeros(J.range(T))
    J.mul_(dt_, T, J.from_numpy(hx))
    want_t = oracle.block_df(ora_t, [np.zeros(n, dt) for n in col_len], [hx])
    assert_bits_equal(dt_.to_numpy(), np.concatenate(want_t), "tall forward")
    back = J.zeros(J.domain(T))
    J.mul_(back, T.H, dt_)
    assert_bits_equal(back.to_numpy().ravel(), oracle.block_df_adj(ora_t, [np.zeros(6, dt)], want_t)[0], "tall adjoint")


def test_nonlinear_f_mode_with_dense_children_in_one_launch(Jets, oracle):
    """JetBlock_f! (988-1008) on [A11 F12; F21 Z22] with F = JopSquare: every child applied, the zero block adds +0."""
    J = Jets
    dt, n = np.float32, 24
    spc = J.JetSpace(dt, n)
    hA = np.asfortranarray(u01(oracle, dt, 21, 0, n * n).reshape((n, n), order="F"))
    F = J.blockop([[J.JopDense(J.from_numpy(hA)), J.JopSquare(spc)], [J.JopSquare(spc), J.JopZeroBlock(spc, spc)]])
    hm = [u01(oracle, dt, 22, j, n) for j in range(2)]
    m = J.from_numpy(np.concatenate(hm), J.domain(F))
    d = F * m
    ora = [[oracle.Block("dense", n, n, coeff=hA), oracle.Block("square", n, coeff=hm[1])],
           [oracle.Block("square", n, coeff=hm[0]), oracle.Block("zero", n)]]
    want = oracle.block_f(ora, [np.zeros(n, dt) for _ in range(2)], hm)
    assert_bits_equal(d.to_numpy(), np.concatenate(want), "F(m)")
    Jm = J.jacobian_(F, m)
    hdm = [u01(oracle, dt, 23, j, n) for j in range(2)]
    dd = Jm * J.from_numpy(np.concatenate(hdm), J.domain(F))
    want_j = oracle.block_df(ora, [np.zeros(n, dt) for _ in range(2)], hdm)
    assert_bits_equal(dd.to_numpy(), np.concatenate(want_j), "J(m) dm")
    back = Jm.H * dd
    assert_bits_equal(back.to_numpy(), np.concatenate(oracle.block_df_adj(ora, [np.zeros(n, dt) for _ in range(2)], want_j)), "J(m)' dd")
