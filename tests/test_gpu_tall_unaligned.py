"""Tall operators whose rows are NOT whole, 16-byte aligned packs (round 5, session 3).  A block vector is one contiguous slab (src/Jets.jl:742-748), so with
an odd block length -- 101^3 elements -- most block rows start off a 16-byte boundary and end inside a pack.  Such operators run the MIXED instantiations of
the tall kernels on under-aligned packs (jh_tall.hip: tall_unaligned_ok; jh_blockop_common.h: ldu / st_pack): the same terms in the same order as the general
kernels they used to fall to, so the oracle's bits -- forward (src/Jets.jl:1015-1031), adjoint (1042-1053), the fused A'A (530-534) -- and nothing written
outside the rows."""
import numpy as np
import pytest

from .helpers import assert_bits_equal, u01
from .test_gpu_blockop import _mixed_ops

pytestmark = pytest.mark.gpu

DTYPES = [np.float32, np.float64, np.complex64, np.complex128]
# block lengths: odd, 2 mod 4, 3 mod 4; shorter than a tile (256 packs), a tile plus a bit, several tiles with a ragged last one; the smallest that holds a pack
LENGTHS = [5, 7, 67, 1025, 1026, 1027, 4099, 2 * 1024 * 4 + 1, 6 * 1024 + 3]


def _kinds(nrow, name):
    names = ["diag", "diag_adj", "identity", "scale", "zero"]
    if name == "diag":
        return [["diag"] for _ in range(nrow)]
    return [[names[(3 * i + i // 5) % 5]] for i in range(nrow)]


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("n", LENGTHS)
@pytest.mark.parametrize("nrow,name", [(2, "diag"), (7, "diag"), (33, "diag"), (3, "mixed"), (18, "mixed")])
def test_tall_rows_off_the_pack_grid_have_the_oracles_bits(Jets, oracle, dt, n, nrow, name):
    J = Jets
    if n * np.dtype(dt).itemsize < 16:
        pytest.skip("less than one pack per row: the general kernels")
    kinds = _kinds(nrow, name)
    A, ops = _mixed_ops(J, oracle, dt, kinds, [n] * nrow, [n])
    hm = [u01(oracle, dt, 81, 0, n)]
    hd = [u01(oracle, dt, 82, i, n) for i in range(nrow)]
    want_d = oracle.block_df(ops, [b.copy() for b in hd], hm)           # into a DIRTY d: zero rows stay as found (1022)
    want_m = oracle.block_df_adj(ops, [np.zeros(n, dt)], want_d)
    want_y = oracle.block_df_adj(ops, [np.zeros(n, dt)], oracle.block_df(ops, [np.zeros(n, dt) for _ in range(nrow)], hm))
    try:
        for route in (1, 0):                                            # the under-aligned tall kernels / the general kernels (as before)
            J.tune(tall_unaligned=route)
            d = J.from_numpy(np.concatenate(hd), J.range(A))
            J.mul_(d, A, J.from_numpy(hm[0], J.domain(A)))
            assert_bits_equal(d.to_numpy(), np.concatenate(want_d), f"{name} {nrow} x {n} forward, tall_unaligned={route}")
            mt = J.mul_(J.rand(J.domain(A), seed=3, stream=3), A.H, d)
            assert_bits_equal(mt.to_numpy().ravel(order="F"), want_m[0], f"{name} {nrow} x {n} adjoint, tall_unaligned={route}")
    finally:
        J.tune(tall_unaligned=1)
    # the fused normal operator (the chain's bits: forward into zeros, then the adjoint)
    N = J.compose(A.H, A)
    y = J.mul_(J.rand(J.domain(A), seed=5, stream=5), N, J.from_numpy(hm[0], J.domain(A)))
    assert_bits_equal(y.to_numpy().ravel(order="F"), want_y[0], f"{name} {nrow} x {n} fused A'A")
    J.close(A)


@pytest.mark.parametrize("dt", [np.float32, np.complex64, np.float64])
def test_the_under_aligned_rows_write_nothing_outside_themselves(Jets, oracle, dt):
    """The operator's range is the MIDDLE of a longer slab (a view of blocks 1 .. nrow of nrow + 2): the blocks in front and behind keep their bits
    (the partial last pack of a row is stored scalar by scalar, never as a whole pack that would reach into the next block)."""
    import ctypes as C

    from jets_jl_amd._ffi import check, lib
    from jets_jl_amd.arrays import BlockArray

    J = Jets
    n, nrow = 3 * 1024 + 1, 9
    spc = J.JetSpace(dt, n)
    big = J.rand(J.JetBSpace([spc] * (nrow + 2)), seed=9, stream=1)
    before = big.to_numpy().copy()
    # the diagonals are blocks of ONE slab as well: their rows are off the 16-byte grid like the range vector's
    slab = J.rand(J.JetBSpace([spc] * nrow), seed=21, stream=4)
    A = J.blockop([[J.JopDiagonal(slab.arrays[i])] for i in range(nrow)])
    hs = slab.to_numpy()
    ops = [[oracle.Block("diag", n, coeff=hs[i * n:(i + 1) * n].copy())] for i in range(nrow)]
    hm = [u01(oracle, dt, 91, 0, n)]
    want_d = oracle.block_df(ops, [np.zeros(n, dt) for _ in range(nrow)], hm)
    h = C.c_void_p()
    check(lib.jh_bvec_view(big.handle, 1, nrow, C.byref(h)))
    view = BlockArray(h, [spc] * nrow, np.dtype(dt), owner=big)
    J.mul_(view, A, J.from_numpy(hm[0], J.domain(A)))
    got = big.to_numpy()
    assert_bits_equal(got[:n], before[:n], "the block in front of the rows")
    assert_bits_equal(got[(nrow + 1) * n:], before[(nrow + 1) * n:], "the block behind the rows")
    assert_bits_equal(got[n:(nrow + 1) * n], np.concatenate(want_d), "the rows")
    # and the adjoint reads exactly its rows: poison the neighbours, same result
    big[0:n] = np.full(n, np.nan, dt)
    big[(nrow + 1) * n:(nrow + 2) * n] = np.full(n, np.nan, dt)
    want_m = oracle.block_df_adj(ops, [np.zeros(n, dt)], want_d)
    mt = J.mul_(J.rand(J.domain(A), seed=3, stream=3), A.H, view)
    assert_bits_equal(mt.to_numpy().ravel(order="F"), want_m[0], "adjoint from the view")
    J.close(A)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("ncol,n", [(5, 1027), (40, 4099)])
def test_a_wide_operator_off_the_pack_grid_runs_on_its_tall_twin(Jets, oracle, dt, ncol, n):
    """1 x K of elementwise blocks: forward = the twin's ordered adjoint sum continued from what d holds (1024: no zeroing), adjoint = the twin's forward."""
    J = Jets
    names = ["diag", "diag_adj", "identity", "scale", "zero"]
    kinds = [[names[(2 * j + j // 3) % 5] for j in range(ncol)]]
    A, ops = _mixed_ops(J, oracle, dt, kinds, [n], [n] * ncol)
    hm = [u01(oracle, dt, 61, j, n) for j in range(ncol)]
    hd = [u01(oracle, dt, 62, 0, n)]
    want_d = oracle.block_df(ops, [hd[0].copy()], hm)
    want_m = oracle.block_df_adj(ops, [u01(oracle, dt, 63, j, n) for j in range(ncol)], want_d)
    try:
        for tw in (2, 0):
            J.tune(wide_twin=tw)
            d = J.from_numpy(hd[0], J.range(A))
            J.mul_(d, A, J.from_numpy(np.concatenate(hm), J.domain(A)))
            assert_bits_equal(d.to_numpy().ravel(order="F"), want_d[0], f"wide 1 x {ncol} of {n} forward, wide_twin={tw}")
            mt = J.from_numpy(np.concatenate([u01(oracle, dt, 63, j, n) for j in range(ncol)]), J.domain(A))
            J.mul_(mt, A.H, d)
            assert_bits_equal(mt.to_numpy(), np.concatenate(want_m), f"wide 1 x {ncol} of {n} adjoint, wide_twin={tw}")
    finally:
        J.tune(wide_twin=1)
    J.close(A)


@pytest.mark.parametrize("dt", [np.float32, np.float64, np.complex64])
def test_many_small_rows_off_the_pack_grid_take_the_split_walk_and_its_fold(Jets, oracle, dt):
    """300 rows of 1027 elements: fewer element tiles than CUs, so the adjoint sums row ranges into slabs and folds them (deterministic, tolerance parity
    like every split sum); with adj_split = 0 the ordered walk gives the oracle's bits.  The fold's last pack is partial too."""
    J = Jets
    n, nrow = 1027, 300
    kinds = [["diag"] if i % 7 else ["identity"] for i in range(nrow)]
    A, ops = _mixed_ops(J, oracle, dt, kinds, [n] * nrow, [n])
    hd = [u01(oracle, dt, 82, i, n) for i in range(nrow)]
    want_m = oracle.block_df_adj(ops, [np.zeros(n, dt)], hd)
    d = J.from_numpy(np.concatenate(hd), J.range(A))
    try:
        J.tune(adj_split=0)
        mt = J.mul_(J.rand(J.domain(A), seed=3, stream=3), A.H, d)
        assert_bits_equal(mt.to_numpy().ravel(order="F"), want_m[0], "ordered walk")
        J.tune(adj_split=-1)
        mt = J.mul_(J.rand(J.domain(A), seed=3, stream=3), A.H, d)
        got = mt.to_numpy().ravel(order="F")
        tol = 1e-5 if np.dtype(dt).itemsize in (4, 8) and np.dtype(dt) != np.float64 else 1e-12
        assert np.linalg.norm(got - want_m[0]) <= tol * np.linalg.norm(want_m[0])
        J.tune(adj_split=5)
        mt = J.mul_(J.rand(J.domain(A), seed=3, stream=3), A.H, d)
        got = mt.to_numpy().ravel(order="F")
        assert np.linalg.norm(got - want_m[0]) <= tol * np.linalg.norm(want_m[0])
    finally:
        J.tune(adj_split=-1)
    J.close(A)
