"""Block-SPARSE operators of dense children (block-diagonal, block-bidiagonal, arrow; children of one shape or ragged, some adjointed, some next to
diagonal / identity blocks) on the list route of late round 5: jh_blockop_create lists the dense children per direction and pass, k_gemv_rows_list /
k_gemv_cols_list run exactly those (no workgroup for a block pair without a dense child), their products land in ONE compact scratch vector whose pieces the
block table names, and the combine launch walks each line's step list instead of the whole block row / column (src/Jets.jl:1020-1024, 1045-1049 with zero
blocks skipped, 1022 / 1047).  The reference's dense child is LinearAlgebra's gemv (test/runtests.jl:27-33): tolerance parity, stated here; with the knob
dense_list_split = 0 the forward of un-adjointed children keeps the sequential loop's bits (= the oracle's)."""
import numpy as np
import pytest

from .helpers import DTYPES, assert_bits_equal, u01

pytestmark = pytest.mark.gpu


def _tol(dt):
    return 2e-6 if np.dtype(dt) in (np.dtype(np.float32), np.dtype(np.complex64)) else 1e-13


def _err(a, b):
    return float(np.linalg.norm(a.astype(np.complex128) - b.astype(np.complex128)) / max(np.linalg.norm(b.astype(np.complex128)), 1e-300))


def _build(J, oracle, dt, kinds, row_len, col_len, seed):
    dev, ora = [], []
    for i, row in enumerate(kinds):
        dr, orow = [], []
        for j, k in enumerate(row):
            nr, nc, st = row_len[i], col_len[j], 1000 * i + j
            if k in ("dense", "dense_adj"):
                shape = (nc, nr) if k == "dense_adj" else (nr, nc)          # block = B' : B is nc x nr
                hA = np.asfortranarray(u01(oracle, dt, seed, st, shape[0] * shape[1]).reshape(shape, order="F"))
                op = J.JopDense(J.from_numpy(hA))
                dr.append(op.H if k == "dense_adj" else op)
                orow.append(oracle.Block("dense", shape[0], shape[1], coeff=hA, adjoint=(k == "dense_adj")))
            elif k == "diag":
                assert nr == nc
                dr.append(J.JopDiagonal(J.rand(J.JetSpace(dt, nr), seed=seed, stream=st)))
                orow.append(oracle.Block("diag", nr, coeff=u01(oracle, dt, seed, st, nr)))
            elif k == "id":
                assert nr == nc
                dr.append(J.JopIdentity(J.JetSpace(dt, nr))); orow.append(oracle.Block("identity", nr))
            else:
                dr.append(J.JopZeroBlock(J.JetSpace(dt, nc), J.JetSpace(dt, nr))); orow.append(oracle.Block("zero", nr, nc))
        dev.append(dr); ora.append(orow)
    return J.blockop(dev), ora


def _pattern(name, M, rng):
    kinds = [["zero"] * M for _ in range(M)]
    for i in range(M):
        for j in range(M):
            if name == "blockdiag":
                on = i == j
            elif name == "bidiag":
                on = i == j or i == j + 1
            elif name == "arrow":
                on = i == j or i == 0 or j == 0
            else:                                                           # "mixed": dense diagonal, elementwise blocks and adjointed children around it
                on = i == j or rng.random() < 0.12
            if on:
                kinds[i][j] = "dense"
    if name == "mixed":
        for i in range(M):
            for j in range(M):
                if kinds[i][j] == "dense" and i != j:
                    kinds[i][j] = ["dense", "dense_adj", "diag", "id"][rng.integers(4)]
    return kinds


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("name,M,n", [("blockdiag", 24, 96), ("blockdiag", 7, 352), ("bidiag", 12, 128), ("arrow", 9, 72), ("mixed", 10, 104), ("blockdiag", 40, 20)])
def test_block_sparse_operators_of_dense_children_on_the_list_route(Jets, oracle, dt, name, M, n):
    J = Jets
    rng = np.random.default_rng(17 * M + n)
    kinds = _pattern(name, M, rng)
    A, ops = _build(J, oracle, dt, kinds, [n] * M, [n] * M, seed=300 + M)
    hm = [u01(oracle, dt, 41, j, n) for j in range(M)]
    hd = [u01(oracle, dt, 42, i, n) for i in range(M)]
    hmt = [u01(oracle, dt, 43, j, n) for j in range(M)]
    want_d = np.concatenate(oracle.block_df(ops, [b.copy() for b in hd], hm))           # forward into a dirty d
    want_m = np.concatenate(oracle.block_df_adj(ops, [b.copy() for b in hmt], [want_d[i * n:(i + 1) * n] for i in range(M)]))
    got = {}
    try:
        for route in ("lists", "lists-combine", "lists-in-order", "grid", "loop"):    # lists-combine: never the one-launch direct mode of block-diagonal operators
            J.tune(small_loop_max_kib=0, dense_list=0 if route == "grid" else 1, dense_list_split=0 if route == "lists-in-order" else 1,
                   dense_mixed=0 if route == "loop" else 1, small_loop=0 if route == "loop" else 1, dense_direct=0 if route == "lists-combine" else 1)
            m = J.from_numpy(np.concatenate(hm), J.domain(A))
            d = J.from_numpy(np.concatenate(hd), J.range(A))
            J.mul_(d, A, m)
            if route != "loop":
                assert 1 <= J.tune_get("last_launches") <= 3                             # the children's pass(es) + the combine, whatever M is
            if route == "lists" and name == "blockdiag":
                assert J.tune_get("last_launches") == 1                                  # every line holds one dense child: the children's launch writes d itself
            mt = J.from_numpy(np.concatenate(hmt), J.domain(A))
            J.mul_(mt, A.H, J.from_numpy(want_d, J.range(A)))
            got[route] = (d.to_numpy(), mt.to_numpy())
            assert _err(got[route][0], want_d) < _tol(dt), f"{name} {M} x {M} of {n}: forward, {route}"
            assert _err(got[route][1], want_m) < _tol(dt), f"{name} {M} x {M} of {n}: adjoint, {route}"
        # deterministic: the same bits on a second run
        J.tune(small_loop_max_kib=0, dense_list=1, dense_list_split=1, dense_mixed=1, small_loop=1)
        d = J.from_numpy(np.concatenate(hd), J.range(A))
        J.mul_(d, A, J.from_numpy(np.concatenate(hm), J.domain(A)))
        assert_bits_equal(d.to_numpy(), got["lists"][0], "second run of the list route")
        if name != "mixed":                                                              # only un-adjointed children: columns in order = the oracle's loop
            assert_bits_equal(got["lists-in-order"][0], want_d, f"{name}: forward with columns in order vs the oracle")
        assert_bits_equal(got["lists"][0], got["lists-combine"][0], "direct mode: the forward's bits with and without the combine launch")
        assert_bits_equal(got["lists"][1], got["lists-combine"][1], "direct mode: the adjoint's bits with and without the combine launch")
        # rows / columns without any block stay as the reference leaves them: d as found (1022), m zeroed (1042)
    finally:
        J.tune(small_loop_max_kib=512, dense_list=1, dense_list_split=1, dense_mixed=1, small_loop=1, dense_direct=1)
    J.close(A)


def test_ragged_children_and_lines_without_blocks(Jets, oracle):
    """Children of different shapes (one row chunk of the longest covers the others; lanes past a child's rows idle), a block row and a block column without
    any block, an EMPTY block row: forward leaves the blockless row of d as found, the adjoint zeroes the blockless column of m."""
    J, dt = Jets, np.float32
    row_len, col_len = [96, 40, 0, 200, 64], [64, 200, 96, 40]
    kinds = [["zero", "zero", "dense", "zero"],
             ["zero", "zero", "zero", "dense_adj"],
             ["zero", "zero", "zero", "zero"],
             ["zero", "dense", "zero", "zero"],
             ["zero", "zero", "zero", "zero"]]                                           # row 4 and column 0 hold no block
    A, ops = _build(J, oracle, dt, kinds, row_len, col_len, seed=77)
    hm = [u01(oracle, dt, 51, j, col_len[j]) for j in range(4)]
    hd = [u01(oracle, dt, 52, i, row_len[i]) for i in range(5)]
    hmt = [u01(oracle, dt, 53, j, col_len[j]) for j in range(4)]
    want_d = oracle.block_df(ops, [b.copy() for b in hd], hm)
    want_m = oracle.block_df_adj(ops, [b.copy() for b in hmt], want_d)
    try:
        J.tune(small_loop_max_kib=0)
        d = J.from_numpy(np.concatenate(hd), J.range(A))
        J.mul_(d, A, J.from_numpy(np.concatenate(hm), J.domain(A)))
        mt = J.from_numpy(np.concatenate(hmt), J.domain(A))
        J.mul_(mt, A.H, J.from_numpy(np.concatenate(want_d), J.range(A)))
    finally:
        J.tune(small_loop_max_kib=512)
    assert _err(d.to_numpy(), np.concatenate(want_d)) < 2e-6 and _err(mt.to_numpy(), np.concatenate(want_m)) < 2e-6
    assert_bits_equal(d.to_numpy()[-64:], hd[4], "the row without blocks keeps d as found")
    assert not mt.to_numpy()[:64].any(), "the column without blocks is zeroed (1042)"
    J.close(A)


def test_many_small_children_leave_the_one_launch_loop(Jets):
    """256 children of 64 x 64 on a block diagonal (4 MiB together): the batched list route by default (ONE launch: every line holds one dense child), the loop when pinned."""
    J = Jets
    n, M = 64, 256
    spc, mat = J.JetSpace(np.float32, n), J.JetSpace(np.float32, n, n)
    A = J.blockop([[J.JopDense(J.rand(mat, seed=9, stream=i)) if i == j else J.JopZeroBlock(spc, spc) for j in range(M)] for i in range(M)])
    m = J.rand(J.domain(A), seed=10, stream=0)
    d = J.mul_(J.zeros(J.range(A)), A, m)
    assert J.tune_get("last_launches") == 1
    try:
        J.tune(small_loop_max_kib=1 << 40)
        d2 = J.mul_(J.zeros(J.range(A)), A, m)
    finally:
        J.tune(small_loop_max_kib=512)
    assert _err(d.to_numpy(), d2.to_numpy()) < 2e-6
    J.close(A)
