"""GPU parity: tall block operators of DENSE children (the reference's JopBaz, test/runtests.jl:27-33; the tall-and-skinny test
set 720-742 is built from exactly these) through the BATCHED GEMV kernels of jh_dense.hip -- every child of the operator in
one launch instead of one tiny launch per child.

Bar: forward BIT-EXACT against the oracle's loop (each output row: columns in order, product rounded then added) while the
columns are not split, i.e. whenever the children alone fill the chip or a child is below 1 MiB; adjoint within rel-l2 1e-6
(Float32 / ComplexF32) or 1e-14 of an 80-bit host sum (it is a wave reduction in fp64 -- tolerance parity, like the per-child
kernel it replaces and like any BLAS).
"""
import numpy as np
import pytest

from .helpers import DTYPES, SEED_D, SEED_M, assert_bits_equal, u01

pytestmark = pytest.mark.gpu


def _tol(dt):
    return 1e-6 if np.dtype(dt) in (np.dtype(np.float32), np.dtype(np.complex64)) else 1e-14


def _err(a, b):
    a, b = np.asarray(a, dtype=np.clongdouble).ravel(), np.asarray(b, dtype=np.clongdouble).ravel()
    return float(np.linalg.norm(np.abs(a - b).astype(np.longdouble)) / np.linalg.norm(np.abs(b).astype(np.longdouble)))


def _tall_dense(Jets, oracle, dt, nchild, nr, nc, seed=900):
    mats, ora = [], []
    for z in range(nchild):
        hA = np.asfortranarray(u01(oracle, dt, seed, z, nr * nc).reshape((nr, nc), order="F"))
        mats.append(hA)
        ora.append([oracle.Block("dense", nr, nc, coeff=hA)])
    A = Jets.blockop([[Jets.JopDense(Jets.from_numpy(hA))] for hA in mats])
    return A, ora, mats


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("nchild,nr,nc", [(2, 8, 8), (3, 10, 10), (5, 64, 48), (40, 33, 7), (300, 16, 128), (7, 256, 512), (1100, 2, 4096)])
def test_batched_dense_children_forward_and_adjoint(Jets, oracle, dt, nchild, nr, nc):
    if nchild * nr * nc * np.dtype(dt).itemsize > (1 << 28):
        pytest.skip("kept small")
    A, ora, mats = _tall_dense(Jets, oracle, dt, nchild, nr, nc)
    assert Jets.nblocks_op(A) == (nchild, 1)
    m = Jets.rand(Jets.domain(A), seed=SEED_M, stream=0)
    hm = u01(oracle, dt, SEED_M, 0, nc)
    d = Jets.rand(Jets.range(A), seed=5, stream=5)                                # dirty: every row is overwritten (1026)
    Jets.mul_(d, A, m)
    ref_d = oracle.block_df(ora, [np.full(nr, 9, dtype=dt) for _ in range(nchild)], [hm])
    item = np.dtype(dt).itemsize
    lanes = -(-(nr * item) // 16)                                                 # 16-byte packs per column
    split = nr * nc * item >= (1 << 20) and -(-lanes // 256) * nchild < 512       # big children that do not fill the chip: columns split
    if split:
        assert _err(d.to_numpy(), np.concatenate(ref_d)) < _tol(dt)
    else:
        assert_bits_equal(d.to_numpy(), np.concatenate(ref_d), "A*m == [B1 m; B2 m; ...]  (test/runtests.jl:724)")
    dd = Jets.rand(Jets.range(A), seed=SEED_D, stream=0)
    hd = u01(oracle, dt, SEED_D, 0, nchild * nr).reshape(nchild, nr)
    mt = Jets.rand(Jets.domain(A), seed=6, stream=6)                              # dirty: zeroed first (1042)
    Jets.mul_(mt, A.H, dd)
    wide = np.clongdouble if np.iscomplexobj(mats[0]) else np.longdouble
    truth = sum(np.conj(mats[z].astype(wide)).T @ hd[z].astype(wide) for z in range(nchild))   # A'd == sum_i B_i' d_i  (:733)
    assert _err(mt.to_numpy().ravel(order="F"), truth) < _tol(dt)
    again = Jets.zeros(Jets.domain(A))
    Jets.mul_(again, A.H, dd)
    assert_bits_equal(again.to_numpy(), mt.to_numpy(), "batched dense adjoint, second run")
    lhs, rhs = Jets.dot_product_test(A, m, dd)
    assert abs(lhs - rhs) / abs(lhs + rhs) < (1e-5 if _tol(dt) > 1e-10 else 1e-12)


def test_batched_dense_large_children_split_columns(Jets, oracle):
    """Few big children: the columns are split over the grid until the chip is full (partial rows added in chunk order):
    tolerance parity for the forward, like the single-matrix kernel on a large matrix."""
    dt, nchild, nr, nc = np.float32, 3, 512, 2048                                 # 4 MiB per child
    A, ora, mats = _tall_dense(Jets, oracle, dt, nchild, nr, nc)
    m = Jets.rand(Jets.domain(A), seed=SEED_M, stream=0)
    hm = u01(oracle, dt, SEED_M, 0, nc)
    d = A * m
    truth = np.concatenate([mats[z].astype(np.float64) @ hm.astype(np.float64) for z in range(nchild)])
    assert _err(d.to_numpy(), truth) < 1e-6
    dd = Jets.rand(Jets.range(A), seed=SEED_D, stream=0)
    hd = u01(oracle, dt, SEED_D, 0, nchild * nr).reshape(nchild, nr)
    mt = A.H * dd
    truth_m = sum(mats[z].astype(np.float64).T @ hd[z].astype(np.float64) for z in range(nchild))
    assert _err(mt.to_numpy().ravel(order="F"), truth_m) < 1e-6


def test_adjointed_dense_children_keep_the_loop_and_ragged_ones_batch(Jets, oracle):
    """A child carrying the adjoint flag is outside the batched kernels: the reference's per-block loop runs (and still matches
    the oracle); children that differ only in their ROW count (shots with different trace counts) batch through the row table."""
    dt = np.float64
    hA = [np.asfortranarray(u01(oracle, dt, 901, z, 12 * 12).reshape((12, 12), order="F")) for z in range(3)]
    dev = [Jets.JopDense(Jets.from_numpy(a)) for a in hA]
    A = Jets.blockop([[dev[0]], [dev[1].H], [dev[2]]])
    ora = [[oracle.Block("dense", 12, 12, coeff=hA[0])], [oracle.Block("dense", 12, 12, coeff=hA[1], adjoint=True)], [oracle.Block("dense", 12, 12, coeff=hA[2])]]
    m = Jets.rand(Jets.domain(A), seed=SEED_M, stream=0)
    hm = u01(oracle, dt, SEED_M, 0, 12)
    ref = oracle.block_df(ora, [np.zeros(12, dtype=dt) for _ in range(3)], [hm])
    assert _err((A * m).to_numpy(), np.concatenate(ref)) < 1e-14
    hB = np.asfortranarray(u01(oracle, dt, 902, 0, 5 * 12).reshape((5, 12), order="F"))
    B = Jets.blockop([[dev[0]], [Jets.JopDense(Jets.from_numpy(hB))]])
    refB = oracle.block_df([[oracle.Block("dense", 12, 12, coeff=hA[0])], [oracle.Block("dense", 5, 12, coeff=hB)]],
                           [np.zeros(12, dtype=dt), np.zeros(5, dtype=dt)], [hm])
    assert_bits_equal((B * m).to_numpy(), np.concatenate(refB), "ragged dense children")


def _wide_dense(Jets, oracle, dt, nchild, nr, nc, seed=950):
    mats, orow = [], []
    for z in range(nchild):
        hA = np.asfortranarray(u01(oracle, dt, seed, z, nr * nc).reshape((nr, nc), order="F"))
        mats.append(hA)
        orow.append(oracle.Block("dense", nr, nc, coeff=hA))
    A = Jets.blockop([[Jets.JopDense(Jets.from_numpy(hA)) for hA in mats]])
    return A, [orow], mats


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("nchild,nr,nc", [(2, 8, 8), (3, 5, 5), (5, 64, 48), (40, 12, 7), (300, 16, 128), (7, 256, 512),
                                          (70, 100, 33), (130, 256, 40), (1000, 20, 7), (300, 66, 19), (2100, 128, 128)])   # round 4: the fused wide forward's shapes
def test_batched_dense_children_of_a_wide_operator(Jets, oracle, dt, nchild, nr, nc):
    """test/runtests.jl:744-758 (short-and-fat): A*m == B1 m1 + B2 m2 + ... accumulated into d AS FOUND (src/Jets.jl:1024),
    A'd == [B1'd; B2'd; ...].  Up to 64 children the forward keeps the reference's order and rounding (bit-exact while the
    columns are not split); beyond, an fp64 fold (tolerance parity).  The adjoint is a wave reduction (tolerance parity)."""
    A, ora, mats = _wide_dense(Jets, oracle, dt, nchild, nr, nc)
    assert Jets.nblocks_op(A) == (1, nchild)
    m = Jets.rand(Jets.domain(A), seed=SEED_M, stream=0)
    hm = u01(oracle, dt, SEED_M, 0, nchild * nc)
    mb = [hm[j * nc:(j + 1) * nc].copy() for j in range(nchild)]
    d = Jets.rand(Jets.range(A), seed=5, stream=5)                                # dirty: the reference adds into it
    hd0 = u01(oracle, dt, 5, 5, nr)
    Jets.mul_(d, A, m)
    item = np.dtype(dt).itemsize
    fused_fwd = nchild > 64 and (nr * item) % 16 == 0 and nr * item // 16 <= 64 and nr * nc * item < (1 << 20)
    assert Jets.tune_get("last_dense_fused") == (1 if fused_fwd else 0), "many small children: the one-kernel forward (round 4)"
    ref_d = oracle.block_df(ora, [hd0.copy()], mb)
    lanes = -(-(nr * item) // 16)
    split = nr * nc * item >= (1 << 20) and -(-lanes // 256) * nchild < 512
    if nchild <= 64 and not split:
        assert_bits_equal(d.to_numpy(), ref_d[0], "A*m == sum_j B_j m_j into d as found")
    else:
        assert _err(d.to_numpy(), ref_d[0]) < _tol(dt)
    dd = Jets.rand(Jets.range(A), seed=SEED_D, stream=0)
    hd = u01(oracle, dt, SEED_D, 0, nr)
    mt = Jets.rand(Jets.domain(A), seed=6, stream=6)                              # dirty: overwritten (1051)
    Jets.mul_(mt, A.H, dd)
    wide = np.clongdouble if np.iscomplexobj(mats[0]) else np.longdouble
    truth = np.concatenate([np.conj(mats[z].astype(wide)).T @ hd.astype(wide) for z in range(nchild)])
    assert _err(mt.to_numpy(), truth) < _tol(dt)
    lhs, rhs = Jets.dot_product_test(A, m, dd)
    assert abs(lhs - rhs) / abs(lhs + rhs) < (1e-5 if _tol(dt) > 1e-10 else 1e-12)


@pytest.fixture(params=["column-batches", "lists-in-order", "lists"])
def grid_route(request, Jets):
    """M x K grids of uniform dense children: one tall batch per block column (rounds 2-4; knob dense_grid = 1, read when the operator is created), or --
    the default since late round 5 -- the children's list in one launch + the combine (small operators: the one-launch loop), with columns in order
    (the sequential loop's bits) or the list kernel's own lane layout (tolerance parity)."""
    Jets.tune(dense_grid=1 if request.param == "column-batches" else 0, dense_list_split=1 if request.param == "lists" else 0)
    yield request.param
    Jets.tune(dense_grid=0, dense_list_split=1)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("nrow,ncol,nr,nc", [(2, 2, 5, 5), (3, 4, 10, 6), (6, 3, 64, 32), (40, 5, 16, 48), (5, 6, 160, 96)])
def test_batched_dense_children_of_a_grid_operator(Jets, oracle, dt, nrow, ncol, nr, nc, grid_route):
    """An M x K operator of uniform dense children: d_i = ((found + A_i1 m_1) + A_i2 m_2) + ... in the reference's order (src/Jets.jl:1020-1024) --
    bit-exact for children below 1 MiB while every child's columns are summed in order --, m_j = sum_i A_ij' d_i."""
    mats = [[np.asfortranarray(u01(oracle, dt, 970, 1000 * i + j, nr * nc).reshape((nr, nc), order="F")) for j in range(ncol)] for i in range(nrow)]
    A = Jets.blockop([[Jets.JopDense(Jets.from_numpy(mats[i][j])) for j in range(ncol)] for i in range(nrow)])
    ora = [[oracle.Block("dense", nr, nc, coeff=mats[i][j]) for j in range(ncol)] for i in range(nrow)]
    m = Jets.rand(Jets.domain(A), seed=SEED_M, stream=0)
    hm = u01(oracle, dt, SEED_M, 0, ncol * nc)
    mb = [hm[j * nc:(j + 1) * nc].copy() for j in range(ncol)]
    d = Jets.rand(Jets.range(A), seed=5, stream=5)
    hd0 = u01(oracle, dt, 5, 5, nrow * nr)
    Jets.mul_(d, A, m)
    ref_d = oracle.block_df(ora, [hd0[i * nr:(i + 1) * nr].copy() for i in range(nrow)], mb)
    if grid_route == "lists":
        assert _err(d.to_numpy(), np.concatenate(ref_d)) < _tol(dt), "grid of dense children, forward into d as found (lane layout of the list kernel)"
    else:
        assert_bits_equal(d.to_numpy(), np.concatenate(ref_d), "grid of dense children, forward into d as found")
    dd = Jets.rand(Jets.range(A), seed=SEED_D, stream=0)
    hd = u01(oracle, dt, SEED_D, 0, nrow * nr).reshape(nrow, nr)
    mt = Jets.rand(Jets.domain(A), seed=6, stream=6)
    Jets.mul_(mt, A.H, dd)
    wide = np.clongdouble if np.iscomplexobj(mats[0][0]) else np.longdouble
    truth = np.concatenate([sum(np.conj(mats[i][j].astype(wide)).T @ hd[i].astype(wide) for i in range(nrow)) for j in range(ncol)])
    assert _err(mt.to_numpy(), truth) < _tol(dt)
    lhs, rhs = Jets.dot_product_test(A, m, dd)
    assert abs(lhs - rhs) / abs(lhs + rhs) < (1e-5 if _tol(dt) > 1e-10 else 1e-12)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("rows,nc", [((8, 12, 4, 16), 8), ((5, 7, 3), 6), (tuple(4 * (1 + (k * 7) % 9) for k in range(300)), 64), (tuple(3 + (k * 5) % 11 for k in range(50)), 33)])
def test_batched_ragged_tall_dense_children(Jets, oracle, dt, rows, nc):
    """Tall operator of dense children with different row counts (the seismic case: shots with different numbers of traces) on the
    batched kernels through the row-offset table: forward bit-exact (every child writes its rows in one ordered column sweep),
    adjoint within tolerance; 16-byte aligned and unaligned row counts."""
    mats = [np.asfortranarray(u01(oracle, dt, 990, z, nr * nc).reshape((nr, nc), order="F")) for z, nr in enumerate(rows)]
    A = Jets.blockop([[Jets.JopDense(Jets.from_numpy(a))] for a in mats])
    ora = [[oracle.Block("dense", a.shape[0], nc, coeff=a)] for a in mats]
    m = Jets.rand(Jets.domain(A), seed=SEED_M, stream=0)
    hm = u01(oracle, dt, SEED_M, 0, nc)
    d = Jets.rand(Jets.range(A), seed=5, stream=5)
    Jets.mul_(d, A, m)
    ref = oracle.block_df(ora, [np.zeros(nr, dtype=dt) for nr in rows], [hm])
    assert_bits_equal(d.to_numpy(), np.concatenate(ref), "ragged tall dense forward")
    dd = Jets.rand(Jets.range(A), seed=SEED_D, stream=0)
    hd = u01(oracle, dt, SEED_D, 0, sum(rows))
    off = np.cumsum((0,) + tuple(rows))
    mt = Jets.rand(Jets.domain(A), seed=6, stream=6)
    Jets.mul_(mt, A.H, dd)
    wide = np.clongdouble if np.iscomplexobj(mats[0]) else np.longdouble
    truth = sum(np.conj(mats[z].astype(wide)).T @ hd[off[z]:off[z + 1]].astype(wide) for z in range(len(rows)))
    assert _err(mt.to_numpy().ravel(order="F"), truth) < _tol(dt)


# ---- round 3: BIG dense children in mixed company -- one batched launch per block column + one combine launch -----------------
def _mixed_operator(Jets, oracle, dt, layout, row_len, col_len, seed=970):
    """layout[i][j] in {"dense", "dense_adj", "diag", "id", "zero"}; dense children are row_len[i] x col_len[j] (an adjointed one is
    the adjoint of a stored col_len[j] x row_len[i] matrix), the others square."""
    dev_rows, ora_rows = [], []
    for i, row in enumerate(layout):
        drow, orow = [], []
        for j, kind in enumerate(row):
            nr, nc = row_len[i], col_len[j]
            if kind == "dense":
                hA = np.asfortranarray(u01(oracle, dt, seed, i * 16 + j, nr * nc).reshape((nr, nc), order="F"))
                drow.append(Jets.JopDense(Jets.from_numpy(hA)))
                orow.append(oracle.Block("dense", nr, nc, coeff=hA))
            elif kind == "dense_adj":                                   # the block is B' with B stored col_len x row_len
                hB = np.asfortranarray(u01(oracle, dt, seed, i * 16 + j, nr * nc).reshape((nc, nr), order="F"))
                drow.append(Jets.JopDense(Jets.from_numpy(hB)).H)
                orow.append(oracle.Block("dense", nc, nr, coeff=hB, adjoint=True))
            elif kind == "diag":
                assert nr == nc
                g = u01(oracle, dt, seed + 1, i * 16 + j, nr)
                drow.append(Jets.JopDiagonal(Jets.from_numpy(g)))
                orow.append(oracle.Block("diag", nr, coeff=g))
            elif kind == "id":
                assert nr == nc
                drow.append(Jets.JopIdentity(Jets.JetSpace(dt, nr)))
                orow.append(oracle.Block("identity", nr))
            else:
                drow.append(Jets.JopZeroBlock(Jets.JetSpace(dt, nc), Jets.JetSpace(dt, nr)))
                orow.append(oracle.Block("zero", nr, nc))
        dev_rows.append(drow)
        ora_rows.append(orow)
    return Jets.blockop(dev_rows), ora_rows


def _mixed_shapes(dt):
    """(layout, row lengths, column lengths) per shape.  n1 x n1 and n1 x n2 dense children are 370-512 KiB for every element type:
    beyond the one-launch loop's 256 KiB, far below the 8 MiB from which a FEW such children run child by child (column split: tolerance
    parity, see `few_big`)."""
    n1 = {4: 352, 8: 256, 16: 176}[np.dtype(dt).itemsize]
    n2, n3 = 3 * n1 // 4, n1 // 2
    return {
        "grid3x4": ([["dense", "zero", "diag", "dense"], ["zero", "zero", "zero", "zero"], ["id", "dense", "dense", "zero"]],
                    [n1, n3, n1], [n1, n2, n1, n2]),
        "tall": ([["dense"], ["diag"], ["dense"], ["zero"], ["id"]], [n2, n1, n1, n1, n1], [n1]),
        "wide": ([["dense", "diag", "zero", "dense", "id"]], [n1], [n2, n1, n1, n1, n1]),
        "ragged_dense_only": ([["dense", "dense"], ["dense", "dense"], ["dense", "dense"]], [n2, n1, n3], [n1, n2]),
        "odd_lengths": ([["id", "dense"], ["dense", "diag"]], [n1 - 3, n1 + 5], [n1 - 3, n1 + 5]),   # nothing 16-byte aligned: the scalar kernels
        "with_adjointed": ([["dense", "dense_adj", "diag"], ["dense_adj", "zero", "dense"], ["id", "dense", "dense_adj"]],
                           [n1, n2, n1], [n1, n2, n1]),                                          # both batched kernels in both directions: three launches
        "few_big": ([["dense", "diag"], ["id", "dense"]], [6 * n1, 6 * n1], [6 * n1, 6 * n1]),   # 17-19 MiB children, two of them: child by child
    }


MIXED_SHAPES = ["grid3x4", "tall", "wide", "ragged_dense_only", "odd_lengths", "with_adjointed", "few_big"]


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("shape", MIXED_SHAPES)
def test_big_dense_children_in_mixed_company(Jets, oracle, dt, shape):
    """Dense children too big for the one-launch loop (> 256 KiB) next to diagonal / identity / zero blocks, or of differing shapes:
    ONE batched GEMV launch leaves every dense child's product in a slab, ONE launch of the general kernel combines
    every output line in the reference's order (src/Jets.jl:1020-1024, 1042-1049).  Forward: the bits of the oracle's loops, into a
    dirty d (`_d .+=`, a row of zero blocks left as found); adjoint: a dirty m zeroed (or, with one block row, written directly,
    a zero block's column left as found), the dense terms from an fp64 wave reduction -> tolerance; never more than K + 1 launches."""
    J = Jets
    layout, row_len, col_len = _mixed_shapes(dt)[shape]
    A, ops = _mixed_operator(J, oracle, dt, layout, row_len, col_len)
    M, K = len(row_len), len(col_len)
    hm = [u01(oracle, dt, SEED_M, j, col_len[j]) for j in range(K)]
    hd = [u01(oracle, dt, SEED_D, i, row_len[i]) for i in range(M)]
    hmt = [u01(oracle, dt, SEED_D + 1, j, col_len[j]) for j in range(K)]
    want_d = oracle.block_df(ops, [b.copy() for b in hd], hm)
    want_m = oracle.block_df_adj(ops, [b.copy() for b in hmt], want_d)
    got = {}
    for knob in (1, 0, 2):                                             # the batched path with columns in order, the per-block loop it replaces, the batched path
        J.tune(dense_mixed=1 if knob else 0, dense_list_split=1 if knob == 2 else 0)   # with its own lane layout (late round 5: column groups per workgroup)
        try:
            m = J.from_numpy(np.concatenate(hm), J.domain(A)) if K > 1 else J.from_numpy(hm[0])
            d = J.from_numpy(np.concatenate(hd), J.range(A))            # dirty
            J.mul_(d, A, m)
            if knob:
                assert 1 <= J.tune_get("last_launches") <= (3 if shape == "with_adjointed" else 2), "forward: one batched launch (two with adjointed children) + the combine"
            mt = J.from_numpy(np.concatenate(hmt), J.domain(A)) if K > 1 else J.from_numpy(hmt[0])   # dirty
            J.mul_(mt, A.H, d)
            if knob:
                assert 1 <= J.tune_get("last_launches") <= (3 if shape == "with_adjointed" else 2), "adjoint: one batched launch (two with adjointed children) + the combine"
            got[knob] = (d.to_numpy(), mt.to_numpy().ravel(order="F"))
        finally:
            J.tune(dense_mixed=1, dense_list_split=1)
    # the default lane layout of the list kernels (column groups that meet in LDS): deterministic, tolerance parity -- like the column chunks of the per-child
    # kernel and like any BLAS (the reference's dense child is LinearAlgebra's gemv)
    assert _err(got[2][0], np.concatenate(want_d)) < _tol(dt), f"{shape}: forward, automatic lane layout"
    assert _err(got[2][1], np.concatenate(want_m)) < _tol(dt), f"{shape}: adjoint, automatic lane layout"
    if shape == "few_big":                                             # the per-child kernel of the loop splits a big child's columns: tolerance; since late
        assert _err(got[1][0], np.concatenate(want_d)) < _tol(dt)      # round 5 the batched route takes such children from its lists too (columns in order here)
        assert _err(got[0][0], np.concatenate(want_d)) < _tol(dt)
    elif shape == "with_adjointed":                                    # an adjointed child's forward is B' m: an fp64 wave reduction here and in the loop
        assert _err(got[1][0], np.concatenate(want_d)) < _tol(dt)
        assert _err(got[0][0], np.concatenate(want_d)) < _tol(dt)
    else:
        assert_bits_equal(got[1][0], np.concatenate(want_d), f"{shape}: forward vs the oracle's loop")
        assert_bits_equal(got[1][0], got[0][0], f"{shape}: forward vs the per-block loop")
    assert _err(got[1][1], np.concatenate(want_m)) < _tol(dt), f"{shape}: adjoint"
    assert _err(got[0][1], np.concatenate(want_m)) < _tol(dt)
    # what the reference leaves untouched stays untouched: a block row of zero blocks (forward), a zero block's column of a one-row operator
    if shape == "grid3x4":
        assert_bits_equal(got[1][0][row_len[0]:row_len[0] + row_len[1]], hd[1], "the row of zero blocks keeps d as found")
    if shape == "wide":
        lo = col_len[0] + col_len[1]
        assert_bits_equal(got[1][1][lo:lo + col_len[2]], hmt[2], "the zero block's column keeps m as found (1047 / 1051)")
    J.close(A)


# ---- round 4: the fused adjoint kernel (k_gemv_cols_fused) -----------------------------------------------------------------------
# shapes chosen for its cases: columns of P 16-byte packs with P a power of two (a batch is contiguous) and not (dead lanes), columns
# shorter than a wave load (several columns per unit), exactly one, and 2 / 4 / 8 / 16 wave loads long (the NPX instantiations and
# the per-unit pack of d), row counts that are not whole packs (the one-element-per-lane instantiation), column counts that leave
# the last batch ragged, child counts that leave the last group and the last wave quarter ragged
FUSED_SHAPES = [(37, 100, 33), (9, 256, 40), (130, 128, 128), (5, 512, 24), (3, 1024, 10), (3, 2048, 6), (2, 4096, 5), (1030, 64, 64),
                (11, 66, 19), (6, 257, 9), (65, 96, 50), (2, 256, 256), (4, 300, 7)]


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("nchild,nr,nc", FUSED_SHAPES)
def test_fused_adjoint_of_many_small_dense_children(Jets, oracle, dt, nchild, nr, nc):
    """m = sum_z B_z' d_z (src/Jets.jl:1045-1053 with JopBaz children, test/runtests.jl:27-33, 733) on the one-kernel route: within
    1e-6 / 1e-14 of an 80-bit host sum, the same bits on a second run (no atomics), and within tolerance of the three-launch route
    it replaces; `last_dense_fused` says which route ran."""
    A, ora, mats = _tall_dense(Jets, oracle, dt, nchild, nr, nc, seed=910)
    dd = Jets.rand(Jets.range(A), seed=SEED_D, stream=0)
    hd = u01(oracle, dt, SEED_D, 0, nchild * nr).reshape(nchild, nr)
    mt = Jets.rand(Jets.domain(A), seed=6, stream=6)                              # dirty: zeroed first (1042)
    Jets.mul_(mt, A.H, dd)
    item = np.dtype(dt).itemsize
    packs = nr * item / 16 if (nr * item) % 16 == 0 else nr                       # whole 16-byte packs per column, else one element per lane
    bsz = 8 if np.iscomplexobj(mats[0]) else 16                                   # units per batch; columns of bsz .. 256 packs
    expect_fused = bsz <= packs <= 256
    assert Jets.tune_get("last_dense_fused") == (1 if expect_fused else 0)
    wide = np.clongdouble if np.iscomplexobj(mats[0]) else np.longdouble
    truth = sum(np.conj(mats[z].astype(wide)).T @ hd[z].astype(wide) for z in range(nchild))
    assert _err(mt.to_numpy().ravel(order="F"), truth) < _tol(dt)
    again = Jets.rand(Jets.domain(A), seed=7, stream=7)
    Jets.mul_(again, A.H, dd)
    assert_bits_equal(again.to_numpy(), mt.to_numpy(), "fused dense adjoint, second run")
    for gw in (1, 3):                                                             # other groupings: other fp64 orders of the same rounded child sums
        Jets.tune(dense_gw=gw)
        try:
            Jets.mul_(again, A.H, dd)
        finally:
            Jets.tune(dense_gw=0)
        assert _err(again.to_numpy().ravel(order="F"), truth) < _tol(dt)
    Jets.tune(dense_fused=0)
    try:
        old = Jets.zeros(Jets.domain(A))
        Jets.mul_(old, A.H, dd)
        assert Jets.tune_get("last_dense_fused") == 0
    finally:
        Jets.tune(dense_fused=1)
    assert _err(old.to_numpy().ravel(order="F"), truth) < _tol(dt)
    assert _err(mt.to_numpy(), old.to_numpy()) < 2 * _tol(dt)
    m = Jets.rand(Jets.domain(A), seed=SEED_M, stream=0)
    lhs, rhs = Jets.dot_product_test(A, m, dd)
    assert abs(lhs - rhs) / abs(lhs + rhs) < (1e-5 if _tol(dt) > 1e-10 else 1e-12)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("nchild,nr,nc", [(37, 100, 33), (9, 256, 40), (5, 512, 24), (3, 2048, 6), (70, 64, 64), (11, 66, 19)])
def test_fused_adjoint_of_a_wide_operator_of_dense_children(Jets, oracle, dt, nchild, nr, nc):
    """m_j = B_j' d (src/Jets.jl:1051: written directly, test/runtests.jl:752-757) on the same kernel in its DIRECT form."""
    A, ora, mats = _wide_dense(Jets, oracle, dt, nchild, nr, nc, seed=960)
    dd = Jets.rand(Jets.range(A), seed=SEED_D, stream=0)
    hd = u01(oracle, dt, SEED_D, 0, nr)
    mt = Jets.rand(Jets.domain(A), seed=6, stream=6)                              # dirty: overwritten (1051)
    Jets.mul_(mt, A.H, dd)
    item = np.dtype(dt).itemsize
    packs = nr * item / 16 if (nr * item) % 16 == 0 else nr
    bsz = 8 if np.iscomplexobj(mats[0]) else 16
    assert Jets.tune_get("last_dense_fused") == (1 if bsz <= packs <= 256 else 0)
    wide = np.clongdouble if np.iscomplexobj(mats[0]) else np.longdouble
    truth = np.concatenate([np.conj(mats[z].astype(wide)).T @ hd.astype(wide) for z in range(nchild)])
    assert _err(mt.to_numpy(), truth) < _tol(dt)
    ref = oracle.block_df_adj(ora, [np.zeros(nc, dtype=dt) for _ in range(nchild)], [hd])
    assert _err(mt.to_numpy(), np.concatenate(ref)) < 2 * _tol(dt)
