"""GPU parity: the split-row walk of the adjoint-shaped kernels (tall operators with MANY rows of SMALL blocks).

The ordered walk (one thread per 16-byte vector of the domain, all rows in sequence: the reference's loop,
src/Jets.jl:1045-1053, bit for bit) leaves the chip idle when a block has few elements; from 256 rows on, when it would
launch fewer workgroups than the chip has CUs, the library cuts the rows into contiguous parts, sums each part in order
and folds the parts in a fixed order with fp64 accumulation (jh_tall.hip: pick_adj_parts, k_fold_parts).

Bar (stated): the split sum is DETERMINISTIC (same bits run to run) and within rel-l2 1e-6 (Float32 / ComplexF32) or
1e-14 (Float64 / ComplexF64) of the fp64 / exact-order-free truth -- tighter than the ordered Float32 sum itself, whose
rounding error grows with the row count; `tune(adj_split=0)` restores the ordered walk, BIT-EXACT against the oracle.
Everything that does not involve the cross-row sum stays bit-exact: the rows of u in the one-pass Golub-Kahan step.
"""
import ctypes as C

import numpy as np
import pytest

from .helpers import DTYPES, SEED_A, SEED_D, SEED_M, assert_bits_equal, u01

pytestmark = pytest.mark.gpu


def rel_err(a, b) -> float:                      # helpers.rel_err works in complex128; the truth here is wider
    a, b = np.asarray(a, dtype=np.clongdouble).ravel(), np.asarray(b, dtype=np.clongdouble).ravel()
    return float(np.linalg.norm(np.abs(a - b).astype(np.longdouble)) / np.linalg.norm(np.abs(b).astype(np.longdouble)))


def _tol(dt):
    return 1e-6 if np.dtype(dt) in (np.dtype(np.float32), np.dtype(np.complex64)) else 1e-14


def _slab_operator(Jets, oracle, dt, nrow, n, table=False):
    """nrow x 1 operator of diagonal blocks of n elements; coefficients in ONE slab (addressed by stride) or, with
    table=True, in separately allocated arrays (addressed through the device block table)."""
    spc = Jets.JetSpace(dt, n)
    if table:
        diags = [Jets.rand(spc, seed=SEED_A, stream=0, index_base=i * n) for i in range(nrow)]
    else:
        diags = Jets.rand(Jets.JetBSpace([spc] * nrow), seed=SEED_A, stream=0).arrays
    A = Jets.blockop([[Jets.JopDiagonal(g)] for g in diags])
    ha = u01(oracle, dt, SEED_A, 0, nrow * n).reshape(nrow, n)
    return A, ha, diags


def _native(Jets, A):
    from jets_jl_amd.jetblock import _tall_native

    return _tall_native(A)


@pytest.fixture()
def knob(Jets):
    saved = Jets.tune_get("adj_split")
    yield lambda v: Jets.tune(adj_split=v)
    Jets.tune(adj_split=saved)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("nrow,n,table", [(300, 1024, False), (1000, 520, False), (4096, 64, False), (257, 4096, True), (2049, 16, False)])
def test_split_adjoint_is_deterministic_and_accurate(Jets, oracle, knob, dt, nrow, n, table):
    A, ha, _keep = _slab_operator(Jets, oracle, dt, nrow, n, table)
    d = Jets.rand(Jets.range(A), seed=SEED_D, stream=0)
    hd = u01(oracle, dt, SEED_D, 0, nrow * n).reshape(nrow, n)
    wide = np.clongdouble if np.iscomplexobj(ha) else np.longdouble   # 64-bit mantissa on x86: a truth the Float64 sums can be measured against
    truth = np.sum(np.conj(ha.astype(wide)) * hd.astype(wide), axis=0)           # sum_i conj(a_i) .* d_i  (1045-1053), exact order-free
    knob(-1)
    mt = Jets.rand(Jets.domain(A), seed=99, stream=99)                            # dirty output: must be overwritten
    Jets.mul_(mt, A.H, d)
    assert Jets.tune_get("last_adj_parts") > 1, "the automatic policy should split this shape"
    got = mt.to_numpy().ravel(order="F")
    assert got.dtype == np.dtype(dt)
    assert rel_err(got, truth) < _tol(dt)
    again = Jets.zeros(Jets.domain(A))
    Jets.mul_(again, A.H, d)
    assert_bits_equal(again.to_numpy().ravel(order="F"), got, "split adjoint, second run")
    # the ordered walk stays available and bit-exact
    knob(0)
    Jets.mul_(mt, A.H, d)
    assert Jets.tune_get("last_adj_parts") == 1
    ops = [[oracle.Block("diag", n, coeff=ha[i].copy())] for i in range(nrow)]
    ref = oracle.block_df_adj(ops, [np.full(n, 7, dtype=dt)], [hd[i].copy() for i in range(nrow)])
    assert_bits_equal(mt.to_numpy().ravel(order="F"), ref[0], "ordered adjoint (adj_split=0)")
    assert rel_err(got, truth) <= 4 * rel_err(ref[0], truth) + _tol(dt) / 10       # never worse than the ordered sum


@pytest.mark.parametrize("dt", DTYPES)
def test_split_fused_normal_equals_split_chain(Jets, oracle, knob, dt):
    """(A' o A) m, src/Jets.jl:530-534: the fused kernel and forward-then-adjoint cut the rows identically, so they agree
    bit for bit in split mode as well; both within tolerance of the truth."""
    nrow, n = 600, 2048
    A, ha, _keep = _slab_operator(Jets, oracle, dt, nrow, n)
    m = Jets.rand(Jets.domain(A), seed=SEED_M, stream=0)
    hm = u01(oracle, dt, SEED_M, 0, n)
    knob(-1)
    y_fused = Jets.mul(A.H @ A, m)
    assert Jets.tune_get("last_adj_parts") > 1
    y_chain = A.H * (A * m)
    assert_bits_equal(y_fused.to_numpy().ravel(order="F"), y_chain.to_numpy().ravel(order="F"), "fused vs chained, split rows")
    wide = np.clongdouble if np.iscomplexobj(ha) else np.longdouble   # 64-bit mantissa on x86: a truth the Float64 sums can be measured against
    truth = np.sum(np.abs(ha.astype(wide)) ** 2, axis=0) * hm.astype(wide)
    assert rel_err(y_fused.to_numpy().ravel(order="F"), truth) < 4 * _tol(dt)     # d_i = a_i .* m is rounded to eltype first


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("beta", [-1.375, 0.0])
def test_split_bidiag_step(Jets, oracle, knob, dt, beta):
    """One-pass Golub-Kahan step on a many-row operator: u (row-wise) BIT-EXACT against the oracle's unfused chain,
    w = A'u within tolerance of the fp64 sum over the oracle's u, ||u||^2 within 1e-6 / 1e-13."""
    from jets_jl_amd._ffi import lib, check

    nrow, n, alpha = 777, 1024, 0.75
    A, ha, _keep = _slab_operator(Jets, oracle, dt, nrow, n)
    v = Jets.rand(Jets.domain(A), seed=51, stream=0)
    u = Jets.rand(Jets.range(A), seed=52, stream=0)
    w = Jets.rand(Jets.domain(A), seed=53, stream=0)
    hv, hu = u01(oracle, dt, 51, 0, n), u01(oracle, dt, 52, 0, nrow * n)
    ops = [[oracle.Block("diag", n, coeff=ha[i].copy())] for i in range(nrow)]
    tmp = oracle.block_df(ops, [np.zeros(n, dtype=dt) for _ in range(nrow)], [hv])
    hu_blocks = [hu[i * n:(i + 1) * n].copy() for i in range(nrow)]
    if beta != 0.0:
        ref_u = oracle.barr_lincomb([np.empty(n, dtype=dt) for _ in range(nrow)], [alpha, beta], [tmp, hu_blocks])
    else:
        ref_u = oracle.barr_lincomb([np.empty(n, dtype=dt) for _ in range(nrow)], [alpha], [tmp])
    knob(-1)
    out = C.c_double(0)
    check(lib.jh_blockop_bidiag_step(_native(Jets, A).handle, u.handle, v.handle, w.handle, alpha, beta, C.byref(out)))
    assert Jets.tune_get("last_adj_parts") > 1
    assert_bits_equal(u.to_numpy(), np.concatenate(ref_u), "u <- alpha*A v + beta*u (split rows)")
    wide = np.clongdouble if np.iscomplexobj(ha) else np.longdouble   # 64-bit mantissa on x86: a truth the Float64 sums can be measured against
    truth_w = np.sum(np.conj(ha.astype(wide)) * np.stack(ref_u).astype(wide), axis=0)
    assert rel_err(w.to_numpy().ravel(order="F"), truth_w) < _tol(dt)
    truth = float(np.sum(np.abs(np.concatenate(ref_u).astype(np.complex128)) ** 2))
    assert out.value == pytest.approx(truth, rel=1e-6 if _tol(dt) > 1e-10 else 1e-13)
    # same rows through the plain adjoint: the two kernels cut the rows alike here (same tiling for small blocks)
    w2 = Jets.zeros(Jets.domain(A))
    Jets.mul_(w2, A.H, u)
    assert rel_err(w2.to_numpy().ravel(order="F"), truth_w) < _tol(dt)


def test_explicit_parts_and_ranges(Jets, oracle, knob):
    """adj_split = k forces k parts on any operator (also below the automatic threshold); the ranged entry point of the
    multi-GPU pipeline (jh_blockop_mul_adj_range) covers the vector piece by piece in split mode too."""
    from jets_jl_amd._ffi import lib, check

    dt, nrow, n = np.float32, 50, 32768
    A, ha, _keep = _slab_operator(Jets, oracle, dt, nrow, n)
    d = Jets.rand(Jets.range(A), seed=SEED_D, stream=0)
    hd = u01(oracle, dt, SEED_D, 0, nrow * n).reshape(nrow, n)
    truth = np.sum(ha.astype(np.float64) * hd.astype(np.float64), axis=0)   # Float32 data: fp64 is wide enough
    knob(-1)
    mt = Jets.zeros(Jets.domain(A))
    Jets.mul_(mt, A.H, d)
    assert Jets.tune_get("last_adj_parts") == 1                                   # 50 rows: ordered by default
    for k in (2, 7, 25, 1000):
        knob(k)
        Jets.mul_(mt, A.H, d)
        assert Jets.tune_get("last_adj_parts") == {2: 2, 7: 7, 25: 25, 1000: 25}[k]  # capped at two rows per part
        assert rel_err(mt.to_numpy(), truth) < 1e-6
    knob(7)
    pieces = Jets.rand(Jets.domain(A), seed=82, stream=0)
    nat = _native(Jets, A)
    for lo, cnt in ((0, 16384), (16384, 8192), (24576, 4), (24580, n - 24580)):
        check(lib.jh_blockop_mul_adj_range(nat.handle, pieces.handle, d.handle, lo, cnt))
    assert rel_err(pieces.to_numpy(), truth) < 1e-6
    with pytest.raises(Jets.JetsHipError):
        Jets.tune(adj_split=-2)


def test_dot_product_test_and_lsqr_on_many_small_rows(Jets, oracle, knob):
    """src/Jets.jl:1211-1226 and the solver loop (docs/src/index.md:235-246) on 5000 rows of 256-element blocks."""
    dt, nrow, n = np.float32, 5000, 256
    A, ha, _keep = _slab_operator(Jets, oracle, dt, nrow, n)
    knob(-1)
    m = Jets.rand(Jets.domain(A), seed=SEED_M, stream=1)
    d = Jets.rand(Jets.range(A), seed=SEED_D, stream=1)
    lhs, rhs = Jets.dot_product_test(A, m, d)
    assert Jets.tune_get("last_adj_parts") > 1
    assert abs(lhs - rhs) / abs(lhs + rhs) < 1e-5
    x_true = Jets.rand(Jets.domain(A), seed=4, stream=0)
    b = A * x_true
    res = Jets.lsqr(A, b, atol=0.0, btol=0.0, conlim=0.0, maxiter=30, force_maxiter=True)
    err = (res.x - x_true).materialize()
    assert float(Jets.norm(err)) / float(Jets.norm(x_true)) < 1e-5


# ---------------------------------------------------------------------------------- general M x K kernels
def _general_case(Jets, oracle, dt, nrow, ncol, n, kind_of, seed=700):
    """nrow x ncol operator of n-element blocks, kind_of(i, j) in {diag, diag_adj, identity, scale, zero}; device + oracle twins."""
    spc = Jets.JetSpace(dt, n)
    dev, ora = [], []
    for i in range(nrow):
        dr, orow = [], []
        for j in range(ncol):
            k = kind_of(i, j)
            if k == "zero":
                dr.append(Jets.JopZeroBlock(spc, spc)); orow.append(oracle.Block("zero", n, n))
            elif k == "identity":
                dr.append(Jets.JopIdentity(spc)); orow.append(oracle.Block("identity", n))
            elif k == "scale":
                a = 0.75 - 0.001 * (i + j)
                dr.append(Jets.JopLn(dom=spc, rng=spc, df=Jets.constdiag_df, df_adj=Jets.constdiag_df_adj, s={"a": a}))
                orow.append(oracle.Block("scale", n, scale=a))
            else:
                stream = 100000 * i + j
                op = Jets.JopDiagonal(Jets.rand(spc, seed=seed, stream=stream))
                dr.append(op.H if k == "diag_adj" else op)
                orow.append(oracle.Block("diag", n, coeff=u01(oracle, dt, seed, stream, n), adjoint=(k == "diag_adj")))
        dev.append(dr); ora.append(orow)
    return Jets.blockop(dev), ora


def _truth_fwd(ora, d_found, m_blocks, accumulate):
    """sum over the blocks of every row in 80-bit arithmetic (zero blocks skipped; a row of zero blocks only keeps d)."""
    wide = np.clongdouble
    out = []
    for i, row in enumerate(ora):
        acc = d_found[i].astype(wide) if accumulate else np.zeros(len(d_found[i]), dtype=wide)
        hit = False
        for j, b in enumerate(row):
            if b.kind == "zero":
                continue
            hit = True
            x = m_blocks[j].astype(wide)
            if b.kind == "identity":
                acc = acc + x
            elif b.kind == "scale":
                acc = acc + wide(b.scale) * x
            else:
                c = b.coeff.astype(wide)
                acc = acc + (np.conj(c) if b.adjoint else c) * x
        out.append(acc if hit else d_found[i].astype(wide))
    return out


KINDS5 = ("diag", "identity", "scale", "zero", "diag_adj")


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("nrow,ncol,n,mix", [(1, 300, 64, False), (1, 1000, 20, True), (3, 260, 128, True), (2, 512, 7, True)])
def test_split_general_forward_many_block_columns(Jets, oracle, knob, dt, nrow, ncol, n, mix):
    """JetBlock_df! (src/Jets.jl:1010-1032) with hundreds of block columns of small blocks: the split walk accumulates into d as
    found (1024), skips zero blocks, leaves a block row of zero blocks only untouched (1022); adj_split=0 is bit-exact."""
    kind_of = (lambda i, j: "zero" if (i == 1 and mix) else KINDS5[(3 * i + j) % 5]) if mix else (lambda i, j: "diag")
    A, ora = _general_case(Jets, oracle, dt, nrow, ncol, n, kind_of)
    m = Jets.rand(Jets.domain(A), seed=SEED_M, stream=0)
    hm = u01(oracle, dt, SEED_M, 0, ncol * n)
    mb = [hm[j * n:(j + 1) * n].copy() for j in range(ncol)]
    hd = u01(oracle, dt, SEED_D, 0, nrow * n)
    db = [hd[i * n:(i + 1) * n].copy() for i in range(nrow)]
    truth = np.concatenate(_truth_fwd(ora, db, mb, accumulate=True))
    knob(-1)
    d = Jets.rand(Jets.range(A), seed=SEED_D, stream=0)                           # dirty: the reference adds into it
    Jets.mul_(d, A, m)
    assert Jets.tune_get("last_adj_parts") > 1
    got = d.to_numpy()
    assert rel_err(got, truth) < 4 * _tol(dt)
    if mix and nrow > 1:
        assert_bits_equal(got[n:2 * n], hd[n:2 * n], "block row of zero blocks only stays as found")
    d2 = Jets.rand(Jets.range(A), seed=SEED_D, stream=0)
    Jets.mul_(d2, A, m)
    assert_bits_equal(d2.to_numpy(), got, "split forward, second run")
    knob(0)
    d3 = Jets.rand(Jets.range(A), seed=SEED_D, stream=0)
    Jets.mul_(d3, A, m)
    assert Jets.tune_get("last_adj_parts") == 1
    ref = oracle.block_df(ora, [b.copy() for b in db], mb)
    assert_bits_equal(d3.to_numpy(), np.concatenate(ref), "ordered forward (adj_split=0)")


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("nrow,ncol,n", [(700, 1, 100), (300, 3, 64), (2000, 2, 9)])
def test_split_general_adjoint_many_block_rows(Jets, oracle, knob, dt, nrow, ncol, n):
    """JetBlock_df'! (1034-1057) of a tall MIXED operator (diag / identity / scale / zero / diag' rows -- not the all-diagonal
    fast path): m zeroed, zero blocks skipped, rows summed per part and folded."""
    kind_of = lambda i, j: KINDS5[(i + 2 * j) % 5]
    A, ora = _general_case(Jets, oracle, dt, nrow, ncol, n, kind_of)
    d = Jets.rand(Jets.range(A), seed=SEED_D, stream=0)
    hd = u01(oracle, dt, SEED_D, 0, nrow * n)
    db = [hd[i * n:(i + 1) * n].copy() for i in range(nrow)]
    knob(-1)
    mt = Jets.rand(Jets.domain(A), seed=99, stream=9)                             # dirty: must be overwritten (1042)
    Jets.mul_(mt, A.H, d)
    assert Jets.tune_get("last_adj_parts") > 1
    got = mt.to_numpy().ravel(order="F")
    # truth: the adjoint is the forward of the transposed block matrix with every child adjointed
    oraT = [[oracle.Block(b.kind, n, n, coeff=b.coeff, scale=np.conj(b.scale), adjoint=not b.adjoint) if b.kind == "diag" else
             oracle.Block(b.kind, n, n, scale=np.conj(b.scale)) for b in (ora[i][j] for i in range(nrow))] for j in range(ncol)]
    truth = np.concatenate(_truth_fwd(oraT, [np.zeros(n, dtype=dt)] * ncol, db, accumulate=False))
    assert rel_err(got, truth) < 4 * _tol(dt)
    knob(0)
    Jets.mul_(mt, A.H, d)
    ref = oracle.block_df_adj(ora, [np.full(n, 3, dtype=dt) for _ in range(ncol)], db)
    assert_bits_equal(mt.to_numpy().ravel(order="F"), np.concatenate(ref), "ordered adjoint (adj_split=0)")


def test_split_general_nonlinear_forward(Jets, oracle, knob):
    """JetBlock_f! (988-1008) through the split walk: no zero-block skip -- every child's output, a zero block's zeros included,
    is added into d as found (1001)."""
    dt, nrow, ncol, n = np.float64, 2, 300, 32
    spc = Jets.JetSpace(dt, n)
    rows, ora = [], []
    for i in range(nrow):
        r, o = [], []
        for j in range(ncol):
            if (i + j) % 3 == 0:
                r.append(Jets.JopSquare(spc)); o.append(oracle.Block("square", n))
            elif (i + j) % 3 == 1:
                g = Jets.rand(spc, seed=31, stream=1000 * i + j)
                r.append(Jets.JopDiagonal(g)); o.append(oracle.Block("diag", n, coeff=u01(oracle, dt, 31, 1000 * i + j, n)))
            else:
                r.append(Jets.JopZeroBlock(spc, spc)); o.append(oracle.Block("zero", n, n))
        rows.append(r); ora.append(o)
    F = Jets.blockop(rows)
    m = Jets.rand(Jets.domain(F), seed=SEED_M, stream=0)
    hm = u01(oracle, dt, SEED_M, 0, ncol * n)
    mb = [hm[j * n:(j + 1) * n].copy() for j in range(ncol)]
    hd = u01(oracle, dt, SEED_D, 0, nrow * n)
    knob(0)
    d0 = Jets.rand(Jets.range(F), seed=SEED_D, stream=0)
    Jets.mul_(d0, F, m)
    ref = oracle.block_f(ora, [hd[i * n:(i + 1) * n].copy() for i in range(nrow)], mb)
    assert_bits_equal(d0.to_numpy(), np.concatenate(ref), "ordered JetBlock_f!")
    knob(-1)
    d1 = Jets.rand(Jets.range(F), seed=SEED_D, stream=0)
    Jets.mul_(d1, F, m)
    assert Jets.tune_get("last_adj_parts") > 1
    assert rel_err(d1.to_numpy(), np.concatenate(ref)) < 1e-13


# ---------------------------------------------------------------------------------- the kernels without a split variant of their own
@pytest.mark.parametrize("dt", DTYPES)
def test_split_fused_adjoint_update(Jets, oracle, knob, dt):
    """jh_blockop_mul_adj_axpby (the adjoint half of an LSQR / CGLS iteration) on many small rows: the row sum goes through
    the split walk of the plain adjoint, the axpby + norm through a small epilogue; adj_split=0 keeps the fused ordered kernel."""
    from jets_jl_amd._ffi import lib, check

    nrow, n = 900, 512
    A, ha, _keep = _slab_operator(Jets, oracle, dt, nrow, n)
    d = Jets.rand(Jets.range(A), seed=SEED_D, stream=0)
    hd = u01(oracle, dt, SEED_D, 0, nrow * n).reshape(nrow, n)
    hm = u01(oracle, dt, 41, 0, n)
    alpha, beta, gamma = 0.75, -1.375, 0.5
    wide = np.clongdouble if np.iscomplexobj(ha) else np.longdouble
    truth = wide(alpha) * np.sum(np.conj(ha.astype(wide)) * (wide(gamma) * hd.astype(wide)), axis=0) + wide(beta) * hm.astype(wide)
    out = C.c_double(0)
    res = {}
    for split in (-1, 0):
        knob(split)
        m = Jets.rand(Jets.domain(A), seed=41, stream=0)
        check(lib.jh_blockop_mul_adj_axpby(_native(Jets, A).handle, m.handle, d.handle, alpha, beta, gamma, C.byref(out)))
        got = m.to_numpy().ravel(order="F")
        assert rel_err(got, truth) < (4 * _tol(dt) if split else 5e-5 if _tol(dt) > 1e-10 else 1e-13)
        assert out.value == pytest.approx(float(np.sum(np.abs(got.astype(np.complex128)) ** 2)), rel=1e-6 if _tol(dt) > 1e-10 else 1e-13)
        res[split] = got
    assert Jets.tune_get("last_adj_parts") == 1                                   # the last call (adj_split=0) was the ordered kernel
    assert rel_err(res[-1], res[0]) < (5e-5 if _tol(dt) > 1e-10 else 1e-13)


@pytest.mark.parametrize("dt", DTYPES)
def test_split_jetsum_adjoint(Jets, oracle, knob, dt):
    """(1.5*A1 - 2.0*A2)' d, src/Jets.jl:648-655, on many small rows: every term's row sum through the split walk."""
    nrow, n = 800, 256
    spc = Jets.JetSpace(dt, n)
    g1 = Jets.rand(Jets.JetBSpace([spc] * nrow), seed=SEED_A, stream=0).arrays
    g2 = Jets.rand(Jets.JetBSpace([spc] * nrow), seed=77, stream=0).arrays
    A1 = Jets.blockop([[Jets.JopDiagonal(g)] for g in g1])
    A2 = Jets.blockop([[Jets.JopDiagonal(g)] for g in g2])
    h1 = u01(oracle, dt, SEED_A, 0, nrow * n).reshape(nrow, n)
    h2 = u01(oracle, dt, 77, 0, nrow * n).reshape(nrow, n)
    S = 1.5 * A1 - 2.0 * A2
    d = Jets.rand(Jets.range(A1), seed=SEED_D, stream=0)
    hd = u01(oracle, dt, SEED_D, 0, nrow * n).reshape(nrow, n)
    wide = np.clongdouble if np.iscomplexobj(h1) else np.longdouble
    truth = 1.5 * np.sum(np.conj(h1.astype(wide)) * hd.astype(wide), axis=0) - 2.0 * np.sum(np.conj(h2.astype(wide)) * hd.astype(wide), axis=0)
    knob(-1)
    got = (S.H * d).to_numpy().ravel(order="F")
    assert Jets.tune_get("last_adj_parts") > 1
    # the two terms nearly cancel (U[0,1) coefficients): measure against the size of the terms, not of the difference
    scale = float(np.linalg.norm(np.abs(1.5 * np.sum(np.conj(h1.astype(wide)) * hd.astype(wide), axis=0)).astype(np.longdouble)))
    err = float(np.linalg.norm(np.abs(got.astype(wide) - truth).astype(np.longdouble))) / scale
    assert err < 4 * _tol(dt)
    knob(0)
    ordered = (S.H * d).to_numpy().ravel(order="F")
    err0 = float(np.linalg.norm(np.abs(ordered.astype(wide) - truth).astype(np.longdouble))) / scale
    assert err <= 4 * err0 + _tol(dt)


# ---------------------------------------------------------------------------------- round 6: the temporary behind the slabs (round-5 advisor finding)
@pytest.mark.parametrize("dt,nrow,n", [(np.float32, 1024, 40001), (np.float32, 300, 10001), (np.complex64, 512, 2051), (np.float64, 288, 9999),
                                       (np.float32, 512, 8192 + 4)])
def test_fused_adjoint_update_and_jetsum_adjoint_split_walk_off_the_pack_grid(Jets, oracle, knob, dt, nrow, n):
    """Off-grid all-diagonal operators of >= 256 rows of mid-sized blocks: jh_blockop_mul_adj_axpby and jh_blocksum_mul_adj run "split adjoint into a
    temporary behind the slabs + epilogue" (jh_tall.hip: split_adjoint_tmp), and the adjoint they call is the MIXED one, whose launch shape cuts the rows
    into MORE parts than the all-diagonal shape the reservation used to be sized with (1024 rows of 40001 Float32: 52 slabs reserved, 64 used -- the last
    slabs lay over the temporary, or the scratch was regrown under it).  Stated bar as for every split sum: deterministic, rel-l2 1e-6 / 1e-14 of the truth."""
    from jets_jl_amd._ffi import check, lib

    J = Jets
    A, ha, _keep = _slab_operator(J, oracle, dt, nrow, n)
    B, hb, _keep2 = _slab_operator(J, oracle, dt, nrow, n, table=True)
    hd = u01(oracle, dt, SEED_D, 0, nrow * n).reshape(nrow, n)
    d = J.rand(J.range(A), seed=SEED_D, stream=0)
    wide = np.clongdouble if np.iscomplexobj(ha) else np.longdouble
    tA = np.sum(np.conj(ha.astype(wide)) * hd.astype(wide), axis=0)
    tB = np.sum(np.conj(hb.astype(wide)) * hd.astype(wide), axis=0)
    knob(-1)
    natA, natB = _native(J, A), _native(J, B)
    # m <- alpha A'(gamma d) + beta m with ||m||^2
    alpha, beta, gamma = 0.75, -1.5, 1.0
    hm = u01(oracle, dt, SEED_M, 0, n)
    m = J.from_numpy(hm, J.domain(A))
    nrm = C.c_double(0)
    check(lib.jh_blockop_mul_adj_axpby(natA.handle, m.handle, d.handle, alpha, beta, gamma, C.byref(nrm)))
    assert J.tune_get("last_adj_parts") > 1, "this shape should take the split walk"
    got = m.to_numpy().ravel(order="F")
    truth = alpha * tA + beta * hm.astype(wide)
    assert rel_err(got, truth) < 4 * _tol(dt)
    m2 = J.from_numpy(hm, J.domain(A))
    check(lib.jh_blockop_mul_adj_axpby(natA.handle, m2.handle, d.handle, alpha, beta, gamma, C.byref(nrm)))
    assert_bits_equal(m2.to_numpy().ravel(order="F"), got, "fused adjoint update through the split walk, second run")
    assert nrm.value == pytest.approx(float(np.sum(np.abs(got.astype(np.complex128)) ** 2)), rel=1e-5)
    # m = A'd - 0.5 B'd
    hs = (C.c_void_p * 2)(natA.handle, natB.handle)
    sc = (C.c_double * 2)(1.0, 0.5)
    fg = (C.c_int32 * 2)(0, 0)
    sg = (C.c_double * 2)(1.0, -1.0)
    ms = J.rand(J.domain(A), seed=7, stream=7)
    check(lib.jh_blocksum_mul_adj_typed(2, hs, sc, fg, sg, ms.handle, d.handle))
    gs = ms.to_numpy().ravel(order="F")
    assert rel_err(gs, tA - 0.5 * tB) < 8 * _tol(dt)
    ms2 = J.rand(J.domain(A), seed=8, stream=8)
    check(lib.jh_blocksum_mul_adj_typed(2, hs, sc, fg, sg, ms2.handle, d.handle))
    assert_bits_equal(ms2.to_numpy().ravel(order="F"), gs, "JetSum adjoint through the split walk, second run")
    J.close(A)
    J.close(B)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("nrow,n", [(3, 70), (9, 1027), (33, 4096), (17, 20001)])
def test_mixed_adjoint_on_the_chain_kernel_and_on_the_tall_kernel_agree_bit_for_bit(Jets, oracle, dt, nrow, n):
    """Round 6: the adjoint of a tall operator with rows of several kinds (or rows off the 16-byte grid) of up to 2 MiB runs the chain kernel with empty
    stage lists (jh_tall_chain.hip: bare_chain_adjoint; knob adj_bare_chain); with the knob off it is k_tall_diag_adj<MIXED> as in rounds 2-5.  Both sum
    the rows in order from +0: the same bits, and the oracle's (src/Jets.jl:1042-1049).  A dirty output, zero / identity / scalar / adjointed rows."""
    from .test_gpu_blockop import _mixed_ops

    J = Jets
    names = ["diag", "diag_adj", "identity", "scale", "zero"]
    kinds = [[names[(3 * i + i // 5) % 5]] for i in range(nrow)]
    A, ora = _mixed_ops(J, oracle, dt, kinds, [n] * nrow, [n], seed=33)
    hd = [u01(oracle, dt, 92, i, n) for i in range(nrow)]
    d = J.from_numpy(np.concatenate(hd), J.range(A))
    want = oracle.block_df_adj(ora, [np.full(n, 7, dt)], hd)[0]
    got = {}
    for knob in (1, 0):
        J.tune(adj_bare_chain=knob)
        try:
            got[knob] = J.mul_(J.rand(J.domain(A), seed=5, stream=knob), A.H, d).to_numpy().ravel(order="F")
        finally:
            J.tune(adj_bare_chain=1)
        assert_bits_equal(got[knob], want, f"adj_bare_chain={knob}: adjoint vs the oracle")
    J.close(A)
