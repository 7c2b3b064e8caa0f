"""GPU parity: JetBlock_df! / JetBlock_df'! / fused A'oA on the MI355X vs the CPU oracle.

The HIP path is driven through the product package (-> ctypes -> C ABI of include/jetship.h).
Re-encodes on seeded inputs: test/runtests.jl:720-742 (tall-and-skinny), 744-758 (short-and-fat),
622-695 (mixed 3x4 with zero blocks, dirty-output adjoint 684), 704-718 (singleton), 901-918
(dot product test), BASELINE.json configs[0] (4x4 identity, Float64, n = 128).
Bar: BIT-EXACT for the forward, the one-GPU adjoint (ordered mul-then-add) and the fused normal
operator; dot-product test |lhs-rhs|/|lhs+rhs| < 1e-5 (Float32) / 1e-12 (Float64).
"""
import itertools

import numpy as np
import pytest

from .helpers import DTYPES, SEED_D, SEED_M, assert_bits_equal, dev_blocks_to_numpy, make_tall_diag, u01

pytestmark = pytest.mark.gpu


def _dpt_tol(dt):
    return 1e-5 if np.dtype(dt) in (np.dtype(np.float32), np.dtype(np.complex64)) else 1e-12


# ---------------------------------------------------------------------------------- tall fast path
@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("nrow,shape", [(3, (8, 4, 4)), (5, (1024,)), (17, (32, 33, 4)), (64, (16, 16, 16)), (2, (4,))])
def test_tall_diag_forward_adjoint_bit_exact(Jets, oracle, dt, nrow, shape):
    A, _, ops, _ = make_tall_diag(Jets, oracle, dt, nrow, shape)
    n = int(np.prod(shape))
    assert Jets.nblocks_op(A) == (nrow, 1)                                        # test/runtests.jl:739
    assert isinstance(Jets.domain(A), Jets.JetSpace)                              # src/Jets.jl:927
    m = Jets.rand(Jets.domain(A), seed=SEED_M, stream=0)
    hm = u01(oracle, dt, SEED_M, 0, n)
    d = A * m                                                                     # zeros(range) then mul!  (:399)
    ref_d = oracle.block_df(ops, [np.zeros(n, dtype=dt) for _ in range(nrow)], [hm])
    assert_bits_equal(d.to_numpy(), np.concatenate(ref_d), "A*m")
    for i in (0, nrow - 1):                                                       # A*m == [B1 m; B2 m; ...]  (:724)
        assert_bits_equal(Jets.getblock(d, i).to_numpy().ravel(order="F"), ref_d[i], f"block {i}")

    dd = Jets.rand(Jets.range(A), seed=SEED_D, stream=0)
    hd = u01(oracle, dt, SEED_D, 0, nrow * n)
    hd_blocks = [hd[i * n:(i + 1) * n].copy() for i in range(nrow)]
    mt = Jets.rand(Jets.domain(A), seed=99, stream=99)                            # dirty output: must be overwritten (:684)
    Jets.mul_(mt, A.H, dd)
    ref_m = oracle.block_df_adj(ops, [np.full(n, 7, dtype=dt)], hd_blocks)
    assert_bits_equal(mt.to_numpy().ravel(order="F"), ref_m[0], "A'*d")


@pytest.mark.parametrize("knobs", [
    dict(fwd_group=0, fwd_unroll=0, fwd_wg=0, fwd_order=-1, adj_unroll=0, adj_depth=0, adj_wg=0, nt=1),      # automatic shapes
    dict(fwd_group=1, fwd_unroll=1, fwd_wg=256, fwd_order=1, adj_unroll=1, adj_depth=1, adj_wg=256, nt=0),
    dict(fwd_group=4, fwd_unroll=2, fwd_wg=256, adj_unroll=2, adj_depth=2, adj_wg=512, nt=1),
    dict(fwd_group=2, fwd_unroll=1, fwd_wg=512, fwd_order=1, adj_unroll=4, adj_depth=4, adj_wg=1024, nt=1),  # the 1024 x 256^3 shapes
    dict(fwd_group=7, fwd_unroll=8, fwd_wg=512, fwd_order=1, adj_unroll=1, adj_depth=8, adj_wg=512, nt=1),
    dict(fwd_group=64, fwd_unroll=4, fwd_wg=1024, adj_unroll=4, adj_depth=8, adj_wg=256, nt=0),
    dict(fwd_group=3, fwd_unroll=2, fwd_wg=512, fwd_order=0, adj_unroll=2, adj_depth=8, adj_wg=1024, nt=1),
])
def test_every_kernel_shape_gives_identical_bits(Jets, oracle, knobs):
    """The tuning knobs change tiling only -- never results."""
    saved = {k: Jets.tune_get(k) for k in knobs}
    try:
        Jets.tune(**knobs)
        dt, nrow, shape = np.float32, 13, (40, 40, 12)          # 19200 elements: partial tiles for every tiling
        A, _, ops, _ = make_tall_diag(Jets, oracle, dt, nrow, shape)
        n = int(np.prod(shape))
        m = Jets.rand(Jets.domain(A), seed=SEED_M, stream=0)
        hm = u01(oracle, dt, SEED_M, 0, n)
        d = A * m
        ref_d = oracle.block_df(ops, [np.zeros(n, dtype=dt) for _ in range(nrow)], [hm])
        assert_bits_equal(d.to_numpy(), np.concatenate(ref_d), f"forward {knobs}")
        mt = A.H * d
        ref_m = oracle.block_df_adj(ops, [np.zeros(n, dtype=dt)], ref_d)
        assert_bits_equal(mt.to_numpy().ravel(order="F"), ref_m[0], f"adjoint {knobs}")
        y = Jets.mul(A.H @ A, m)
        assert_bits_equal(y.to_numpy().ravel(order="F"), ref_m[0], f"fused normal {knobs}")
    finally:
        Jets.tune(**saved)


@pytest.mark.parametrize("dt", DTYPES)
def test_fused_normal_equals_unfused_and_oracle(Jets, oracle, dt):
    """JetComposite (A' o A): src/Jets.jl:530-534.  Fused kernel == chained kernels == oracle, bitwise."""
    nrow, shape = 9, (12, 8, 4)
    A, _, ops, _ = make_tall_diag(Jets, oracle, dt, nrow, shape)
    n = int(np.prod(shape))
    m = Jets.rand(Jets.domain(A), seed=SEED_M, stream=5)
    hm = u01(oracle, dt, SEED_M, 5, n)
    C = A.H @ A
    assert len(Jets.state(C)["ops"]) == 2
    y_fused = C * m
    y_chain = A.H * (A * m)
    ref = oracle.normal_df(ops, [np.zeros(n, dtype=dt)], [hm])
    assert_bits_equal(y_fused.to_numpy().ravel(order="F"), ref[0], "fused A'A vs oracle")
    assert_bits_equal(y_chain.to_numpy().ravel(order="F"), ref[0], "chained A'A vs oracle")
    ya = C.H * m                                                                  # composite adjoint (self-adjoint chain)
    assert_bits_equal(ya.to_numpy().ravel(order="F"), ref[0], "(A'A)' m")


@pytest.mark.parametrize("dt", [np.float32, np.float64, np.complex64, np.complex128])
def test_dot_product_test_tall(Jets, oracle, dt):
    """src/Jets.jl:1211-1226 with and without masks (test/runtests.jl:901-918)."""
    nrow, shape = 6, (16, 8, 8)
    A, _, ops, _ = make_tall_diag(Jets, oracle, dt, nrow, shape)
    m = Jets.rand(Jets.domain(A), seed=SEED_M, stream=1)
    d = Jets.rand(Jets.range(A), seed=SEED_D, stream=1)
    lhs, rhs = Jets.dot_product_test(A, m, d)
    assert abs(lhs - rhs) / abs(lhs + rhs) < _dpt_tol(dt)
    mmask, dmask = Jets.ones(Jets.domain(A)), Jets.ones(Jets.range(A))
    Jets.getblock(dmask, 0).assign(0.0)
    Jets.setblock_(dmask, 2, 0.0)
    lhs2, rhs2 = Jets.dot_product_test(A, m, d, mmask=mmask, dmask=dmask)
    assert abs(lhs2 - rhs2) / abs(lhs2 + rhs2) < _dpt_tol(dt)
    assert abs(lhs2) < abs(lhs)
    if np.dtype(dt).kind == "c":
        assert np.iscomplexobj(lhs) and np.iscomplexobj(rhs)                      # :1221-1222
    # the oracle's own dot-product test agrees within tolerance
    n = int(np.prod(shape))
    hm, hd = u01(oracle, dt, SEED_M, 1, n), u01(oracle, dt, SEED_D, 1, nrow * n)
    olhs, orhs = oracle.dot_product_test(ops, [hm], [hd[i * n:(i + 1) * n].copy() for i in range(nrow)])
    assert abs(complex(lhs) - complex(olhs)) <= 50 * _dpt_tol(dt) * abs(olhs)


# ---------------------------------------------------------------------------------- general path
def _mixed_ops(Jets, oracle, dt, kinds, lens_r, lens_c, seed=21):
    """Build the same nrow x ncol mixed operator on the device and in the oracle. kinds[i][j] in
    {'zero','identity','scale','diag','diag_adj'}; elementwise blocks need lens_r[i] == lens_c[j]."""
    dev_rows, ora_rows = [], []
    for i, row in enumerate(kinds):
        dr, orow = [], []
        for j, k in enumerate(row):
            nr, nc = lens_r[i], lens_c[j]
            dom, rng = Jets.JetSpace(dt, nc), Jets.JetSpace(dt, nr)
            if k == "zero":
                dr.append(Jets.JopZeroBlock(dom, rng)); orow.append(oracle.Block("zero", nr, nc))
            elif k == "identity":
                dr.append(Jets.JopIdentity(dom)); orow.append(oracle.Block("identity", nr))
            elif k == "scale":
                a = (0.3 + i) - (0.25j * (j + 1) if np.dtype(dt).kind == "c" else 0)
                dr.append(Jets.JopLn(dom=dom, rng=dom, df=Jets.constdiag_df, df_adj=Jets.constdiag_df_adj, s={"a": a}))
                orow.append(oracle.Block("scale", nr, scale=a))
            else:
                stream = 100 * i + j
                dg = Jets.rand(dom, seed=seed, stream=stream)
                op = Jets.JopDiagonal(dg)
                hb = oracle.Block("diag", nr, coeff=u01(oracle, dt, seed, stream, nr), adjoint=(k == "diag_adj"))
                dr.append(op.H if k == "diag_adj" else op); orow.append(hb)
        dev_rows.append(dr); ora_rows.append(orow)
    return Jets.blockop(dev_rows), ora_rows


@pytest.fixture(params=[1, 2, 4, 0], ids=["tiled", "tiled-2-lines", "tiled-4-lines", "one-line"])
def general_tile(request, Jets):
    """Grids of EQUAL elementwise blocks run register-tiled (k_general_tile: two lines x one tile per workgroup in round 3, four from
    four lines on since round 4) or, with the knob off, on the one-line-per-workgroup general kernels: the same bits either way."""
    Jets.tune(general_tile=request.param)
    yield request.param
    Jets.tune(general_tile=1)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("n", [10, 64, 4100])      # 10: scalar kernel (odd 16-byte alignment); 64, 4100: 16-byte vector kernel
def test_mixed_3x4_with_zero_blocks_bit_exact(Jets, oracle, dt, n, general_tile):
    """test/runtests.jl:622-695 shape (3x4, zero blocks at (2,2),(3,4)) with native kinds."""
    kinds = [["diag", "identity", "diag", "scale"],
             ["diag_adj", "zero", "diag", "diag"],
             ["scale", "diag", "diag_adj", "zero"]]
    A, ops = _mixed_ops(Jets, oracle, dt, kinds, [n] * 3, [n] * 4)
    assert Jets.nblocks_op(A) == (3, 4) and Jets.nblocks_op(A, 1) == 3 and Jets.nblocks_op(A, 2) == 4
    assert isinstance(Jets.domain(A), Jets.JetBSpace) and Jets.domain(A).length() == 4 * n and Jets.range(A).length() == 3 * n
    m = Jets.rand(Jets.domain(A), seed=SEED_M, stream=2)
    hm = u01(oracle, dt, SEED_M, 2, 4 * n)
    hm_blocks = [hm[j * n:(j + 1) * n].copy() for j in range(4)]
    d = A * m
    ref_d = oracle.block_df(ops, [np.zeros(n, dtype=dt) for _ in range(3)], hm_blocks)
    assert_bits_equal(d.to_numpy(), np.concatenate(ref_d), "F*m")
    dd = Jets.rand(Jets.range(A), seed=SEED_D, stream=2)
    hd = u01(oracle, dt, SEED_D, 2, 3 * n)
    hd_blocks = [hd[i * n:(i + 1) * n].copy() for i in range(3)]
    mt = Jets.mul_(Jets.rand(Jets.domain(A), seed=5, stream=5), A.H, dd)          # dirty output (:684)
    ref_m = oracle.block_df_adj(ops, [np.full(n, 3, dtype=dt) for _ in range(4)], hd_blocks)
    assert_bits_equal(mt.to_numpy(), np.concatenate(ref_m), "L'*d into a dirty vector")
    lhs, rhs = Jets.dot_product_test(A, m, dd)
    assert abs(lhs - rhs) / abs(lhs + rhs) < _dpt_tol(dt)


def test_forward_accumulates_into_dirty_output_when_ncol_gt_1(Jets, oracle):
    """Reference quirk, src/Jets.jl:1024: `_d .+=` without zeroing -- mul!(d, A, m) on a dirty d
    returns d_old + A m for ncol > 1; overwrite for ncol == 1 (1026)."""
    dt, n = np.float64, 12
    A, ops = _mixed_ops(Jets, oracle, dt, [["diag", "diag"], ["identity", "zero"]], [n, n], [n, n])
    m = Jets.rand(Jets.domain(A), seed=SEED_M, stream=3)
    hm = u01(oracle, dt, SEED_M, 3, 2 * n)
    d = Jets.rand(Jets.range(A), seed=8, stream=8)
    hd0 = u01(oracle, dt, 8, 8, 2 * n)
    Jets.mul_(d, A, m)
    ref = oracle.block_df(ops, [hd0[:n].copy(), hd0[n:].copy()], [hm[:n].copy(), hm[n:].copy()])
    assert_bits_equal(d.to_numpy(), np.concatenate(ref), "dirty accumulate")
    assert not np.array_equal(np.concatenate(ref), np.concatenate(oracle.block_df(ops, [np.zeros(n), np.zeros(n)], [hm[:n].copy(), hm[n:].copy()])))


def test_zero_block_leaves_output_untouched_in_tall_forward(Jets, oracle):
    """src/Jets.jl:1022: a zero block in a one-column operator is skipped, d_i keeps its old value."""
    dt, n = np.float32, 20
    A, ops = _mixed_ops(Jets, oracle, dt, [["diag"], ["zero"], ["scale"]], [n, n, n], [n])
    m = Jets.rand(Jets.domain(A), seed=SEED_M, stream=4)
    hm = u01(oracle, dt, SEED_M, 4, n)
    d = Jets.rand(Jets.range(A), seed=8, stream=9)
    hd0 = u01(oracle, dt, 8, 9, 3 * n)
    Jets.mul_(d, A, m)
    ref = oracle.block_df(ops, [hd0[i * n:(i + 1) * n].copy() for i in range(3)], [hm])
    assert_bits_equal(d.to_numpy(), np.concatenate(ref), "zero block skipped")
    assert_bits_equal(Jets.getblock(d, 1).to_numpy(), hd0[n:2 * n], "untouched block")


@pytest.mark.parametrize("dt", [np.float32, np.complex128])
def test_short_and_fat_and_singleton(Jets, oracle, dt):
    """test/runtests.jl:744-758 (1x3) and 704-718 (1x1)."""
    n = 15
    A, ops = _mixed_ops(Jets, oracle, dt, [["diag", "diag_adj", "scale"]], [n], [n, n, n])
    m = Jets.rand(Jets.domain(A), seed=SEED_M, stream=6)
    hm = u01(oracle, dt, SEED_M, 6, 3 * n)
    d = A * m
    ref_d = oracle.block_df(ops, [np.zeros(n, dtype=dt)], [hm[j * n:(j + 1) * n].copy() for j in range(3)])
    assert_bits_equal(d.to_numpy(), ref_d[0], "wide forward")
    dd = Jets.rand(Jets.range(A), seed=SEED_D, stream=6)
    hd = u01(oracle, dt, SEED_D, 6, n)
    mt = A.H * dd
    ref_m = oracle.block_df_adj(ops, [np.zeros(n, dtype=dt) for _ in range(3)], [hd])
    assert_bits_equal(mt.to_numpy(), np.concatenate(ref_m), "wide adjoint == [B1'd; B2'd; B3'd]")

    S, sops = _mixed_ops(Jets, oracle, dt, [["diag"]], [n], [n])
    ms = Jets.rand(Jets.domain(S), seed=SEED_M, stream=7)
    hms = u01(oracle, dt, SEED_M, 7, n)
    assert_bits_equal((S * ms).to_numpy(), oracle.block_df(sops, [np.zeros(n, dtype=dt)], [hms])[0], "singleton A*m")
    assert_bits_equal((S.H * ms).to_numpy().ravel(order="F"), oracle.block_df_adj(sops, [np.zeros(n, dtype=dt)], [hms])[0], "singleton A'*d")


def test_ragged_block_lengths(Jets, oracle):
    """Heterogeneous block sizes (benchmark/benchmarks.jl:126-157 'Block, heterogeneous'): odd lengths
    put block starts off 16-byte alignment; zero blocks may be rectangular."""
    dt = np.float32
    lens = [7, 130, 1]
    kinds = [["diag", "zero", "zero"], ["zero", "diag", "zero"], ["zero", "zero", "scale"]]
    A, ops = _mixed_ops(Jets, oracle, dt, kinds, lens, lens)
    m = Jets.rand(Jets.domain(A), seed=SEED_M, stream=8)
    hm = u01(oracle, dt, SEED_M, 8, sum(lens))
    offs = np.cumsum([0] + lens)
    hm_blocks = [hm[offs[j]:offs[j + 1]].copy() for j in range(3)]
    d = A * m
    ref_d = oracle.block_df(ops, [np.zeros(k, dtype=dt) for k in lens], hm_blocks)
    assert_bits_equal(d.to_numpy(), np.concatenate(ref_d), "ragged forward")
    mt = A.H * d
    ref_m = oracle.block_df_adj(ops, [np.zeros(k, dtype=dt) for k in lens], ref_d)
    assert_bits_equal(mt.to_numpy(), np.concatenate(ref_m), "ragged adjoint")


def test_config1_4x4_identity_float64_dot_product_test(Jets, oracle):
    """BASELINE.json configs[0]: 4x4 JopBlock of identity JopLn on JetSpace(Float64,128)."""
    dt, n = np.float64, 128
    A, ops = _mixed_ops(Jets, oracle, dt, [["identity"] * 4 for _ in range(4)], [n] * 4, [n] * 4)
    m = Jets.rand(Jets.domain(A), seed=SEED_M, stream=9)
    d = Jets.rand(Jets.range(A), seed=SEED_D, stream=9)
    lhs, rhs = Jets.dot_product_test(A, m, d)
    assert abs(lhs - rhs) / abs(lhs + rhs) < 1e-12
    hm, hd = u01(oracle, dt, SEED_M, 9, 4 * n), u01(oracle, dt, SEED_D, 9, 4 * n)
    split = lambda v: [v[i * n:(i + 1) * n].copy() for i in range(4)]
    olhs, orhs = oracle.dot_product_test(ops, split(hm), split(hd))
    assert abs(olhs - orhs) / abs(olhs + orhs) < 1e-12
    assert lhs == pytest.approx(olhs, rel=1e-12) and rhs == pytest.approx(orhs, rel=1e-12)
    ref = oracle.block_df(ops, [np.zeros(n) for _ in range(4)], split(hm))
    assert_bits_equal((A * m).to_numpy(), np.concatenate(ref), "4x4 identity forward")


def test_tiny_and_denormal_values_survive(Jets, oracle):
    """No flush-to-zero, no FMA contraction: products of tiny values match the CPU bit for bit."""
    dt, n, nrow = np.float32, 64, 4
    tiny = (np.arange(1, n + 1, dtype=np.float32) * np.float32(1e-22)).astype(dt)
    diags = [tiny * np.float32(k + 1) for k in range(nrow)]
    A = Jets.blockop([[Jets.JopDiagonal(Jets.from_numpy(g))] for g in diags])
    ops = [[oracle.Block("diag", n, coeff=g)] for g in diags]
    hm = (np.arange(n, 0, -1, dtype=np.float32) * np.float32(3e-20)).astype(dt)     # products ~1e-40: denormal
    m = Jets.from_numpy(hm)
    d = A * m
    ref_d = oracle.block_df(ops, [np.zeros(n, dtype=dt) for _ in range(nrow)], [hm])
    assert np.any((np.concatenate(ref_d) != 0) & (np.abs(np.concatenate(ref_d)) < np.finfo(np.float32).tiny))
    assert_bits_equal(d.to_numpy(), np.concatenate(ref_d), "denormal forward")
    big = [np.full(n, 3e18, dtype=dt) for _ in range(nrow)]
    dd = Jets.from_numpy(np.concatenate(big), Jets.range(A))
    mt = A.H * dd
    ref_m = oracle.block_df_adj(ops, [np.zeros(n, dtype=dt)], big)
    assert_bits_equal(mt.to_numpy(), ref_m[0], "ordered accumulate")


def test_shape_errors_are_loud(Jets, oracle):
    A, *_ = make_tall_diag(Jets, oracle, np.float32, 3, (8,))
    with pytest.raises(Jets.JetsHipError):
        Jets.mul_(Jets.zeros(Jets.JetSpace(np.float32, 23)), A, Jets.zeros(Jets.domain(A)))
    with pytest.raises(Jets.JetsHipError):
        Jets.mul_(Jets.zeros(Jets.range(A)), A, Jets.zeros(Jets.JetSpace(np.float64, 8)))
    with pytest.raises(Jets.JetsHipError):                                         # elementwise block must be square
        Jets.blockop([[Jets.JopDiagonal(Jets.rand(Jets.JetSpace(np.float32, 4))), Jets.JopDiagonal(Jets.rand(Jets.JetSpace(np.float32, 5)))]]) * \
            Jets.zeros(Jets.JetBSpace([Jets.JetSpace(np.float32, 4), Jets.JetSpace(np.float32, 5)]))


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_scalar_times_tall_block_operator_fused_bit_exact(Jets, oracle, dt):
    """a*A (src/Jets.jl:1159-1164) on a tall diagonal block operator: one fused launch each way, same bits as the
    unfused chain  d = a .* (A m)  and  m = A'(a .* d)  restated with the oracle."""
    nrow, shape, a = 5, (32, 16, 8), 0.7
    A, _, ops, _ = make_tall_diag(Jets, oracle, dt, nrow, shape)
    n = int(np.prod(shape))
    B = a * A
    assert Jets.range(B) == Jets.range(A) and Jets.domain(B) == Jets.domain(A)      # documented fix of src/Jets.jl:1162
    m = Jets.rand(Jets.domain(A), seed=SEED_M, stream=11)
    hm = u01(oracle, dt, SEED_M, 11, n)
    d = Jets.mul_(Jets.rand(Jets.range(A), seed=5, stream=5), B, m)                 # dirty output, overwritten
    tmp = oracle.block_df(ops, [np.zeros(n, dtype=dt) for _ in range(nrow)], [hm])
    ref = oracle.barr_lincomb([np.empty(n, dtype=dt) for _ in range(nrow)], [a], [tmp])
    assert_bits_equal(d.to_numpy(), np.concatenate(ref), "(a*A) m")
    dd = Jets.rand(Jets.range(A), seed=SEED_D, stream=11)
    hd = u01(oracle, dt, SEED_D, 11, nrow * n)
    hd_blocks = [hd[i * n:(i + 1) * n].copy() for i in range(nrow)]
    mt = Jets.mul_(Jets.rand(Jets.domain(A), seed=6, stream=6), B.H, dd)
    scaled = oracle.barr_lincomb([np.empty(n, dtype=dt) for _ in range(nrow)], [a], [hd_blocks])
    ref_m = oracle.block_df_adj(ops, [np.zeros(n, dtype=dt)], scaled)
    assert_bits_equal(mt.to_numpy().ravel(order="F"), ref_m[0], "(a*A)' d")
    lhs, rhs = Jets.dot_product_test(B, m, dd)
    assert abs(lhs - rhs) / abs(lhs + rhs) < _dpt_tol(dt)
    # a linear combination of block operators through JetSum (docs/src/index.md: A = 1.0*A1 - 2.0*A2)
    A2, _, ops2, _ = make_tall_diag(Jets, oracle, dt, nrow, shape, seed=77)
    S = 1.5 * A - 2.0 * A2
    got = (S * m).to_numpy()
    t1 = oracle.barr_lincomb([np.empty(n, dtype=dt) for _ in range(nrow)], [1.5], [tmp])
    tmp2 = oracle.block_df(ops2, [np.zeros(n, dtype=dt) for _ in range(nrow)], [hm])
    t2 = oracle.barr_lincomb([np.empty(n, dtype=dt) for _ in range(nrow)], [2.0], [tmp2])
    acc = oracle.barr_lincomb([np.empty(n, dtype=dt) for _ in range(nrow)], [1.0, 1.0], [[np.zeros(n, dtype=dt)] * nrow, t1])
    acc = oracle.barr_lincomb([np.empty(n, dtype=dt) for _ in range(nrow)], [1.0, -1.0], [acc, t2])
    assert_bits_equal(got, np.concatenate(acc), "1.5*A - 2.0*A2")
    # adjoint of the sum: S' d = A'(1.5 d) - A2'(2.0 d), each term's rows summed in order, then the terms combined
    got_adj = (S.H * dd).to_numpy().ravel(order="F")
    sc1 = oracle.barr_lincomb([np.empty(n, dtype=dt) for _ in range(nrow)], [1.5], [hd_blocks])
    sc2 = oracle.barr_lincomb([np.empty(n, dtype=dt) for _ in range(nrow)], [2.0], [hd_blocks])
    m1 = oracle.block_df_adj(ops, [np.zeros(n, dtype=dt)], sc1)
    m2 = oracle.block_df_adj(ops2, [np.zeros(n, dtype=dt)], sc2)
    macc = oracle.barr_lincomb([np.empty(n, dtype=dt)], [1.0, 1.0], [[np.zeros(n, dtype=dt)], m1])
    macc = oracle.barr_lincomb([np.empty(n, dtype=dt)], [1.0, -1.0], [macc, m2])
    assert_bits_equal(got_adj, macc[0], "(1.5*A - 2.0*A2)' d")
    # three terms, bare and scaled mixed:  A + 0.25*A2 - A
    S3 = A + 0.25 * A2 - A
    t3 = oracle.barr_lincomb([np.empty(n, dtype=dt) for _ in range(nrow)], [0.25], [tmp2])
    a3 = oracle.barr_lincomb([np.empty(n, dtype=dt) for _ in range(nrow)], [1.0, 1.0], [[np.zeros(n, dtype=dt)] * nrow, tmp])
    a3 = oracle.barr_lincomb([np.empty(n, dtype=dt) for _ in range(nrow)], [1.0, 1.0], [a3, t3])
    a3 = oracle.barr_lincomb([np.empty(n, dtype=dt) for _ in range(nrow)], [1.0, -1.0], [a3, tmp])
    assert_bits_equal((S3 * m).to_numpy(), np.concatenate(a3), "A + 0.25*A2 - A")
    lhs, rhs = Jets.dot_product_test(S, m, dd)
    assert abs(lhs - rhs) / abs(lhs + rhs) < 10 * _dpt_tol(dt)


def test_more_than_65535_block_rows_through_the_general_kernels(Jets):
    """grid.y alone caps at 65535 workgroups: the general kernels spread block rows / columns over grid.y x grid.z."""
    import ctypes as C

    from jets_jl_amd._ffi import BlockDesc, lib, KINDS, check

    nrow, n = 70_001, 4
    arr = (BlockDesc * nrow)()
    for i in range(nrow):
        arr[i].kind = KINDS["identity"] if i % 2 else KINDS["scale"]
        arr[i].scale_re = 1.0 + (i % 7)
        arr[i].nr = arr[i].nc = n
    rl = (C.c_int64 * nrow)(*([n] * nrow))
    cl = (C.c_int64 * 1)(n)
    h = C.c_void_p()
    check(lib.jh_blockop_create(nrow, 1, arr, rl, cl, 0, C.byref(h)))
    m = Jets.from_numpy(np.array([1.0, -2.0, 3.0, 0.5], np.float32))
    d = Jets.zeros(Jets.JetBSpace([Jets.JetSpace(np.float32, n)] * nrow))
    check(lib.jh_blockop_mul(h, d.handle, m.handle))
    scale = np.array([1.0 if i % 2 else 1.0 + (i % 7) for i in range(nrow)], np.float32)
    want = scale[:, None] * np.array([1.0, -2.0, 3.0, 0.5], np.float32)[None, :]
    assert_bits_equal(d.to_numpy(), want.ravel(), "70001 x 1 forward")
    mt = Jets.zeros(Jets.JetSpace(np.float32, n))
    check(lib.jh_blockop_mul_adj(h, mt.handle, d.handle))
    acc = np.zeros(n, np.float32)
    for i in range(nrow):                                                             # rows in order, product rounded then added
        acc = acc + scale[i] * want[i]
    assert_bits_equal(mt.to_numpy(), acc, "70001 x 1 adjoint")
    # the transpose shape: 1 x 70001 (block columns over grid.y x grid.z in the adjoint)
    h2 = C.c_void_p()
    check(lib.jh_blockop_create(1, nrow, arr, cl, rl, 0, C.byref(h2)))
    mm = Jets.from_numpy(want.ravel(), Jets.JetBSpace([Jets.JetSpace(np.float32, n)] * nrow))
    out = Jets.zeros(Jets.JetSpace(np.float32, n))
    check(lib.jh_blockop_mul(h2, out.handle, mm.handle))                             # d += sum_j A_j m_j
    assert_bits_equal(out.to_numpy(), acc, "1 x 70001 forward")
    back = Jets.zeros(Jets.JetBSpace([Jets.JetSpace(np.float32, n)] * nrow))
    check(lib.jh_blockop_mul_adj(h2, back.handle, m.handle))
    assert_bits_equal(back.to_numpy(), want.ravel(), "1 x 70001 adjoint")
    lib.jh_blockop_destroy(h)
    lib.jh_blockop_destroy(h2)


@pytest.mark.parametrize("dt", DTYPES)
def test_tall_operator_with_a_few_scalar_rows_stays_on_the_tall_kernels(Jets, oracle, dt):
    """[A_1; ...; A_6; I; a*I]: identity / scalar rows ride the tall kernels through the per-row kind (MIXED instantiations,
    no constant diagonals are materialised) -- bit-identical to the oracle -- and the fused A'A and the one-pass LSQR step apply."""
    import ctypes as C

    from jets_jl_amd._ffi import lib
    from jets_jl_amd import jetblock

    n, ndiag = 4096, 6
    spc = Jets.JetSpace(dt, n)
    a = (0.75 - 0.5j) if np.dtype(dt).kind == "c" else -0.75
    diags = [Jets.rand(spc, seed=7, stream=i) for i in range(ndiag)]
    rows = [[Jets.JopDiagonal(g)] for g in diags]
    rows += [[Jets.JopIdentity(spc)], [Jets.JopLn(dom=spc, rng=spc, df=Jets.constdiag_df, df_adj=Jets.constdiag_df_adj, s={"a": a})]]
    A = Jets.blockop(rows)
    ora = [[oracle.Block("diag", n, coeff=u01(oracle, dt, 7, i, n))] for i in range(ndiag)]
    ora += [[oracle.Block("identity", n)], [oracle.Block("scale", n, scale=a)]]
    nrow = ndiag + 2
    m = Jets.rand(spc, seed=8, stream=0)
    hm = u01(oracle, dt, 8, 0, n)
    d = Jets.rand(Jets.range(A), seed=9, stream=0)
    Jets.mul_(d, A, m)
    ref = oracle.block_df(ora, [np.zeros(n, dt) for _ in range(nrow)], [hm])
    assert_bits_equal(d.to_numpy(), np.concatenate(ref), "forward")
    mt = Jets.mul(A.H, d)
    refm = oracle.block_df_adj(ora, [np.zeros(n, dt)], ref)
    assert_bits_equal(mt.to_numpy(), refm[0], "adjoint")
    assert_bits_equal(Jets.mul(A.H @ A, m).to_numpy(), refm[0], "fused A'A")
    nat = jetblock._native_op(A.jet.s["_native"], A.jet.s["ops"], A.jet.rng.eltype())
    w, out = Jets.zeros(spc), C.c_double(0)
    assert lib.jh_blockop_bidiag_step(nat.handle, d.handle, m.handle, w.handle, 1.0, 0.0, C.byref(out)) == 0    # a tall operator of elementwise rows
    assert_bits_equal(w.to_numpy(), refm[0], "one-pass step")
    # many scalar rows are served the same way
    B = Jets.blockop([[Jets.JopDiagonal(diags[0])], [Jets.JopIdentity(spc)], [Jets.JopIdentity(spc)]])
    natB = jetblock._native_op(B.jet.s["_native"], B.jet.s["ops"], B.jet.rng.eltype())
    u3 = Jets.zeros(Jets.range(B))
    assert lib.jh_blockop_bidiag_step(natB.handle, u3.handle, m.handle, w.handle, 1.0, 0.0, C.byref(out)) == 0
    oraB = [[oracle.Block("diag", n, coeff=u01(oracle, dt, 7, 0, n))], [oracle.Block("identity", n)], [oracle.Block("identity", n)]]
    refB = oracle.block_df(oraB, [np.zeros(n, dt) for _ in range(3)], [hm])
    assert_bits_equal(u3.to_numpy(), np.concatenate(refB), "one-pass step on 1 diagonal + 2 identity rows: u")
    assert_bits_equal(w.to_numpy(), oracle.block_df_adj(oraB, [np.zeros(n, dt)], refB)[0], "... w")
    # a block length that is not a multiple of 16 bytes: under-aligned packs since round 5 (tests/test_gpu_tall_unaligned.py); rows shorter than ONE pack
    # have no tall tiling: the step says so and the caller takes another path
    odd = Jets.JetSpace(dt, 4097)
    Cop = Jets.blockop([[Jets.JopDiagonal(Jets.rand(odd, seed=7, stream=0))], [Jets.JopIdentity(odd)]])
    natC = jetblock._native_op(Cop.jet.s["_native"], Cop.jet.s["ops"], Cop.jet.rng.eltype())
    assert lib.jh_blockop_bidiag_step(natC.handle, Jets.zeros(Jets.range(Cop)).handle, Jets.zeros(odd).handle, Jets.zeros(odd).handle, 1.0, 0.0, C.byref(out)) == 0
    if np.dtype(dt).itemsize < 16:
        tiny = Jets.JetSpace(dt, 1)
        Dop = Jets.blockop([[Jets.JopDiagonal(Jets.rand(tiny, seed=7, stream=0))], [Jets.JopIdentity(tiny)]])
        natD = jetblock._native_op(Dop.jet.s["_native"], Dop.jet.s["ops"], Dop.jet.rng.eltype())
        assert lib.jh_blockop_bidiag_step(natD.handle, Jets.zeros(Jets.range(Dop)).handle, Jets.zeros(tiny).handle, Jets.zeros(tiny).handle, 1.0, 0.0, C.byref(out)) == 4


@pytest.mark.parametrize("rows", [1, 3, 5, 64])
def test_adjoint_in_several_row_launches_gives_the_bits_of_one(Jets, oracle, rows):
    """knob adj_rows_per_launch: the ordered sum continues across launches (also the fused A'A)."""
    dt, nrow, shape = np.float32, 13, (40, 40, 12)
    A, _, ops, _ = make_tall_diag(Jets, oracle, dt, nrow, shape)
    n = int(np.prod(shape))
    m = Jets.rand(Jets.domain(A), seed=SEED_M, stream=0)
    d = A * m
    hm = u01(oracle, dt, SEED_M, 0, n)
    ref_d = oracle.block_df(ops, [np.zeros(n, dtype=dt) for _ in range(nrow)], [hm])
    ref_m = oracle.block_df_adj(ops, [np.zeros(n, dtype=dt)], ref_d)
    try:
        Jets.tune(adj_rows_per_launch=rows)
        mt = Jets.rand(Jets.domain(A), seed=77, stream=0)                             # dirty output
        Jets.mul_(mt, A.H, d)
        assert_bits_equal(mt.to_numpy().ravel(order="F"), ref_m[0], f"adjoint, {rows} rows per launch")
        y = Jets.mul(A.H @ A, m)
        assert_bits_equal(y.to_numpy().ravel(order="F"), ref_m[0], f"fused A'A, {rows} rows per launch")
    finally:
        Jets.tune(adj_rows_per_launch=0)


@pytest.mark.gpu
@pytest.mark.parametrize("dt", [np.float32, np.complex64])
def test_wide_operator_with_a_large_result_streams_it_out(Jets, oracle, dt):
    """1 x 20 diagonal blocks of 4 MiB (8 MiB complex): the adjoint's result (80 / 160 MiB, written once) leaves through
    nontemporal stores in the general kernel; bits of m_j = conj(a_j) .* d (src/Jets.jl:1051) and of the forward into a dirty d."""
    J = Jets
    K, n = 20, 1 << 20
    spc = J.JetSpace(dt, n)
    coeff = [J.rand(spc, seed=31, stream=j) for j in range(K)]
    W = J.blockop([[J.JopDiagonal(c) for c in coeff]])
    hd = u01(oracle, dt, 32, 0, n)
    d = J.from_numpy(hd)
    mt = J.rand(J.domain(W), seed=33, stream=0)                      # dirty
    J.mul_(mt, W.H, d)
    got = mt.to_numpy()
    ops = [[oracle.Block("diag", n, coeff=u01(oracle, dt, 31, j, n)) for j in range(K)]]
    want_m = oracle.block_df_adj(ops, [np.empty(n, dt) for _ in range(K)], [hd])
    for j in (0, 7, 19):
        assert_bits_equal(got[j * n:(j + 1) * n], want_m[j], f"wide adjoint, block column {j}")
    d2 = J.from_numpy(hd)                                            # the forward adds to d as found (1024)
    J.mul_(d2, W, mt)
    want = oracle.block_df(ops, [hd.copy()], want_m)[0]
    assert_bits_equal(d2.to_numpy(), want, "wide forward into d as found, columns in order")
    J.close(W)


@pytest.mark.gpu
@pytest.mark.parametrize("dt", [np.float32, np.complex64])
@pytest.mark.parametrize("mixed", [False, True])
def test_wide_operator_of_large_blocks_runs_on_its_tall_twin(Jets, oracle, dt, mixed):
    """1 x 6 elementwise children of 16 MiB: the adjoint is the tall twin's forward, the forward the twin's ordered adjoint sum
    started from d AS FOUND (src/Jets.jl:1024) -- bits of the oracle's loops, with a zero block and a scalar block in the mixed case."""
    J = Jets
    K = 6
    n = (1 << 22) // (np.dtype(dt).itemsize // 4)
    spc = J.JetSpace(dt, n)
    dev, ora = [], []
    for j in range(K):
        if mixed and j == 2:
            dev.append(J.JopZeroBlock(spc, spc)); ora.append(oracle.Block("zero", n))
        elif mixed and j == 4:
            a = 0.75 - (0.5j if np.dtype(dt).kind == "c" else 0)
            dev.append(J.JopLn(dom=spc, rng=spc, df=J.constdiag_df, df_adj=J.constdiag_df_adj, s={"a": a}))
            ora.append(oracle.Block("scale", n, scale=a))
        else:
            op = J.JopDiagonal(J.rand(spc, seed=41, stream=j))
            adj = mixed and j == 1
            dev.append(op.H if adj else op)
            ora.append(oracle.Block("diag", n, coeff=u01(oracle, dt, 41, j, n), adjoint=adj))
    W = J.blockop([dev])
    hm = [u01(oracle, dt, 42, j, n) for j in range(K)]
    hd = u01(oracle, dt, 43, 0, n)
    m = J.from_numpy(np.concatenate(hm), J.domain(W))
    d = J.from_numpy(hd)                                             # dirty: the forward adds to it
    J.mul_(d, W, m)
    want_d = oracle.block_df([ora], [hd.copy()], hm)[0]
    assert_bits_equal(d.to_numpy(), want_d, "wide forward of large blocks into d as found")
    assert J.tune_get("last_adj_parts") == 1
    mt = J.rand(J.domain(W), seed=44, stream=0)                      # dirty: a zero block's column stays as found (1047)
    found = mt.to_numpy().copy()
    J.mul_(mt, W.H, d)
    want_m = oracle.block_df_adj([ora], [found[j * n:(j + 1) * n].copy() for j in range(K)], [want_d])
    assert_bits_equal(mt.to_numpy(), np.concatenate(want_m), "wide adjoint of large blocks")
    J.close(W)


@pytest.mark.gpu
@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("shape", [(2, 2), (3, 5), (7, 4), (5, 9), (19, 3), (9, 17)])
def test_grid_of_plain_diagonals_on_the_branch_free_kernel(Jets, oracle, dt, shape):
    """M x K grids whose blocks are all un-adjointed diagonals run register-tiled on k_grid_tile (a workgroup owns 2 / 4 / 8 lines x
    one element tile; round 3) or on k_grid_diag (one line per workgroup, four blocks' loads in flight per lane; knob grid_tile = 0):
    the bits of the oracle's loops -- forward into d AS FOUND (1024), adjoint from zero (1042) -- and of the general kernels."""
    J = Jets
    M, K = shape
    n = 1024 + 64 * M                                                # 16-byte multiples for every eltype; several tiles
    spc = J.JetSpace(dt, n)
    coeff = [[J.rand(spc, seed=61, stream=i * K + j) for j in range(K)] for i in range(M)]
    A = J.blockop([[J.JopDiagonal(c) for c in row] for row in coeff])
    ops = [[oracle.Block("diag", n, coeff=u01(oracle, dt, 61, i * K + j, n)) for j in range(K)] for i in range(M)]
    hm = [u01(oracle, dt, 62, j, n) for j in range(K)]
    hd = [u01(oracle, dt, 63, i, n) for i in range(M)]
    got = {}
    routes = [(1, 1), (1, 2), (1, 4), (1, 8), (1, 0), (2, 0), (4, 0), (0, 0)]   # (grid_diag, grid_tile): tiled with automatic / forced R (lines
    for gd in routes:                                                # beyond the last group are clamped), k_grid_diag with 1 / 2 / 4 packs per lane, general kernels
        J.tune(grid_diag=gd[0], grid_tile=gd[1])
        try:
            m = J.from_numpy(np.concatenate(hm), J.domain(A))
            d = J.from_numpy(np.concatenate(hd), J.range(A))         # dirty
            J.mul_(d, A, m)
            mt = J.rand(J.domain(A), seed=64, stream=0)              # dirty
            J.mul_(mt, A.H, d)
            got[gd] = (d.to_numpy(), mt.to_numpy())
        finally:
            J.tune(grid_diag=1, grid_tile=1)
    want_d = oracle.block_df(ops, [b.copy() for b in hd], hm)
    want_m = oracle.block_df_adj(ops, [np.zeros(n, dt) for _ in range(K)], want_d)
    for gd in routes:
        assert_bits_equal(got[gd][0], np.concatenate(want_d), f"grid forward, grid_diag={gd}")
        assert_bits_equal(got[gd][1], np.concatenate(want_m), f"grid adjoint, grid_diag={gd}")
    J.close(A)


@pytest.mark.gpu
@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("shape", [(2, 2), (5, 3), (4, 7), (9, 9)])
def test_grids_of_equal_blocks_of_every_kind_tiled_and_not(Jets, oracle, dt, shape, general_tile):
    """M x K grids of equal blocks drawn from every elementwise kind -- incl. a block row and a block column of zero blocks only -- on
    k_general_tile (two lines x one tile per workgroup, branch-free loads; an odd line count clamps the last group) and on the
    one-line kernels: forward into a DIRTY d (`_d .+=`, 1024; the row of zero blocks stays as found, 1022), adjoint into a dirty m
    (zeroed, 1042): the oracle's bits."""
    J = Jets
    M, K = shape
    n = 2048 + 64                                                     # several tiles, 16-byte multiples for every eltype
    rng = np.random.default_rng(1000 * M + K)
    names = ["diag", "diag", "diag_adj", "identity", "scale", "zero"]
    kinds = [[names[rng.integers(len(names))] for _ in range(K)] for _ in range(M)]
    kinds[M - 1] = ["zero"] * K                                        # a block row of zero blocks only
    for i in range(M):
        kinds[i][K - 1] = "zero" if i else "diag"                      # a block column with ONE non-zero block
    A, ops = _mixed_ops(J, oracle, dt, kinds, [n] * M, [n] * K)
    hm = [u01(oracle, dt, SEED_M, j, n) for j in range(K)]
    hd = [u01(oracle, dt, SEED_D, i, n) for i in range(M)]
    hmt = [u01(oracle, dt, SEED_D + 7, j, n) for j in range(K)]
    m = J.from_numpy(np.concatenate(hm), J.domain(A))
    d = J.from_numpy(np.concatenate(hd), J.range(A))
    J.mul_(d, A, m)
    want_d = oracle.block_df(ops, [b.copy() for b in hd], hm)
    assert_bits_equal(d.to_numpy(), np.concatenate(want_d), f"{M}x{K} forward, general_tile={general_tile}")
    assert_bits_equal(d.to_numpy()[(M - 1) * n:], hd[M - 1], "the row of zero blocks keeps d as found")
    mt = J.from_numpy(np.concatenate(hmt), J.domain(A))
    J.mul_(mt, A.H, d)
    want_m = oracle.block_df_adj(ops, [b.copy() for b in hmt], want_d)
    assert_bits_equal(mt.to_numpy(), np.concatenate(want_m), f"{M}x{K} adjoint, general_tile={general_tile}")
    J.close(A)


# ---- IEEE special values through the hot loops (round 3) -----------------------------------------------------------------------------
def test_special_values_go_through_every_kernel_family_like_on_the_cpu(Jets, oracle):
    """Signed zeros, infinities, NaN, denormals and the largest finite values in the coefficients AND the vectors, four element
    types: tall forward / ordered adjoint / fused A'A, rows of every elementwise kind, the one-pass step, a grid with a zero block,
    a + - + sum, dense children, jh_lincomb and the compiled broadcast give the oracle's value element by element (0 * Inf = NaN
    where the reference multiplies, an untouched 0 where it skips a zero block; complex products by the four-multiplication
    formula; REAL scalars on complex data part by part, as Julia's a::Real * z).  Bit for bit except for the payload of a NaN.
    The checks live in tools/check_specials.py (also a stand-alone program)."""
    import importlib.util
    import os

    spec = importlib.util.spec_from_file_location("check_specials", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "check_specials.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    results = mod.run_checks(Jets, oracle)
    assert len(results) >= 80
    bad = [r for r in results if r[2] is not True]
    assert not bad, bad[:8]
    # the same through the big-block routes (blocks of 16 / 32 MiB: register-tiled grids, chained one-pass step, nontemporal tall walks)
    results = mod.run_checks(Jets, oracle, seed=6, n=4 << 20, dtypes=(np.float32, np.complex64))
    bad = [r for r in results if r[2] is not True]
    assert len(results) >= 40 and not bad, bad[:8]


@pytest.mark.parametrize("dt", [np.complex64, np.complex128])
def test_a_real_scalar_multiplies_complex_data_part_by_part(Jets, oracle, dt):
    """Julia: `a::Real * z` = Complex(a*re, a*im) -- 1.0 * (x + Inf i) keeps x, and (-0.0) parts keep their sign, where the
    four-multiplication formula with an imaginary part of 0 would give NaN / +0.0.  The ABI takes scalars as (re, im): an imaginary
    part that is exactly zero means a real scalar (include/jetship.h)."""
    from .helpers import assert_same_values

    z = np.array([complex(-0.0, 3.0), complex(2.0, -0.0), complex(-0.0, -0.0), complex(1.0, np.inf), complex(-np.inf, 2.0), complex(5.0, -7.0)] * 3, dtype=dt)
    x = Jets.from_numpy(z)
    out = Jets.zeros(Jets.space(x))
    for a in (1.0, 0.75, -2.0):
        expect = np.empty_like(z)
        rt = z.real.dtype.type
        with np.errstate(all="ignore"):
            expect.real, expect.imag = rt(a) * z.real, rt(a) * z.imag
        Jets.lincomb_(out, [a], [x])
        assert_same_values(out.to_numpy(), expect, f"jh_lincomb, a = {a}")
        Jets.broadcast_(out, "s0*x0", [x], [a])
        assert_same_values(out.to_numpy(), expect, f"compiled broadcast, a = {a}")
        spc = Jets.space(x)
        S = Jets.blockop([[Jets.JopLn(dom=spc, rng=spc, df=Jets.constdiag_df, df_adj=Jets.constdiag_df_adj, s={"a": a})], [Jets.JopIdentity(spc)]])
        got = (S * x).to_numpy()
        assert_same_values(got[:z.size], expect, f"scalar block, a = {a}")
        assert_same_values(oracle.block_df([[oracle.Block("scale", z.size, scale=a)], [oracle.Block("identity", z.size)]],
                                           [np.zeros(z.size, dtype=dt) for _ in range(2)], [z])[0], expect, f"oracle scalar block, a = {a}")
    c = complex(0.75, 0.5)                                         # a COMPLEX scalar keeps the four-multiplication formula
    with np.errstate(all="ignore"):
        expect = np.array([complex(c.real * v.real - c.imag * v.imag, c.real * v.imag + c.imag * v.real) for v in z.astype(np.complex128)], dtype=np.complex128)
    Jets.lincomb_(out, [c], [x])
    got = out.to_numpy()
    assert np.array_equal(np.isnan(got.view(got.real.dtype)), np.isnan(expect.astype(dt).view(got.real.dtype)))


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("mixed", [False, True])
def test_tall_forward_in_column_bands_of_any_width_has_the_same_bits(Jets, oracle, dt, mixed):
    """The tall forward's column-band walk (k_tall_diag_fwd's ctiles decode, late round 4: `ctiles` consecutive tiles of one row group, then the
    same tiles of the next group, ... then the next band) only re-orders workgroups: whatever the band width -- narrower than a row, wider than
    it, not a divisor of the tile count -- and whatever the rows per workgroup, with a ragged last tile, strided or table-addressed
    coefficients, all-diagonal or with rows of other kinds, d = A m and the fused update d = alpha (A m) + beta d keep their bits."""
    import ctypes as C

    from jets_jl_amd import jetblock as _blk
    from jets_jl_amd._ffi import check, lib

    J = Jets
    nrow, n = 7, 4 * 4096 + 1028                                  # 17 tiles of 256 packs (Float32) and a ragged one
    spc = J.JetSpace(dt, n)
    hm = u01(oracle, dt, 2, 0, n)
    m = J.from_numpy(hm, spc)
    for strided in (True, False):
        if strided:
            coeff = J.rand(J.JetBSpace([spc] * nrow), seed=31, stream=0)
            devs = list(coeff.arrays)
            hc = [oracle.rng_u01(dt, 31, 0, i * n, n) for i in range(nrow)]
        else:
            devs = [J.rand(spc, seed=31, stream=i) for i in range(nrow)]
            hc = [u01(oracle, dt, 31, i, n) for i in range(nrow)]
        rows = [[J.JopDiagonal(c)] for c in devs]
        ops = [[oracle.Block("diag", n, coeff=c)] for c in hc]
        if mixed:
            rows[2], ops[2] = [J.JopIdentity(spc)], [oracle.Block("identity", n)]
            rows[5], ops[5] = [J.JopZeroBlock(spc, spc)], [oracle.Block("zero", n, n)]
        A = J.blockop(rows)
        hd0 = [u01(oracle, dt, 3, i, n) for i in range(nrow)]
        want = np.concatenate(oracle.block_df(ops, [b.copy() for b in hd0], [hm]))
        tmp = oracle.block_df(ops, [np.zeros(n, dt) for _ in range(nrow)], [hm])
        want_upd = np.concatenate(oracle.barr_lincomb([np.empty(n, dt) for _ in range(nrow)], [0.75, -0.5], [tmp, hd0]))
        nat = _blk._tall_native(A)
        try:
            for ct in (0, 1, 3, 5, 17, 32, 64, 1000):
                for grp in (1, 2, 5):
                    J.tune(fwd_wg=256, fwd_unroll=1, fwd_group=grp, fwd_ctiles=ct)
                    d = J.from_numpy(np.concatenate(hd0), J.range(A))
                    J.mul_(d, A, m)
                    assert_bits_equal(d.to_numpy(), want, f"forward, bands of {ct} tiles, {grp} rows per workgroup, strided {strided}, mixed {mixed}, {np.dtype(dt)}")
                    if nat is not None:
                        d = J.from_numpy(np.concatenate(hd0), J.range(A))
                        check(lib.jh_blockop_mul_axpby(nat.handle, d.handle, m.handle, 0.75, -0.5, None))
                        assert_bits_equal(d.to_numpy(), want_upd, f"forward update, bands of {ct} tiles, {grp} rows per workgroup, strided {strided}, mixed {mixed}, {np.dtype(dt)}")
        finally:
            J.tune(fwd_wg=0, fwd_unroll=0, fwd_group=0, fwd_ctiles=-1)
        J.close(A)
