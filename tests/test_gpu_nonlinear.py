"""GPU parity: nonlinear block operators -- JetBlock_f! (src/Jets.jl:988-1008), block point! (1059-1066) and the
Jacobian through JetBlock_df!/df'! -- on the MI355X vs the CPU oracle, BIT-EXACT.

The nonlinear child is the reference's own fixture JopBar (test/runtests.jl:19-24: f! d .= m.^2, df! dd .= 2 .* mo .* dm)
as the device-native kind SQUARE.  Re-encodes on seeded inputs the nonlinear halves of test/runtests.jl:704-718
(singleton), 720-742 (tall-and-skinny) and 744-758 (short-and-fat), then sweeps mixed linear/nonlinear block matrices.
"""
import numpy as np
import pytest

from .helpers import DTYPES, assert_bits_equal, u01

pytestmark = pytest.mark.gpu


def _split(v, lens):
    offs = np.cumsum([0] + list(lens))
    return [v[offs[i]:offs[i + 1]].copy() for i in range(len(lens))]


def _square_grid(Jets, oracle, dt, nrow, ncol, n, mo_blocks):
    spc = Jets.JetSpace(dt, n)
    F = Jets.blockop([[Jets.JopSquare(spc) for _ in range(ncol)] for _ in range(nrow)])
    ops = [[oracle.Block("square", n, coeff=mo_blocks[j]) for j in range(ncol)] for _ in range(nrow)]
    return F, ops


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("nrow,ncol,n", [(1, 1, 5), (3, 1, 5), (1, 3, 5), (3, 1, 4100), (2, 3, 1024), (4, 1, 257)])
def test_reference_nonlinear_block_identities(Jets, oracle, dt, nrow, ncol, n):
    """F*m, J = jacobian!(F, m), J*m, J'*d (test/runtests.jl:711-716, 727-733, 751-757)."""
    hm = u01(oracle, dt, 11, 0, ncol * n)
    hmb = _split(hm, [n] * ncol)
    F, ops = _square_grid(Jets, oracle, dt, nrow, ncol, n, hmb)
    assert isinstance(F, Jets.JopNl)
    m = Jets.rand(Jets.domain(F), seed=11, stream=0)
    d = F * m                                                                        # zeros(range) then mul! (:399)
    ref = oracle.block_f(ops, [np.zeros(n, dt) for _ in range(nrow)], hmb)
    assert_bits_equal(d.to_numpy(), np.concatenate(ref), "F*m")
    closed = sum((b * b for b in hmb[1:]), hmb[0] * hmb[0]) if ncol > 1 else hmb[0] * hmb[0]
    if ncol == 1 and np.dtype(dt).kind != "c":                                        # (numpy's complex product rounds differently)
        assert_bits_equal(ref[0], closed, "oracle vs closed form, m.^2")              # F*m == [G1 m; G2 m; G3 m]  (:728)
    else:
        np.testing.assert_allclose(ref[0], closed, rtol=1e-5)                        # F*m == sum_j Gj m_j        (:752)

    J = Jets.jacobian_(F, m)
    assert isinstance(J, Jets.JopLn)
    dm = Jets.rand(Jets.domain(F), seed=12, stream=0)
    hdm = _split(u01(oracle, dt, 12, 0, ncol * n), [n] * ncol)
    dd = Jets.rand(Jets.range(F), seed=13, stream=0)                                 # dirty output, ncol > 1 accumulates into it
    hdd = _split(u01(oracle, dt, 13, 0, nrow * n), [n] * nrow)
    Jets.mul_(dd, J, dm)
    ref = oracle.block_df(ops, [x.copy() for x in hdd], hdm)
    assert_bits_equal(dd.to_numpy(), np.concatenate(ref), "J*dm")

    mt = Jets.rand(Jets.domain(F), seed=14, stream=0)
    hmt = _split(u01(oracle, dt, 14, 0, ncol * n), [n] * ncol)
    Jets.mul_(mt, J.H, dd)
    refm = oracle.block_df_adj(ops, hmt, ref)
    assert_bits_equal(mt.to_numpy().ravel(order="F"), np.concatenate(refm), "J'*d")

    # the Jacobian shares the jet with F (src/Jets.jl:364): re-pointing F moves J too
    m2 = Jets.rand(Jets.domain(F), seed=15, stream=0)
    hm2 = _split(u01(oracle, dt, 15, 0, ncol * n), [n] * ncol)
    Jets.point_(F, m2)
    _, ops2 = _square_grid(Jets, oracle, dt, nrow, ncol, n, hm2)
    out = Jets.zeros(Jets.range(F))
    Jets.mul_(out, J, dm)
    ref2 = oracle.block_df(ops2, [np.zeros(n, dt) for _ in range(nrow)], hdm)
    assert_bits_equal(out.to_numpy(), np.concatenate(ref2), "J*dm after point!")


def test_single_square_operator_matches_fixture_formulas(Jets, oracle):
    """JopBar on its own (test/runtests.jl:19-24, 203-217 style): F*m == m.^2, J*dm == 2 .* mo .* dm, J'*dd the same."""
    n, dt = 1000, np.float64
    F = Jets.JopSquare(Jets.JetSpace(dt, n))
    m = Jets.rand(Jets.domain(F), seed=21, stream=0)
    hm = u01(oracle, dt, 21, 0, n)
    assert_bits_equal((F * m).to_numpy(), hm * hm, "m.^2")
    J = Jets.jacobian(F, m)                                                          # copy of the jet and of the point (:374)
    dm = Jets.rand(Jets.domain(F), seed=22, stream=0)
    hdm = u01(oracle, dt, 22, 0, n)
    assert_bits_equal((J * dm).to_numpy(), (2 * hm) * hdm, "2 .* mo .* dm")
    assert_bits_equal((J.H * dm).to_numpy(), (2 * hm) * hdm, "adjoint == df! for a real diagonal Jacobian")
    lhs, rhs = Jets.dot_product_test(J, Jets.rand(Jets.domain(J)), Jets.rand(Jets.range(J)))
    assert abs(lhs - rhs) <= 1e-12 * abs(lhs + rhs)


KINDS = ["square", "square", "zero", "identity", "scale", "diag", "diag_adj"]


def _mixed(Jets, oracle, dt, len_r, len_c, kinds, seed, hmo):
    dev_rows, ora_rows = [], []
    for i, row in enumerate(kinds):
        dr, orow = [], []
        for j, k in enumerate(row):
            nr, nc = len_r[i], len_c[j]
            dom, rng_ = Jets.JetSpace(dt, nc), Jets.JetSpace(dt, nr)
            if k == "zero":
                dr.append(Jets.JopZeroBlock(dom, rng_)); orow.append(oracle.Block("zero", nr, nc))
            elif k == "identity":
                dr.append(Jets.JopIdentity(dom)); orow.append(oracle.Block("identity", nr))
            elif k == "square":
                dr.append(Jets.JopSquare(dom)); orow.append(oracle.Block("square", nr, coeff=hmo[j]))
            elif k == "scale":
                a = (0.3 + 0.5 * i - 0.25 * j) - (0.125j * (j + 1) if np.dtype(dt).kind == "c" else 0)
                dr.append(Jets.JopLn(dom=dom, rng=dom, df=Jets.constdiag_df, df_adj=Jets.constdiag_df_adj, s={"a": a}))
                orow.append(oracle.Block("scale", nr, scale=a))
            elif k == "dense":
                hA = u01(oracle, dt, seed, 5000 + 100 * i + j, nr * nc).reshape(nr, nc, order="F")
                dr.append(Jets.JopDense(Jets.from_numpy(hA))); orow.append(oracle.Block("dense", nr, nc, coeff=hA))
            else:
                stream = 1000 * i + j
                op = Jets.JopDiagonal(Jets.rand(dom, seed=seed, stream=stream))
                hb = oracle.Block("diag", nr, coeff=u01(oracle, dt, seed, stream, nr), adjoint=(k == "diag_adj"))
                dr.append(op.H if k == "diag_adj" else op); orow.append(hb)
        dev_rows.append(dr); ora_rows.append(orow)
    return Jets.blockop(dev_rows), ora_rows


@pytest.mark.parametrize("case", range(40))
def test_random_mixed_nonlinear_block_operator_bitwise(Jets, oracle, case):
    """f!, Jacobian forward and Jacobian adjoint of random mixes of nonlinear and linear children, dirty outputs,
    ragged and 16-byte-unaligned block lengths: the fused f-mode launch keeps JetBlock_f!'s differences from the linear
    loop (no zero-block skip; every child's output added into d as found)."""
    rng = np.random.default_rng(77_000 + case)
    dt = DTYPES[rng.integers(len(DTYPES))]
    nrow, ncol = int(rng.integers(1, 5)), int(rng.integers(1, 5))
    pool = [1, 3, 4, 7, 16, 64, 100, 257, 1024, 4100]
    if rng.random() < 0.6:
        lens = [int(rng.choice(pool))] * max(nrow, ncol)
    else:
        lens = [int(rng.choice(pool)) for _ in range(max(nrow, ncol))]
    len_r, len_c = lens[:nrow], lens[:ncol]
    kinds = [[("zero" if len_r[i] != len_c[j] else KINDS[rng.integers(len(KINDS))]) for j in range(ncol)] for i in range(nrow)]
    if not any(k == "square" for row in kinds for k in row):
        i, j = 0, 0
        if len_r[0] == len_c[0]:
            kinds[0][0] = "square"
    NR, NC = sum(len_r), sum(len_c)
    hmo = _split(u01(oracle, dt, 31, case, NC), len_c)
    F, ops = _mixed(Jets, oracle, dt, len_r, len_c, kinds, seed=900 + case, hmo=hmo)
    tag = f"case {case}: {np.dtype(dt).name} rows={len_r} cols={len_c} kinds={kinds}"
    nonlinear = any(k == "square" for row in kinds for k in row)
    assert isinstance(F, Jets.JopNl if nonlinear else Jets.JopLn)

    mo = Jets.rand(Jets.domain(F), seed=31, stream=case)
    d = Jets.rand(Jets.range(F), seed=32, stream=case)
    hd = _split(u01(oracle, dt, 32, case, NR), len_r)
    if nonlinear:
        Jets.mul_(d, F, mo)                                                          # f! into a dirty range vector
        ref = oracle.block_f(ops, [x.copy() for x in hd], hmo)
        assert_bits_equal(d.to_numpy(), np.concatenate(ref), "f!, " + tag)
    J = Jets.jacobian_(F, mo)
    dm = Jets.rand(Jets.domain(F), seed=33, stream=case)
    hdm = _split(u01(oracle, dt, 33, case, NC), len_c)
    d2 = Jets.rand(Jets.range(F), seed=34, stream=case)
    hd2 = _split(u01(oracle, dt, 34, case, NR), len_r)
    Jets.mul_(d2, J, dm)
    ref = oracle.block_df(ops, hd2, hdm)
    assert_bits_equal(d2.to_numpy(), np.concatenate(ref), "J*dm, " + tag)
    mt = Jets.rand(Jets.domain(F), seed=35, stream=case)
    hmt = _split(u01(oracle, dt, 35, case, NC), len_c)
    Jets.mul_(mt, J.H, d2)
    refm = oracle.block_df_adj(ops, hmt, ref)
    assert_bits_equal(mt.to_numpy().ravel(order="F"), np.concatenate(refm), "J'*d, " + tag)


@pytest.mark.parametrize("dt", [np.float32, np.complex128])
def test_nonlinear_with_dense_children_uses_the_per_block_loop(Jets, oracle, dt):
    """A DENSE child sends the operator down the per-block loop (one child launch + one accumulate per block): the
    nonlinear child and the f-mode rules must hold there too.  Forward bit-exact; the dense adjoint is a wave reduction
    (tolerance parity, see jh_gemv)."""
    len_r, len_c = [64, 48], [64, 48, 64]
    kinds = [["square", "zero", "diag"], ["dense", "square", "dense"]]
    hmo = _split(u01(oracle, dt, 41, 0, sum(len_c)), len_c)
    F, ops = _mixed(Jets, oracle, dt, len_r, len_c, kinds, seed=950, hmo=hmo)
    mo = Jets.rand(Jets.domain(F), seed=41, stream=0)
    d = Jets.rand(Jets.range(F), seed=42, stream=0)
    hd = _split(u01(oracle, dt, 42, 0, sum(len_r)), len_r)
    Jets.mul_(d, F, mo)
    ref = oracle.block_f(ops, hd, hmo)
    assert_bits_equal(d.to_numpy(), np.concatenate(ref), "f! through the per-block loop")
    J = Jets.jacobian_(F, mo)
    out = Jets.zeros(Jets.range(F))
    Jets.mul_(out, J, mo)
    ref = oracle.block_df(ops, [np.zeros(n, dt) for n in len_r], hmo)
    assert_bits_equal(out.to_numpy(), np.concatenate(ref), "J*m through the per-block loop")
    mt = Jets.mul(J.H, out)
    refm = oracle.block_df_adj(ops, [np.zeros(n, dt) for n in len_c], ref)
    tol = 1e-5 if np.dtype(dt).itemsize <= 8 and np.dtype(dt) != np.float64 else 1e-12
    np.testing.assert_allclose(mt.to_numpy(), np.concatenate(refm), rtol=tol, atol=tol)


def test_f_mode_does_not_skip_zero_blocks(Jets, oracle):
    """JetBlock_f! has no iszero test (src/Jets.jl:998-1004 vs 1022): with one column a zero block OVERWRITES its row with
    zeros (the linear loop leaves the row untouched), with several columns it adds +0.0 (turning a -0.0 into +0.0)."""
    dt, n = np.float32, 260
    spc = Jets.JetSpace(dt, n)
    F = Jets.blockop([[Jets.JopSquare(spc)], [Jets.JopZeroBlock(spc, spc)]])
    m = Jets.rand(spc, seed=51, stream=0)
    d = Jets.rand(Jets.range(F), seed=52, stream=0)
    Jets.mul_(d, F, m)
    hm = u01(oracle, dt, 51, 0, n)
    assert_bits_equal(Jets.getblock(d, 1).to_numpy(), np.zeros(n, dt), "tall: zero block overwrites in f!")
    assert_bits_equal(Jets.getblock(d, 0).to_numpy(), hm * hm, "tall: square row")
    J = Jets.jacobian_(F, m)
    d2 = Jets.rand(Jets.range(F), seed=52, stream=0)
    Jets.mul_(d2, J, m)
    assert_bits_equal(Jets.getblock(d2, 1).to_numpy(), u01(oracle, dt, 52, 0, 2 * n)[n:], "tall: zero block skipped in df!")

    G = Jets.blockop([[Jets.JopZeroBlock(spc, spc), Jets.JopSquare(spc)]])
    neg0 = Jets.from_numpy(np.full(n, -0.0, dt))
    x = Jets.zeros(Jets.domain(G))                                                   # 0^2 = +0: -0.0 + 0.0 + 0.0 = +0.0
    out = Jets.similar(neg0)
    Jets.copyto_(out, neg0)
    Jets.mul_(out, G, x)
    assert not np.signbit(out.to_numpy()).any(), "wide f!: adding a zero block's +0.0 clears the sign of -0.0"
    ref = oracle.block_f([[oracle.Block("zero", n, n), oracle.Block("square", n, coeff=np.zeros(n, dt))]],
                         [np.full(n, -0.0, dt)], [np.zeros(n, dt), np.zeros(n, dt)])
    assert_bits_equal(out.to_numpy(), ref[0], "wide f! on -0.0")


def test_jacobian_without_a_point_fails_loudly(Jets):
    """The reference's jet holds an EMPTY mo until point! (src/Jets.jl:187), so df! dies with a DimensionMismatch; the
    C ABI reports JH_ERR_STATE, the host mirror raises before any launch."""
    spc = Jets.JetSpace(np.float32, 64)
    F = Jets.blockop([[Jets.JopSquare(spc)], [Jets.JopSquare(spc)]])
    with pytest.raises(Exception):
        Jets.mul(Jets.JopLn(Jets.jet(F)), Jets.rand(spc))
    # straight through the ABI
    import ctypes as C
    from jets_jl_amd._ffi import BlockDesc, lib, KINDS
    arr = (BlockDesc * 1)()
    arr[0].kind, arr[0].adjoint, arr[0].nr, arr[0].nc = KINDS["square"], 0, 64, 64
    h = C.c_void_p()
    ln = (C.c_int64 * 1)(64)
    assert lib.jh_blockop_create(1, 1, arr, ln, ln, 0, C.byref(h)) == 0
    x, y = Jets.rand(spc), Jets.zeros(spc)
    assert lib.jh_blockop_mul(h, y.handle, x.handle) == 5                             # JH_ERR_STATE
    assert b"jh_blockop_point" in lib.jh_last_error()
    assert lib.jh_blockop_mul_adj(h, y.handle, x.handle) == 5
    assert lib.jh_blockop_f(h, y.handle, x.handle) == 0                               # f! needs no point
    assert lib.jh_blockop_point(h, x.handle) == 0
    assert lib.jh_blockop_mul(h, y.handle, x.handle) == 0
    hx = x.to_numpy()
    assert_bits_equal(y.to_numpy(), (2 * hx) * hx, "J(x) x through the raw ABI")
    wrong = Jets.rand(Jets.JetSpace(np.float32, 65))
    assert lib.jh_blockop_point(h, wrong.handle) == 1                                 # JH_ERR_INVALID: length
    arr[0].adjoint = 1
    h2 = C.c_void_p()
    assert lib.jh_blockop_create(1, 1, arr, ln, ln, 0, C.byref(h2)) == 1              # a nonlinear child has no adjoint
    lib.jh_blockop_destroy(h)


def test_children_pointed_individually_take_the_per_child_path(Jets, oracle):
    """point! on a CHILD jet (not on the block jet) leaves the children at different points than the blocks of the block
    jet's mo; the reference then uses each child's own mo (mul! of the child, src/Jets.jl:1024).  The host mirror detects
    that and runs the reference's loop with one device launch per child instead of the fused launch."""
    dt, n = np.float64, 300
    spc = Jets.JetSpace(dt, n)
    G = [Jets.JopSquare(spc) for _ in range(3)]
    F = Jets.blockop([[g] for g in G])
    m = Jets.rand(spc, seed=61, stream=0)
    J = Jets.jacobian_(F, m)
    own = Jets.rand(spc, seed=62, stream=0)
    Jets.point_(Jets.jet(G[1]), own)                                                 # child 1 now linearised elsewhere
    hm, hown = u01(oracle, dt, 61, 0, n), u01(oracle, dt, 62, 0, n)
    ops = [[oracle.Block("square", n, coeff=hm)], [oracle.Block("square", n, coeff=hown)], [oracle.Block("square", n, coeff=hm)]]
    dm = Jets.rand(spc, seed=63, stream=0)
    hdm = u01(oracle, dt, 63, 0, n)
    d = J * dm
    ref = oracle.block_df(ops, [np.zeros(n, dt) for _ in range(3)], [hdm])
    assert_bits_equal(d.to_numpy(), np.concatenate(ref), "per-child points, forward")
    mt = J.H * d
    refm = oracle.block_df_adj(ops, [np.zeros(n, dt)], ref)
    assert_bits_equal(mt.to_numpy(), refm[0], "per-child points, adjoint")


def test_gauss_newton_on_a_tall_nonlinear_block_operator(Jets, oracle):
    """The caller the nonlinear path exists for: Gauss-Newton on F(m) = [m.^2; a .* m] with the Jacobian solves done by
    the device LSQR -- every product stays on the device (f!, point!, J, J')."""
    dt, n = np.float64, 2048
    spc = Jets.JetSpace(dt, n)
    a = Jets.rand(spc, seed=71, stream=0)
    F = Jets.blockop([[Jets.JopSquare(spc)], [Jets.JopDiagonal(a)], [Jets.JopSquare(spc)]])
    ha = u01(oracle, dt, 71, 0, n)
    x_true = 0.5 + u01(oracle, dt, 72, 0, n)
    dobs = Jets.from_numpy(np.concatenate([x_true * x_true, ha * x_true, x_true * x_true]), Jets.range(F))
    m = Jets.ones(spc)
    for _ in range(8):
        r = Jets.zeros(Jets.range(F))
        Jets.mul_(r, F, m)
        Jets.lincomb_(r, [1.0, -1.0], [dobs, r])                                     # r = dobs - F(m)
        J = Jets.jacobian_(F, m)
        step = Jets.lsqr(J, r, maxiter=30, atol=1e-14, btol=1e-14).x
        Jets.lincomb_(m, [1.0, 1.0], [m, step])
    np.testing.assert_allclose(m.to_numpy(), x_true, rtol=1e-9)


def test_multiple_simultaneous_linearizations_literal_values(Jets):
    """test/runtests.jl:203-217, literal values: jacobian (copies the jet) keeps J1 and J2 apart; jacobian! (shares the
    jet) makes both follow the last point."""
    F = Jets.JopSquare(Jets.JetSpace(np.float64, 2))
    p1, p2 = Jets.from_numpy(np.array([1.0, 2.0])), Jets.from_numpy(np.array([3.0, 4.0]))
    dm = Jets.from_numpy(np.array([1.0, 2.0]))
    J1, J2 = Jets.jacobian(F, p1), Jets.jacobian(F, p2)
    assert (J1 * dm).to_numpy().tolist() == [2.0, 8.0]                               # 2 .* [1,2] .* dm
    assert (J2 * dm).to_numpy().tolist() == [6.0, 16.0]                              # 2 .* [3,4] .* dm
    J1, J2 = Jets.jacobian_(F, p1), Jets.jacobian_(F, p2)
    assert (J1 * dm).to_numpy().tolist() == [6.0, 16.0]
    assert (J2 * dm).to_numpy().tolist() == (J1 * dm).to_numpy().tolist()
    # the same through a block operator: jacobian of a copy must not disturb the original's fused handle
    G = Jets.blockop([[Jets.JopSquare(Jets.JetSpace(np.float64, 2))] for _ in range(2)])
    K1, K2 = Jets.jacobian(G, p1), Jets.jacobian(G, p2)
    assert (K1 * dm).to_numpy().tolist() == [2.0, 8.0, 2.0, 8.0]
    assert (K2 * dm).to_numpy().tolist() == [6.0, 16.0, 6.0, 16.0]
    assert (K1.H * (K1 * dm)).to_numpy().tolist() == [8.0, 64.0]                     # sum_i (2 mo)^2 dm


# ---------------------------------------------------------------------------------- arbitrary elementwise nonlinear children
def test_elementwise_operator_and_its_jacobian(Jets, oracle):
    """JopElementwise: f and f' as expressions; point! refreshes the Jacobian's diagonal in one fused pass."""
    dt, n = np.float64, 4096
    spc = Jets.JetSpace(dt, n)
    F = Jets.JopElementwise(spc, "s0*x0*x0*x0 + exp(x0)", "3*s0*x0*x0 + exp(x0)", [0.5])
    m = Jets.rand(spc, seed=91, stream=0)
    hm = u01(oracle, dt, 91, 0, n)
    np.testing.assert_allclose((F * m).to_numpy(), 0.5 * hm ** 3 + np.exp(hm), rtol=1e-14)
    with pytest.raises(ValueError, match="linearization point"):
        Jets.mul(Jets.JopLn(Jets.jet(F)), m)
    J = Jets.jacobian_(F, m)
    dm = Jets.rand(spc, seed=92, stream=0)
    hdm = u01(oracle, dt, 92, 0, n)
    np.testing.assert_allclose((J * dm).to_numpy(), (1.5 * hm ** 2 + np.exp(hm)) * hdm, rtol=1e-14)
    np.testing.assert_allclose((J.H * dm).to_numpy(), (1.5 * hm ** 2 + np.exp(hm)) * hdm, rtol=1e-14)
    lhs, rhs = Jets.dot_product_test(J, Jets.rand(spc), Jets.rand(spc))
    assert abs(lhs - rhs) <= 1e-12 * abs(lhs + rhs)
    # linearization test in the reference's sense (src/Jets.jl:1228-1269): F(m + h dm) - F(m) - h J dm = O(h^2)
    errs = []
    for h in (1e-2, 1e-3):
        mp = Jets.zeros(spc)
        Jets.lincomb_(mp, [1.0, h], [m, dm])
        r = Jets.zeros(spc)
        Jets.lincomb_(r, [1.0, -1.0, -h], [F * mp, F * m, J * dm])
        errs.append(float(Jets.norm(r)))
    assert errs[1] < errs[0] / 50                                                     # second order: ~100x smaller for 10x smaller h
    # two simultaneous linearizations through jacobian (copies the jet AND its diagonal)
    m2 = Jets.rand(spc, seed=93, stream=0)
    hm2 = u01(oracle, dt, 93, 0, n)
    J1, J2 = Jets.jacobian(F, m), Jets.jacobian(F, m2)
    np.testing.assert_allclose((J1 * dm).to_numpy(), (1.5 * hm ** 2 + np.exp(hm)) * hdm, rtol=1e-14)
    np.testing.assert_allclose((J2 * dm).to_numpy(), (1.5 * hm2 ** 2 + np.exp(hm2)) * hdm, rtol=1e-14)


def test_tall_operator_of_elementwise_children_linearises_onto_the_fast_path(Jets, oracle):
    """A tall block operator of JopElementwise children: f! loops over the children (one fused pass each), the Jacobian is
    a native all-DIAG operator -- the tall kernels, the fused A'A and the one-pass LSQR step apply -- and follows point!
    without rebuilding the device handle (the diagonals are refreshed in place)."""
    import ctypes as C
    from jets_jl_amd._ffi import lib
    from jets_jl_amd import jetblock

    dt, n, nrow = np.float32, 1 << 14, 5
    spc = Jets.JetSpace(dt, n)
    exprs = [("x0*x0", "2*x0"), ("sin(x0)", "cos(x0)"), ("s0*x0", "s0"), ("exp(-x0)", "-exp(-x0)"), ("x0*x0*x0", "3*x0*x0")]
    F = Jets.blockop([[Jets.JopElementwise(spc, f, j, [1.5])] for f, j in exprs])
    assert isinstance(F, Jets.JopNl)
    m = Jets.rand(spc, seed=94, stream=0)
    hm = u01(oracle, dt, 94, 0, n).astype(np.float64)
    d = F * m
    want = np.concatenate([hm * hm, np.sin(hm), 1.5 * hm, np.exp(-hm), hm ** 3])
    np.testing.assert_allclose(d.to_numpy(), want, rtol=3e-6, atol=1e-6)
    J = Jets.jacobian_(F, m)
    coeffs = [2 * hm, np.cos(hm), np.full(n, 1.5), -np.exp(-hm), 3 * hm * hm]
    dm = Jets.rand(spc, seed=95, stream=0)
    hdm = u01(oracle, dt, 95, 0, n).astype(np.float64)
    np.testing.assert_allclose((J * dm).to_numpy(), np.concatenate([c * hdm for c in coeffs]), rtol=3e-6, atol=1e-6)
    dd = J * dm
    np.testing.assert_allclose((J.H * dd).to_numpy(), sum(c * c for c in coeffs) * hdm, rtol=1e-5, atol=1e-5)
    nat = jetblock._native_op(J.jet.s["_native"], J.jet.s["ops"], J.jet.rng.eltype())
    assert nat is not None and nat.host_f and not nat.nonlinear
    w, out = Jets.zeros(spc), C.c_double(0)
    u = Jets.rand(Jets.range(J), seed=96, stream=0)
    assert lib.jh_blockop_bidiag_step(nat.handle, u.handle, dm.handle, w.handle, 1.0, 0.0, C.byref(out)) == 0   # the all-DIAG fast path
    np.testing.assert_allclose(Jets.mul(J.H @ J, dm).to_numpy(), sum(c * c for c in coeffs) * hdm, rtol=1e-5, atol=1e-5)
    handle_before = nat.handle.value
    m2 = Jets.rand(spc, seed=97, stream=0)
    hm2 = u01(oracle, dt, 97, 0, n).astype(np.float64)
    Jets.point_(F, m2)                                                               # J shares the jet: new point, same device handle
    coeffs2 = [2 * hm2, np.cos(hm2), np.full(n, 1.5), -np.exp(-hm2), 3 * hm2 * hm2]
    np.testing.assert_allclose((J * dm).to_numpy(), np.concatenate([c * hdm for c in coeffs2]), rtol=3e-6, atol=1e-6)
    assert jetblock._native_op(J.jet.s["_native"], J.jet.s["ops"], J.jet.rng.eltype()).handle.value == handle_before


def test_gauss_newton_with_elementwise_children_and_device_lsqr(Jets, oracle):
    dt, n = np.float64, 4096
    spc = Jets.JetSpace(dt, n)
    F = Jets.blockop([[Jets.JopElementwise(spc, "exp(x0)", "exp(x0)")], [Jets.JopElementwise(spc, "x0*x0*x0", "3*x0*x0")],
                      [Jets.JopSquare(spc)]])
    x_true = 0.25 + u01(oracle, dt, 98, 0, n)
    dobs = Jets.from_numpy(np.concatenate([np.exp(x_true), x_true ** 3, x_true ** 2]), Jets.range(F))
    m = Jets.ones(spc)
    for _ in range(8):
        r = Jets.zeros(Jets.range(F))
        Jets.mul_(r, F, m)
        Jets.lincomb_(r, [1.0, -1.0], [dobs, r])
        J = Jets.jacobian_(F, m)
        step = Jets.lsqr(J, r, maxiter=30, atol=1e-14, btol=1e-14).x
        Jets.lincomb_(m, [1.0, 1.0], [m, step])
    np.testing.assert_allclose(m.to_numpy(), x_true, rtol=1e-9)


def test_linearization_and_linearity_tests(Jets, oracle):
    """test/runtests.jl:920-930: linearization_test on JopBar (observed ratios == (mu_{i-1}/mu_i)^2 for a quadratic f),
    linearity_test on JopFoo; and on block operators."""
    spc = Jets.JetSpace(np.float64, 10)
    F = Jets.JopSquare(spc)
    mu_obs, mu_exp = Jets.linearization_test(F, Jets.rand(spc), seed=7)
    assert np.max(mu_obs) == pytest.approx(np.max(mu_exp), rel=1e-8)                  # @test maximum(mu_obs) ≈ maximum(mu_exp)
    np.testing.assert_allclose(mu_obs, 4.0, rtol=1e-8)                                # exactly second order: m.^2
    A = Jets.JopDiagonal(Jets.rand(spc))
    lhs, rhs = Jets.linearity_test(A)
    np.testing.assert_allclose(lhs.to_numpy(), rhs.to_numpy(), rtol=1e-13)            # @test lhs ≈ rhs
    R = Jets.JetSpace(np.float64, 2000)
    G = Jets.blockop([[Jets.JopElementwise(R, "exp(x0)", "exp(x0)")], [Jets.JopSquare(R)], [Jets.JopElementwise(R, "sin(x0)", "cos(x0)")]])
    mu_obs, mu_exp = Jets.linearization_test(G, Jets.rand(R), mu=(0.1, 0.05, 0.025, 0.0125), seed=8)
    np.testing.assert_allclose(mu_obs, mu_exp, rtol=0.15)                             # second order up to O(mu^3) terms
    J = Jets.jacobian_(G, Jets.rand(R))
    lhs, rhs = Jets.linearity_test(J)
    np.testing.assert_allclose(lhs.to_numpy(), rhs.to_numpy(), rtol=1e-12, atol=1e-13)
    lhs, rhs = Jets.linearity_test(J.H)
    np.testing.assert_allclose(lhs.to_numpy(), rhs.to_numpy(), rtol=1e-12, atol=1e-13)


def test_repointing_at_the_same_vector_replays_and_follows_the_contents(Jets, oracle):
    """A Gauss-Newton loop updates its model IN PLACE and calls jacobian! again with the same vector: the block point! then
    replays the children's packed upstate! batch instead of walking them -- and must pick up the new contents; a state!
    on a child or a point! at another vector invalidates the replay."""
    dt, nrow, n = np.float64, 12, 256
    spc = Jets.JetSpace(dt, n)
    F = Jets.blockop([[Jets.JopElementwise(spc, "s0*x0*x0*x0", "3*s0*x0*x0", [0.5 + i])] for i in range(nrow)])
    m = Jets.rand(Jets.domain(F), seed=2, stream=0)
    dm = Jets.rand(Jets.domain(F), seed=3, stream=0)
    hdm = dm.to_numpy().ravel(order="F")

    def expect(hm, scales):
        return np.concatenate([(3 * scales[i] * hm * hm) * hdm for i in range(nrow)])

    scales = [0.5 + i for i in range(nrow)]
    J1 = Jets.jacobian_(F, m)
    assert np.allclose((J1 * dm).to_numpy(), expect(m.to_numpy().ravel(order="F"), scales), rtol=1e-14)
    m.assign(2.0 * m)                                                             # in place: same vector, new contents
    J2 = Jets.jacobian_(F, m)                                                     # replayed batch
    assert np.allclose((J2 * dm).to_numpy(), expect(m.to_numpy().ravel(order="F"), scales), rtol=1e-14)
    Jets.state_(Jets.getblock_op(F, 0, 0, Jets.JopNl), {"params": (7.0,)})        # a child's parameter changes
    scales[0] = 7.0
    J3 = Jets.jacobian_(F, m)
    assert np.allclose((J3 * dm).to_numpy(), expect(m.to_numpy().ravel(order="F"), scales), rtol=1e-14)
    m2 = Jets.rand(Jets.domain(F), seed=9, stream=0)                              # another vector
    J4 = Jets.jacobian_(F, m2)
    assert np.allclose((J4 * dm).to_numpy(), expect(m2.to_numpy().ravel(order="F"), scales), rtol=1e-14)
    d1, d2 = Jets.zeros(Jets.range(F)), Jets.zeros(Jets.range(F))                 # F(m) replays its packed batch too
    Jets.mul_(d1, F, m2)
    Jets.mul_(d1, F, m2)
    hm2 = m2.to_numpy().ravel(order="F")
    assert np.allclose(d1.to_numpy(), np.concatenate([scales[i] * hm2 ** 3 for i in range(nrow)]), rtol=1e-14)
    Jets.mul_(d2, F, m)
    hm = m.to_numpy().ravel(order="F")
    assert np.allclose(d2.to_numpy(), np.concatenate([scales[i] * hm ** 3 for i in range(nrow)]), rtol=1e-14)


@pytest.mark.parametrize("dt", [np.float32, np.float64, np.complex64])
def test_f_mode_of_big_dense_children_in_mixed_company_takes_two_launches(Jets, oracle, dt):
    """Round 4: JetBlock_f! (src/Jets.jl:988-1008) of an operator that mixes dense children too big for the one-launch loop with a
    nonlinear child (JopBar's d .= m.^2, test/runtests.jl:19-24), a diagonal and a zero block: ONE batched GEMV launch for the dense
    children + ONE launch of the general kernel in f! mode (a zero block's `d .= 0` is added, not skipped), where the per-block loop
    made two launches per block.  The oracle's bits, and those of the loop it replaces."""
    J = Jets
    n1 = {4: 352, 8: 256}[np.dtype(dt).itemsize]                  # 484 KiB / 512 KiB children: beyond the one-launch loop's 256 KiB
    len_r, len_c = [n1, n1, n1], [n1, n1, n1]
    kinds = [["dense", "square", "zero"], ["zero", "dense", "diag"], ["square", "zero", "dense"]]
    hmo = _split(u01(oracle, dt, 43, 0, sum(len_c)), len_c)
    F, ops = _mixed(J, oracle, dt, len_r, len_c, kinds, seed=960, hmo=hmo)
    mo = J.rand(J.domain(F), seed=43, stream=0)
    hd = _split(u01(oracle, dt, 44, 0, sum(len_r)), len_r)
    ref = oracle.block_f(ops, [x.copy() for x in hd], hmo)
    got = {}
    for knob in (1, 0, 2):                                         # batched route with columns in order / the per-block loop / batched, automatic lane layout
        J.tune(dense_mixed=1 if knob else 0, dense_list_split=1 if knob == 2 else 0)
        try:
            d = J.rand(J.range(F), seed=44, stream=0)              # dirty: `_d .+=` accumulates into d as found (1001)
            J.mul_(d, F, mo)
            if knob:
                assert 1 <= J.tune_get("last_launches") <= 2, "one batched launch for the dense children + the combine"
            got[knob] = d.to_numpy()
        finally:
            J.tune(dense_mixed=1, dense_list_split=1)
    assert_bits_equal(got[1], np.concatenate(ref), "f! with big dense children: batched route vs the oracle")
    assert_bits_equal(got[1], got[0], "f! with big dense children: batched route vs the per-block loop")
    # late round 5: by default the list kernel splits a child's columns over lane groups of a workgroup: deterministic, tolerance parity (as any BLAS gemv)
    want = np.concatenate(ref)
    assert np.linalg.norm(got[2] - want) / np.linalg.norm(want) < (1e-5 if np.dtype(dt).itemsize <= 8 and np.dtype(dt) != np.float64 else 1e-12)


@pytest.mark.parametrize("dt", [np.float32, np.complex128])
def test_batched_broadcast_of_a_tall_nonlinear_operator_in_column_bands(Jets, dt):
    """F(m) of a tall operator of elementwise children is ONE batched launch whose items share the model vector; for big children the items are the
    fastest block index, in column bands of 32 tiles since late round 4 (jh_bcast.hip: item_fast_ > 1).  Forced here on small children
    (knob bcast_item_fast = 1) with bands narrower than, equal to and wider than the vector's tile count, a ragged last tile: the same bits as the
    item-by-item evaluation."""
    J = Jets
    nrow, n = 12, 70004                                            # 69 tiles of 256 packs (Float32), the last one ragged
    spc = J.JetSpace(dt, n)
    F = J.blockop([[J.JopElementwise(spc, "x0*x0 + x0", "2*x0 + 1")] for _ in range(nrow)])
    rng = np.random.default_rng(5)
    hm = (rng.standard_normal(n) + (1j * rng.standard_normal(n) if np.dtype(dt).kind == "c" else 0)).astype(dt)
    m = J.from_numpy(hm, spc)
    with np.errstate(all="ignore"):
        want = np.concatenate([hm * hm + hm] * nrow)
    ref = J.mul(F, m).to_numpy()
    try:
        J.tune(bcast_item_fast=1)
        for band in (1, 3, 32, 34, 1000):
            J.tune(bcast_band=band)
            d = J.zeros(J.range(F))
            J.mul_(d, F, m)
            got = d.to_numpy()
            assert got.tobytes() == ref.tobytes(), f"bands of {band} tiles against the default order, {np.dtype(dt)}"
            np.testing.assert_allclose(got, want, rtol=1e-5 if np.dtype(dt) == np.dtype(np.float32) else 1e-12)
    finally:
        J.tune(bcast_item_fast=-1, bcast_band=0)
    J.close(F)


def test_repeated_batched_broadcast_reuses_its_tables_and_notices_changes(Jets):
    """F(m) of a tall nonlinear operator into the same vectors again and again goes straight to the launches (jh_bcast.hip: the device tables of
    the last batch are kept while the argument arrays are the same and no vector handle has died since).  The tables hold POINTERS, so new
    contents of m are seen; another destination, a destroyed handle, another model vector all rebuild them."""
    import gc

    J = Jets
    nrow, n = 64, 4096
    spc = J.JetSpace(np.float32, n)
    F = J.blockop([[J.JopElementwise(spc, "x0*x0 + x0", "2*x0 + 1")] for _ in range(nrow)])
    rng = np.random.default_rng(9)

    def check(d, m, hm, what):
        J.mul_(d, F, m)
        got = d.to_numpy()
        want = np.concatenate([hm * hm + hm] * nrow)
        assert got.tobytes() == want.tobytes(), what

    hm = rng.standard_normal(n).astype(np.float32)
    m = J.from_numpy(hm, spc)
    d = J.zeros(J.range(F))
    check(d, m, hm, "first call (tables built)")
    check(d, m, hm, "second call (tables re-used)")
    hm2 = rng.standard_normal(n).astype(np.float32)
    m.assign(hm2)                                                  # same handle, new contents
    check(d, m, hm2, "new contents of m through the kept tables")
    d2 = J.zeros(J.range(F))
    check(d2, m, hm2, "another destination")
    check(d, m, hm2, "back to the first destination")
    scratch = J.zeros(spc)
    scratch.close()                                                # a handle died: the kept tables are not trusted any more
    check(d, m, hm2, "after an unrelated vector was destroyed")
    d.close()
    del d
    gc.collect()
    d3 = J.zeros(J.range(F))                                       # may well get d's slab and d's handle address back
    m3 = J.from_numpy(hm, spc)
    check(d3, m3, hm, "a new destination and a new model vector")
    check(d3, m3, hm, "and again")
    J.close(F)
