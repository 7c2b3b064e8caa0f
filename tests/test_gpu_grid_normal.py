"""GPU parity: the fused normal operator (A' o A) of an N x K GRID of equal diagonals, K = 2 .. 4 (round 6; jh_grid_normal.hip behind
jh_blockop_normal_mul).

The reference applies the composite (A', A) stage by stage (src/Jets.jl:530-534): JetBlock_df! into zeros(range(A)) (1010-1032), then
JetBlock_df'! (1034-1057).  The fused pass keeps m_1 .. m_K and y_1 .. y_K in registers, reads every coefficient once and rounds every
product and sum where the two stages round them, so it is BIT-EXACT against the two stages applied on the device and against the CPU
oracle's two loops; with many rows of small blocks it sums in parts like every row-summing kernel (tolerance; adj_split = 0: ordered, bit-exact)."""
import ctypes as C

import numpy as np
import pytest

from .helpers import DTYPES, assert_bits_equal, u01
from .test_gpu_blockop import _mixed_ops

pytestmark = pytest.mark.gpu


def _native(J, A):
    from jets_jl_amd import jetblock as _blk

    return _blk._native_op(A.jet.s["_native"], A.jet.s["ops"], A.jet.rng.eltype())


def _grid(J, oracle, dt, nrow, ncol, n, adjointed=False):
    kinds = [[("diag_adj" if (adjointed and (i + j) % 3 == 0) else "diag") for j in range(ncol)] for i in range(nrow)]
    return _mixed_ops(J, oracle, dt, kinds, [n] * nrow, [n] * ncol, seed=41)


def _two_stage(J, A, m):
    t = J.mul_(J.zeros(J.range(A)), A, m)                                       # zeros(range(A)) (531): the grid forward adds to d as found (1024)
    return J.mul_(J.rand(J.domain(A), seed=6, stream=9), A.H, t)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("ncol", [2, 3, 4])
@pytest.mark.parametrize("nrow,n", [(2, 1024), (5, 1027), (9, 4096 + 64), (7, 67), (3, 6)])
def test_fused_normal_of_a_grid_of_diagonals_has_the_bits_of_the_two_stages(Jets, oracle, dt, ncol, nrow, n):
    J = Jets
    if n * np.dtype(dt).itemsize < 16:
        pytest.skip("blocks shorter than one pack take the chain")
    A, ora = _grid(J, oracle, dt, nrow, ncol, n)
    hm = [u01(oracle, dt, 91, j, n) for j in range(ncol)]
    m = J.from_numpy(np.concatenate(hm), J.domain(A))
    nat = _native(J, A)
    y = J.rand(J.domain(A), seed=7, stream=3)                                   # a DIRTY output
    from jets_jl_amd._ffi import check, lib

    check(lib.jh_blockop_normal_mul(nat.handle, y.handle, m.handle))            # the library takes the grid (no JH_ERR_UNSUPPORTED)
    y2 = J.mul_(J.rand(J.domain(A), seed=8, stream=3), J.compose(A.H, A), m)    # the same through the composite
    want_dev = _two_stage(J, A, m)
    t = oracle.block_df(ora, [np.zeros(n, dt) for _ in range(nrow)], hm)
    want = np.concatenate(oracle.block_df_adj(ora, [np.zeros(n, dt) for _ in range(ncol)], t))
    assert_bits_equal(y.to_numpy().ravel(order="F"), want, "fused A'A of a grid vs the oracle's two loops")
    assert_bits_equal(y2.to_numpy().ravel(order="F"), want, "the composite (A', A) takes the fused pass")
    assert_bits_equal(want_dev.to_numpy().ravel(order="F"), want, "the two stages on the device")
    J.tune(grid_normal=0)
    try:
        y3 = J.mul_(J.rand(J.domain(A), seed=9, stream=3), J.compose(A.H, A), m)   # knob off: JH_ERR_UNSUPPORTED -> the reference's chain
        assert_bits_equal(y3.to_numpy().ravel(order="F"), want, "knob off: the chain")
    finally:
        J.tune(grid_normal=1)
    J.close(A)


@pytest.mark.parametrize("dt", [np.float32, np.complex64, np.complex128])
def test_grids_the_fused_pass_declines_run_the_chain(Jets, oracle, dt):
    """Five block columns, or blocks that are not all plain diagonals (an adjointed diagonal of a complex type): jh_blockop_normal_mul says
    JH_ERR_UNSUPPORTED and the composite applies its two stages."""
    J = Jets
    n = 515
    for ncol, adj in ((5, False), (3, True)):
        if adj and np.dtype(dt).kind != "c":
            continue
        A, ora = _grid(J, oracle, dt, 4, ncol, n, adjointed=adj)
        hm = [u01(oracle, dt, 91, j, n) for j in range(ncol)]
        m = J.from_numpy(np.concatenate(hm), J.domain(A))
        y = J.mul_(J.rand(J.domain(A), seed=8, stream=3), J.compose(A.H, A), m)
        t = oracle.block_df(ora, [np.zeros(n, dt) for _ in range(4)], hm)
        want = np.concatenate(oracle.block_df_adj(ora, [np.zeros(n, dt) for _ in range(ncol)], t))
        assert_bits_equal(y.to_numpy().ravel(order="F"), want, f"{ncol} columns, adjointed={adj}")
        J.close(A)


@pytest.mark.parametrize("dt", [np.float32, np.float64, np.complex64])
@pytest.mark.parametrize("ncol", [2, 4])
def test_many_small_rows_take_the_split_walk(Jets, oracle, dt, ncol):
    J = Jets
    nrow, n = 600, 515
    A, ora = _grid(J, oracle, dt, nrow, ncol, n)
    hm = [u01(oracle, dt, 91, j, n) for j in range(ncol)]
    m = J.from_numpy(np.concatenate(hm), J.domain(A))
    N = J.compose(A.H, A)
    t = oracle.block_df(ora, [np.zeros(n, dt) for _ in range(nrow)], hm)
    want = np.concatenate(oracle.block_df_adj(ora, [np.zeros(n, dt) for _ in range(ncol)], t))
    y = J.mul_(J.rand(J.domain(A), seed=8, stream=3), N, m)
    assert J.tune_get("last_adj_parts") > 1
    tol = (2e-5 if np.dtype(dt).itemsize // (2 if np.dtype(dt).kind == "c" else 1) == 4 else 1e-13) * np.sqrt(nrow) * np.abs(want).max()
    assert np.abs(y.to_numpy().ravel(order="F") - want).max() <= tol
    y2 = J.mul_(J.rand(J.domain(A), seed=9, stream=3), N, m)
    assert_bits_equal(y2.to_numpy().ravel(order="F"), y.to_numpy().ravel(order="F"), "the split walk is deterministic")
    J.tune(adj_split=0)
    try:
        y3 = J.mul_(J.rand(J.domain(A), seed=10, stream=3), N, m)
        assert J.tune_get("last_adj_parts") == 1
        assert_bits_equal(y3.to_numpy().ravel(order="F"), want, "ordered walk: the oracle's bits")
    finally:
        J.tune(adj_split=-1)
    J.close(A)


@pytest.mark.parametrize("dt,xtol", [(np.float32, 2e-4), (np.float64, 1e-11), (np.complex64, 2e-4)])
@pytest.mark.parametrize("native", ["1", "0"])
def test_cgnr_on_a_grid_runs_through_the_fused_normal_operator(Jets, oracle, dt, xtol, native, monkeypatch):
    """CG on the normal equations of a 6 x 3 grid of diagonals (jh_cgnr_solve takes such grids since round 6: one pass over the coefficients per
    iteration; native = 0: A then A' through the engines) against the textbook fp64 CGLS on the same numbers."""
    from .test_gpu_cgls import cgls_fp64

    monkeypatch.setenv("JETS_CGLS_NATIVE", native)
    J = Jets
    nrow, ncol, n, iters = 6, 3, 1500, 10
    A, ora = _grid(J, oracle, dt, nrow, ncol, n)
    dt64 = np.complex128 if np.dtype(dt).kind == "c" else np.float64
    coef = [[ora[i][j].coeff.astype(dt64) for j in range(ncol)] for i in range(nrow)]

    def matvec(x):
        xs = np.split(x, ncol)
        return np.concatenate([sum(coef[i][j] * xs[j] for j in range(ncol)) for i in range(nrow)])

    def rmatvec(d):
        ds = np.split(d, nrow)
        return np.concatenate([sum(np.conj(coef[i][j]) * ds[i] for i in range(nrow)) for j in range(ncol)])

    hb = (u01(oracle, dt, 51, 0, nrow * n) - dt(0.5)).astype(dt)
    b = J.from_numpy(hb, J.range(A))
    res = J.cgnr(A, b, atol=0.0, btol=0.0, maxiter=iters)
    xr, info = cgls_fp64(matvec, rmatvec, hb.astype(dt64), ncol * n, atol=0.0, btol=0.0, maxiter=iters)
    assert res.itn == iters == info["itn"] and res.istop == 7
    x = res.x.to_numpy().ravel(order="F").astype(dt64)
    assert np.linalg.norm(x - xr) / np.linalg.norm(xr) < xtol
    for (i1, r1, ar1), (i2, r2, ar2) in zip(res.history, info["history"]):
        assert i1 == i2 and r1 == pytest.approx(r2, rel=max(10 * xtol, 1e-8))
    assert np.array_equal(b.to_numpy(), hb), "b is read, never written"
    J.close(A)


def test_lsqr_and_cgls_on_a_grid_reuse_their_temporaries_correctly(Jets, oracle):
    """A block operator with several columns ADDS to the output as found (src/Jets.jl:1024); the solvers' generic engine reuses one range-sized
    temporary across iterations and must hand mul! zeros each time, as the reference's own `A*m` does (395).  LSQR and CGLS on a 5 x 2 grid against
    the textbook fp64 CGLS (the same Krylov iterates in exact arithmetic)."""
    from .test_gpu_cgls import cgls_fp64

    J = Jets
    dt, nrow, ncol, n, iters = np.float64, 5, 2, 700, 8
    A, ora = _grid(J, oracle, dt, nrow, ncol, n)
    coef = [[ora[i][j].coeff for j in range(ncol)] for i in range(nrow)]
    matvec = lambda x: np.concatenate([sum(coef[i][j] * np.split(x, ncol)[j] for j in range(ncol)) for i in range(nrow)])
    rmatvec = lambda d: np.concatenate([sum(coef[i][j] * np.split(d, nrow)[i] for i in range(nrow)) for j in range(ncol)])
    hb = u01(oracle, dt, 51, 0, nrow * n) - 0.5
    xr, _ = cgls_fp64(matvec, rmatvec, hb, ncol * n, atol=0.0, btol=0.0, maxiter=iters)
    for solve in (J.lsqr, J.cgls):
        kw = dict(conlim=0.0) if solve is J.lsqr else {}
        res = solve(A, J.from_numpy(hb, J.range(A)), atol=0.0, btol=0.0, maxiter=iters, **kw)
        x = res.x.to_numpy().ravel(order="F")
        assert np.linalg.norm(x - xr) <= 1e-9 * np.linalg.norm(xr), solve.__name__
    J.close(A)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("ncol", [2, 3, 4])
@pytest.mark.parametrize("nrow,n", [(3, 1024), (7, 1027), (11, 67), (600, 515)])
def test_fused_normal_of_a_grid_with_blocks_of_several_kinds(Jets, oracle, dt, ncol, nrow, n):
    """The regularised multi-parameter operator -- data rows of diagonals over rows of zero / identity / scalar / adjointed blocks: a packed table of one word
    per block, batches of plain rows on the tight loop (jh_grid_normal.hip: k_grid_normal_mixed).  The two stages' bits (zero blocks skipped: 1022 / 1047)."""
    J = Jets
    names = ["diag", "zero", "identity", "scale", "diag_adj", "diag"]
    kinds = [[("diag" if i < max(1, nrow - ncol - 2) and (i % 5 != 3) else names[(2 * i + 3 * j) % 6]) for j in range(ncol)] for i in range(nrow)]
    kinds[nrow - 1] = ["zero"] * ncol                                            # a whole row of zero blocks
    A, ora = _mixed_ops(J, oracle, dt, kinds, [n] * nrow, [n] * ncol, seed=43)
    hm = [u01(oracle, dt, 91, j, n) - dt(0.5) for j in range(ncol)]
    m = J.from_numpy(np.concatenate(hm).astype(dt), J.domain(A))
    hm = [h.astype(dt) for h in hm]
    nat = _native(J, A)
    from jets_jl_amd._ffi import check, lib

    J.tune(adj_split=0)
    try:
        y = J.rand(J.domain(A), seed=7, stream=3)
        check(lib.jh_blockop_normal_mul(nat.handle, y.handle, m.handle))        # the library takes the mixed grid
        t = oracle.block_df(ora, [np.zeros(n, dt) for _ in range(nrow)], hm)
        want = np.concatenate(oracle.block_df_adj(ora, [np.zeros(n, dt) for _ in range(ncol)], t))
        assert_bits_equal(y.to_numpy().ravel(order="F"), want, "fused A'A of a mixed grid vs the oracle's two loops")
        y2 = J.mul_(J.rand(J.domain(A), seed=8, stream=3), J.compose(A.H, A), m)
        assert_bits_equal(y2.to_numpy().ravel(order="F"), want, "through the composite")
        J.tune(grid_normal=2)                                                   # plain diagonals only: the composite chains its two stages
        y3 = J.mul_(J.rand(J.domain(A), seed=9, stream=3), J.compose(A.H, A), m)
        assert_bits_equal(y3.to_numpy().ravel(order="F"), want, "knob 2: the chain")
    finally:
        J.tune(grid_normal=1, adj_split=-1)
    if nrow >= 256:                                                             # the split walk: tolerance, deterministic
        y4 = J.mul_(J.rand(J.domain(A), seed=10, stream=3), J.compose(A.H, A), m)
        assert J.tune_get("last_adj_parts") > 1
        tol = (2e-5 if np.dtype(dt).itemsize // (2 if np.dtype(dt).kind == "c" else 1) == 4 else 1e-13) * np.sqrt(nrow) * np.abs(want).max()
        assert np.abs(y4.to_numpy().ravel(order="F") - want).max() <= tol
    J.close(A)
